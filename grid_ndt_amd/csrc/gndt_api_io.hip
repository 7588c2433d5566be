// gndt_api_io.hip — input side: raw records -> packed xyz (gndt_pack.hpp), gndt_build_cloud = the callback's body.
#include "gndt_handle.hpp"
#include "gndt_pack.hpp"
using namespace gndt;
using namespace gndt_host;

extern "C" {

int gndt_pack_points_device(gndt_handle* h, const void* raw_dev, size_t n, const gndt_point_layout* layout,
                            float* xyz_out_dev, uint64_t* n_valid, void* hip_stream) {
    if (!h || !layout || !n_valid) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    *n_valid = 0;
    if (n == 0) return GNDT_OK;
    if (!raw_dev || !xyz_out_dev) { h->err = "null buffer"; return GNDT_ERR_INVALID; }
    if (n >= 0xFFFFFFFFull) { h->err = "too many points"; return GNDT_ERR_INVALID; }
    const PointLayout L{layout->point_step, layout->offset_x, layout->offset_y, layout->offset_z};
    if (L.point_step < 12 || (L.point_step & 3u) || ((L.off_x | L.off_y | L.off_z) & 3u) ||
        std::max(L.off_x, std::max(L.off_y, L.off_z)) + 4 > L.point_step) {
        h->err = "point layout: 4-byte aligned float fields inside a record of point_step bytes expected";
        return GNDT_ERR_INVALID;
    }
    hipStream_t s = stream_of(h, hip_stream);
    int rc = partition_resolve(h);              // the buffers below are shared with a pending build's ordering pass
    if (rc) return rc;
    auto& q = h->part;
    const uint64_t words = ((n + 63) / 64) * 2;
    if ((rc = ensure_words(h, words))) return rc;
    h->incr_ok = false;                        // the order's bitmap is used as scratch here
    if (!h->d_nvalid) {
        HIP_TRY(h, hipMalloc(&h->d_nvalid, sizeof(uint32_t)));
        HIP_TRY(h, hipHostMalloc(&h->h_nvalid, sizeof(uint32_t)));
    }
    const unsigned char* raw = static_cast<const unsigned char*>(raw_dev);
    hipLaunchKernelGGL(k_pack_flags, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, raw, (uint64_t)n, L,
                       reinterpret_cast<unsigned long long*>(q.bitmap));
    const uint32_t nbw = (uint32_t)((words + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(k_scan_reduce<true>, dim3(nbw), dim3(kScanThreads), 0, s, q.bitmap, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words);
    hipLaunchKernelGGL(k_scan_apply<true>, dim3(nbw), dim3(kScanThreads), 0, s, q.bitmap, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words, q.word_base);
    hipLaunchKernelGGL(k_pack_write, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, raw, (uint64_t)n, L, q.bitmap,
                       q.word_base, xyz_out_dev, h->d_nvalid);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_nvalid, h->d_nvalid, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    *n_valid = *h->h_nvalid;
    return GNDT_OK;
}

int gndt_build_cloud(gndt_handle* h, const void* raw_host, size_t n, const gndt_point_layout* layout) {
    if (!h || !layout) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!raw_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    if (n == 0) { h->err = "empty cloud: there is no first point to take the origin from (receiver.cpp:145)"; return GNDT_ERR_INVALID; }
    int rc = stage_host_input(h, raw_host, n, layout->point_step, h->own_stream);
    if (rc) return rc;
    if (n > h->packed_cap) {
        release_device(h, h->packed);
        h->packed = nullptr; h->packed_cap = 0;
        HIP_TRY(h, hipMalloc(&h->packed, n * 12));
        h->packed_cap = n;
    }
    uint64_t valid = 0;
    rc = gndt_pack_points_device(h, h->stage, n, layout, h->packed, &valid, h->own_stream);
    if (rc) return rc;
    if (valid == 0) { h->err = "no finite point in the cloud"; return GNDT_ERR_INVALID; }
    float origin[3];
    HIP_TRY(h, hipMemcpy(origin, h->packed, sizeof origin, hipMemcpyDeviceToHost));
    if (h->table_dirty) { rc = gndt_reset(h, h->own_stream); if (rc) return rc; }
    rc = gndt_set_origin(h, origin);
    if (rc) return rc;
    rc = gndt_build_device(h, h->packed + 3, (size_t)valid - 1, 12, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

}  // extern "C"
