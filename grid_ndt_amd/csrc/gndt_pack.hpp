// gndt_pack.hpp — the input side on the device: raw point records as a PointCloud2 /
// PCD payload lays them out (fields at byte offsets inside a `point_step`-byte record) -> packed fp32 xyz, with the
// rows holding a non-finite coordinate dropped and the order kept.
//   sensor_msgs::PointCloud2 -> pcl::PointCloud<pcl::PointXYZ>   src/receiver.cpp:140-143 (pcl::fromPCLPointCloud2)
//   pcl::removeNaNFromPointCloud                                  src/publisher.cpp:24-26
// Order matters downstream: point 0 is the origin (receiver.cpp:145) and nodes are listed in first-seen order.
// Stable compaction without a sort: one validity bit per point (a wave ballot = one 64-bit store), popcount
// prefix over the bitmap words (the two-level scan kernels of gndt_partition.hpp), destination = prefix + popcount below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_partition.hpp"

namespace gndt {

struct PointLayout {
    uint32_t point_step, off_x, off_y, off_z;    // bytes; offsets are multiples of 4
};

__device__ __forceinline__ float load_field(const unsigned char* rec, uint32_t off) {
    return *reinterpret_cast<const float*>(rec + off);
}

// grid covers n rounded up to whole waves
static __global__ void __launch_bounds__(256) k_pack_flags(const unsigned char* __restrict__ raw, uint64_t n, PointLayout L,
                                                    unsigned long long* __restrict__ valid_bits) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ull;
    for (uint64_t i = gid; i < n_round; i += stride) {
        bool ok = false;
        if (i < n) {
            const unsigned char* rec = raw + i * L.point_step;
            const float x = load_field(rec, L.off_x), y = load_field(rec, L.off_y), z = load_field(rec, L.off_z);
            ok = isfinite(x) && isfinite(y) && isfinite(z);       // pcl::removeNaNFromPointCloud drops the row otherwise
        }
        const unsigned long long m = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid_bits[i >> 6] = m;
    }
}

// word_prefix: exclusive popcount prefix over the 32-bit words of valid_bits
static __global__ void __launch_bounds__(256) k_pack_write(const unsigned char* __restrict__ raw, uint64_t n, PointLayout L,
                                                    const uint32_t* __restrict__ valid_words,
                                                    const uint32_t* __restrict__ word_prefix, float* __restrict__ xyz_out,
                                                    uint32_t* __restrict__ n_valid) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n; i += stride) {
        const uint32_t w = valid_words[i >> 5], bit = (uint32_t)(i & 31u);
        if (!((w >> bit) & 1u)) continue;
        const uint32_t dst = word_prefix[i >> 5] + (uint32_t)__popc(w & ((1u << bit) - 1u));
        const unsigned char* rec = raw + i * L.point_step;
        xyz_out[3 * (uint64_t)dst] = load_field(rec, L.off_x);
        xyz_out[3 * (uint64_t)dst + 1] = load_field(rec, L.off_y);
        xyz_out[3 * (uint64_t)dst + 2] = load_field(rec, L.off_z);
    }
    if (gid == 0) {
        const uint64_t last = (n - 1) >> 5;
        *n_valid = n ? word_prefix[last] + (uint32_t)__popc(valid_words[last]) : 0u;
    }
}

}  // namespace gndt
