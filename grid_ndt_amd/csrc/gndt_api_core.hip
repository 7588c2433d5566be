// gndt_api_core.hip — handle life cycle, origin, sync / export, profiling, and the buffers every strategy shares.
#include "gndt_handle.hpp"

using namespace gndt;
using namespace gndt_host;

namespace {
thread_local std::string g_create_error;
}

namespace gndt_host {

const Tuning& tuning_mut_ref();
namespace {
// (Round 5: twelve knobs that had lost — or never had — an A/B became constants of gndt_host::Tuning.  Round 6: the last
//  environment reads are gone as well; the two diagnostics, AUTO's threshold and the flood's launch mode are set through
//  gndt_debug_enable_stamps / gndt_debug_set_option, process-wide.)
Tuning& tuning_storage() { static Tuning t; return t; }
}  // namespace
const Tuning& tuning() { return tuning_storage(); }
void tuning_force_stamps(bool on) { tuning_storage().stamps = on; }
void tuning_force_fp_bits(int bits) { tuning_storage().fp_bits = std::min(21, std::max(0, bits)); }
int tuning_set_option(int option, double value) {
    Tuning& t = tuning_storage();
    switch (option) {
        case GNDT_DEBUG_VERBOSE: t.verbose = value != 0.0; return GNDT_OK;
        case GNDT_DEBUG_TILE_RATIO: if (!(value >= 1.0)) return GNDT_ERR_INVALID; t.tile_ratio = value; return GNDT_OK;
        case GNDT_DEBUG_COST_ONE_WORKGROUP: t.cost_one_workgroup = value != 0.0; return GNDT_OK;
        default: return GNDT_ERR_INVALID;
    }
}

int ensure_out(gndt_handle* h, uint64_t n) {
    if (n <= h->out_cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the result arrays");
    void* ptrs[] = {h->out.sx, h->out.sy, h->out.sz, h->out.count, h->out.first_idx, h->out.mean, h->out.cov,
                    h->out.rough, h->out.normal, h->out.flags};
    for (void* p : ptrs) release_device(h, p);
    h->out = OutView{};
    h->out_cap = 0;
    uint64_t c = std::max<uint64_t>(1024, n + n / 8);
    HIP_TRY(h, hipMalloc(&h->out.sx, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.sy, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.sz, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.count, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.first_idx, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.mean, c * 12));
    HIP_TRY(h, hipMalloc(&h->out.cov, c * 24));
    HIP_TRY(h, hipMalloc(&h->out.rough, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.normal, c * 12));
    HIP_TRY(h, hipMalloc(&h->out.flags, c * 4));
    h->out_cap = c;
    return GNDT_OK;
}

int ensure_stats_buffers(gndt_handle* h, uint64_t n) {
    if (n <= h->st_cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the statistics buffers");
    void* ptrs[] = {h->st_key, h->st_sums, h->st_count, h->st_first};
    for (void* p : ptrs) release_device(h, p);
    h->st_key = nullptr; h->st_sums = nullptr; h->st_count = nullptr; h->st_first = nullptr; h->st_cap = 0;
    uint64_t c = std::max<uint64_t>(1024, n + n / 8);
    HIP_TRY(h, hipMalloc(&h->st_key, c * 8));
    HIP_TRY(h, hipMalloc(&h->st_sums, c * 72));
    HIP_TRY(h, hipMalloc(&h->st_count, c * 4));
    HIP_TRY(h, hipMalloc(&h->st_first, c * 4));
    h->st_cap = c;
    return GNDT_OK;
}

// read the device counters (synchronises the stream)
int fetch_counters(gndt_handle* h, hipStream_t s) {
    GNDT_NO_CAPTURE(h, "the device counters");
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return GNDT_OK;
}

int check_ready(gndt_handle* h) {
    if (!h) return GNDT_ERR_INVALID;
    h->capturing = false;              // (use_stream, which every call that enqueues work makes next, sets it for the call's stream)
    if (!h->origin_set) { h->err = "gndt_set_origin must be called first (setCloudFirst, receiver.cpp:145)"; return GNDT_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    return GNDT_OK;
}

void free_part(gndt_handle* h) {
    auto& q = h->part;
    void* ptrs[] = {q.recs, q.recs1, q.cursors, q.range_lo, q.range_hi, q.range_cap, q.hist, q.totals, q.bucket_base, q.stage, q.ord_cf, q.ord_idx, q.inv, q.row_of, q.row_ncol, q.raw,
                    q.bitmap, q.word_weight, q.word_base, q.bsum_words, q.ncol_at, q.d_pc, q.dbg, q.cursors_alt, q.d_pc_alt};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (q.h_pc) (void)hipHostFree(q.h_pc);
    if (q.d_extent) (void)hipFree(q.d_extent);
    if (q.h_extent) (void)hipHostFree(q.h_extent);
    q = gndt_handle::Part{};
}

int ensure_stage(gndt_handle* h, uint64_t nodes, bool rows) {
    auto& q = h->part;
    if (nodes > q.stage_cap) {
        GNDT_NO_CAPTURE(h, "the staging rows");
        void* ptrs[] = {q.ord_cf, q.ord_idx, q.inv, q.row_of};
        for (void* p : ptrs) release_device(h, p);
        q.ord_cf = q.ord_idx = q.inv = q.row_of = nullptr;
        q.stage_cap = 0;
        uint32_t** arrs[] = {&q.ord_cf, &q.ord_idx, &q.inv, &q.row_of};
        for (auto a : arrs) HIP_TRY(h, hipMalloc(a, nodes * 4));
        q.stage_cap = nodes;
        // (the column index of the result rows has its own capacity: an adopted map — gndt_adopt_rows_device — can be larger than
        // anything this handle staged)
        const int rc = grow_buf(h, q.row_ncol, q.row_ncol_cap, nodes);
        if (rc) return rc;
    }
    // StageRow records: as many as the order arrays address (the kernels that fill them are given stage_cap as their bound), but
    // only for the paths that stage such rows — a PARTITION build stages RawNode records (ensure_raw)
    if (rows && q.stage_rows_cap < q.stage_cap) {
        GNDT_NO_CAPTURE(h, "the staging rows");
        release_device(h, q.stage);
        q.stage = nullptr; q.stage_rows_cap = 0;
        HIP_TRY(h, hipMalloc(&q.stage, q.stage_cap * sizeof(StageRow)));
        q.stage_rows_cap = q.stage_cap;
    }
    return GNDT_OK;
}

// the PARTITION strategies' staging rows (RawNode).  Their number is NOT tied to the order arrays': a table-path call between two
// partition builds may grow those (ensure_stage) without the records having to follow — the bucket kernel is given the smaller of
// the two capacities (round 6, tools/fuzz_graph.py: tying them made a capture after such a call ask for memory)
int ensure_raw(gndt_handle* h, uint64_t records) {
    auto& q = h->part;
    if (records <= q.raw_cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the bucket kernel's staging rows");
    release_device(h, q.raw);
    q.raw = nullptr; q.raw_cap = 0;
    records = std::max<uint64_t>(records, std::min<uint64_t>(q.stage_cap, 2 * records));      // (in step with the order arrays where that is cheap)
    HIP_TRY(h, hipMalloc(&q.raw, records * sizeof(RawNode)));
    q.raw_cap = records;
    return GNDT_OK;
}

int ensure_words(gndt_handle* h, uint64_t words) {
    auto& q = h->part;
    if (words <= q.word_cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the column-order bitmap (raise gndt_params.max_points_hint)");
    for (uint32_t** a : {&q.bitmap, &q.word_weight, &q.word_base, &q.bsum_words, &q.ncol_at}) { release_device(h, *a); *a = nullptr; }
    q.word_cap = 0;
    words += words / 4;
    HIP_TRY(h, hipMalloc(&q.bitmap, words * 4));
    HIP_TRY(h, hipMalloc(&q.word_weight, words * 4));
    HIP_TRY(h, hipMalloc(&q.word_base, words * 4));
    HIP_TRY(h, hipMalloc(&q.bsum_words, ((words + kScanChunk - 1) / kScanChunk + 1) * 4));
    HIP_TRY(h, hipMalloc(&q.ncol_at, words * 32 * 4));     // one entry per point index, touched only at column-first indices
    q.word_cap = words;
    return GNDT_OK;
}

// Zero a fresh allocation and WAIT for it.  hipMemset() is not that: it fills on the null stream, which is not ordered with the
// non-blocking streams the builds run on, and it returns before the fill has run when the GPU is busy — the zeros then land
// seconds of GPU work later, e.g. on the overflow flags a first build has just raised (round 3: a shard built by one of six
// host threads on a shared GPU lost its "region overflow" flag in 3 % of the runs and with it the points that did not fit;
// tools/fuzz_owner.py).  Filled on the handle's own stream and awaited, nothing can follow it.
int zero_device_now(gndt_handle* h, void* p, size_t bytes) {
    GNDT_NO_CAPTURE(h, "a fresh buffer");
    HIP_TRY(h, hipMemsetAsync(p, 0, bytes, h->own_stream));
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    return GNDT_OK;
}

int ensure_part_counters(gndt_handle* h) {
    auto& q = h->part;
    if (q.d_pc) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the partition counters");
    HIP_TRY(h, hipMalloc(&q.d_pc, sizeof(PartCounters)));
    HIP_TRY(h, hipMalloc(&q.d_pc_alt, sizeof(PartCounters)));
    HIP_TRY(h, hipHostMalloc(&q.h_pc, sizeof(PartCounters)));
    { const int rc = zero_device_now(h, q.d_pc, sizeof(PartCounters)); if (rc) return rc; }
    { const int rc = zero_device_now(h, q.d_pc_alt, sizeof(PartCounters)); if (rc) return rc; }
    q.alt_clean = false;
    memset(q.h_pc, 0, sizeof(PartCounters));
    return GNDT_OK;
}

// Host input -> the handle's device staging buffer, asynchronously on stream s.
//   pinned host memory (hipHostMalloc / hipHostRegister, e.g. a pinned torch tensor): ONE async copy, the DMA engine's rate;
//   pageable memory, large: the cloud goes through two pinned bounce buffers of 8 MB — the CPU fills one while the DMA engine
//     empties the other, and the build's kernels queue behind the last chunk — instead of the runtime's own pageable path
//     (a 120 MB cloud: 150 ms measured through torch in round 3);
//   pageable, small (a LiDAR frame): the runtime's path, which is fast for a few MB.
int stage_host_input(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes, hipStream_t s) {
    const size_t bytes = n * stride_bytes;
    if (bytes > h->stage_bytes) {
        GNDT_NO_CAPTURE(h, "the host-input staging buffer");
        release_device(h, h->stage);          // (a recorded host-input build copies into it: retired, not freed, once a graph exists)
        h->stage = nullptr; h->stage_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->stage, bytes + 64));
        h->stage_bytes = bytes;
    }
    if (!bytes) return GNDT_OK;
    constexpr size_t kChunk = 8u << 20;
    bool pinned = false;
    {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, xyz_host) == hipSuccess) pinned = a.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                        // (plain malloc memory: "invalid value" on some runtimes — not an error here)
    }
    if (pinned || bytes < 2 * kChunk) {
        HIP_TRY(h, hipMemcpyAsync(h->stage, xyz_host, bytes, hipMemcpyHostToDevice, s));
        return GNDT_OK;
    }
    GNDT_NO_CAPTURE(h, "the pinned bounce buffers of a pageable host input");     // (allocated lazily and waited on below)
    for (int b = 0; b < 2; ++b) {
        if (!h->bounce[b]) HIP_TRY(h, hipHostMalloc(&h->bounce[b], kChunk));
        if (!h->bounce_ev[b]) HIP_TRY(h, hipEventCreateWithFlags(&h->bounce_ev[b], hipEventDisableTiming));
    }
    const char* src = static_cast<const char*>(xyz_host);
    char* dst = static_cast<char*>(h->stage);
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += kChunk, ++i) {
        const int b = (int)(i & 1);
        const size_t len = std::min(kChunk, bytes - off);
        // the copy that last used this buffer has left it — also the PREVIOUS call's (a call that failed between staging and its
        // sync leaves its copies in flight; an event never recorded is complete)
        HIP_TRY(h, hipEventSynchronize(h->bounce_ev[b]));
        memcpy(h->bounce[b], src + off, len);
        HIP_TRY(h, hipMemcpyAsync(dst + off, h->bounce[b], len, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipEventRecord(h->bounce_ev[b], s));
    }
    return GNDT_OK;
}


// The handle's buffers are shared by whatever was enqueued last; when the caller moves to another stream the new
// work is ordered behind it (the round-1 code reused them with no cross-stream dependency).
int use_stream(gndt_handle* h, hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    h->capturing = cap != hipStreamCaptureStatusNone;
    h->cur_capture_id = 0;
    if (h->capturing) {
        h->ever_captured = true;                         // (from now on outgrown buffers are retired, not freed: Handle::retired)
        h->cur_capture_id = ++h->capture_seq;            // this recorded call (what it is, and with which buffers, is noted when it launches)
        auto& rec = h->captures[h->cur_capture_id % 32];
        rec = gndt_handle::CaptureRec{};
        rec.id = h->cur_capture_id; rec.realloc_gen = h->realloc_gen; rec.table_gen = h->table_gen;
    }
    // (a stream under graph capture cannot wait for un-captured work: the caller orders the capture itself)
    if (h->last_stream != s && cap == hipStreamCaptureStatusNone) {     // (last_stream is always a stream: own_stream from gndt_create on)
        if (!h->xstream_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->xstream_ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->xstream_ev, h->last_stream));
        HIP_TRY(h, hipStreamWaitEvent(s, h->xstream_ev, 0));
    }
    h->last_stream = s;
    return GNDT_OK;
}

}  // namespace gndt_host

extern "C" {

const char* gndt_last_error(const gndt_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gndt_create(const gndt_params* params, gndt_handle** out) { return gndt_host::create_handle(params, out, nullptr); }

}  // extern "C"

namespace gndt_host {
int create_handle(const gndt_params* params, gndt_handle** out, hipStream_t borrowed_stream) {
    if (!params || !out) { g_create_error = "null argument"; return GNDT_ERR_INVALID; }
    *out = nullptr;
    if (!(params->grid_len > 0.f) || !(params->z_len > 0.f) || params->min_points < 1 ||
        (params->demand != GNDT_DEMAND_SLOPE && params->demand != GNDT_DEMAND_TRUE)) {
        g_create_error = "invalid gndt_params (grid_len/z_len must be > 0, demand 0|1, min_points >= 1)";
        return GNDT_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (libgndt has no CPU path)";
        return GNDT_ERR_NO_DEVICE;
    }
    if (params->device_id < 0 || params->device_id >= ndev) { g_create_error = "device_id out of range"; return GNDT_ERR_INVALID; }
    gndt_handle* h = new (std::nothrow) gndt_handle;
    if (!h) { g_create_error = "out of host memory"; return GNDT_ERR_NOMEM; }
    h->P = *params;
    h->device = params->device_id;
    auto fail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        gndt_destroy(h);
        return GNDT_ERR_HIP;
    };
    if ((e = hipSetDevice(h->device)) != hipSuccess) return fail("hipSetDevice", e);
    if (borrowed_stream) { h->own_stream = borrowed_stream; h->stream_borrowed = true; }
    else if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipMalloc(&h->d_cnt, sizeof(Counters))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc(&h->h_cnt, sizeof(Counters))) != hipSuccess) return fail("hipHostMalloc", e);
    if ((e = hipMemsetAsync(h->d_cnt, 0, sizeof(Counters), h->own_stream)) != hipSuccess) return fail("hipMemsetAsync", e);   // (not hipMemset: zero_device_now)
    if ((e = hipStreamSynchronize(h->own_stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    memset(h->h_cnt, 0, sizeof(Counters));
    // (the sharded builds' agreement words: 8 KB each — see Exchange::d_agree)
    if ((e = hipMalloc(&h->exch.d_agree, (size_t)(1024 + 1) * sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc(&h->exch.h_agree, (size_t)(1024 + 1) * sizeof(unsigned long long))) != hipSuccess) return fail("hipHostMalloc", e);
    h->last_stream = h->own_stream;
    if (params->max_nodes_hint) {
        int rc = alloc_table(h, cap_for_nodes(params->max_nodes_hint), h->own_stream);
        if (rc) { g_create_error = h->err; gndt_destroy(h); return rc; }
        if ((e = hipStreamSynchronize(h->own_stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    }
    h->part.load_pct = tuning().bucket_load;
    *out = h;
    return GNDT_OK;
}
}  // namespace gndt_host

extern "C" {

void gndt_destroy(gndt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_table(h);
    free_part(h);
    free_cost(h);
    void* ptrs[] = {h->out.sx, h->out.sy, h->out.sz, h->out.count, h->out.first_idx, h->out.mean, h->out.cov,
                    h->out.rough, h->out.normal, h->out.flags, h->st_key, h->st_sums, h->st_count, h->st_first,
                    h->stage, h->d_cnt, h->packed, h->d_nvalid};
    if (h->h_nvalid) (void)hipHostFree(h->h_nvalid);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& set : h->ev)
        for (auto& e : set)
            if (e) (void)hipEventDestroy(e);
    if (h->h_cnt) (void)hipHostFree(h->h_cnt);
    {
        auto& X = h->exch;
        void* xp[] = {X.d_counts, X.keys_in, X.keys_all, X.keys_sorted, X.canon, X.d_unique, X.d_missing, X.scratch, X.packed, X.pfirst, X.r_sums, X.r_count, X.red_tmp,
                      X.send_recs, X.own_recs, X.d_matrix, X.d_split_cnt, X.pairs, X.pairs_all, X.d_npairs, X.global_row, X.d_totals,
                      X.owner_msg, X.owner_msgs_all, X.bkey, X.bcnt, X.bown, X.d_owner_full, X.d_colmsg, X.gw, X.grec, X.grec_all, X.d_tally, X.d_status, X.row_of_pair, X.place_all, X.place_mine, X.d_slice};
        for (void* p : xp) if (p) (void)hipFree(p);
        if (X.h_counts) (void)hipHostFree(X.h_counts);
        if (X.h_matrix) (void)hipHostFree(X.h_matrix);
        if (X.h_split_cnt) (void)hipHostFree(X.h_split_cnt);
        if (X.h_totals) (void)hipHostFree(X.h_totals);
        if (X.h_colmsg) (void)hipHostFree(X.h_colmsg);
        if (X.h_bad) (void)hipHostFree(X.h_bad);
        if (X.d_agree) (void)hipFree(X.d_agree);
        if (X.h_agree) (void)hipHostFree(X.h_agree);
        if (X.h_tally) (void)hipHostFree(X.h_tally);
        for (auto& e : X.ev) if (e) (void)hipEventDestroy(e);
    }
    if (h->xstream_ev) (void)hipEventDestroy(h->xstream_ev);
    if (h->exp_host) (void)hipHostFree(h->exp_host);
    for (int b = 0; b < 2; ++b) { if (h->bounce[b]) (void)hipHostFree(h->bounce[b]); if (h->bounce_ev[b]) (void)hipEventDestroy(h->bounce_ev[b]); }
    for (void* p : h->retired) (void)hipFree(p);
    if (h->d_sketch) (void)hipFree(h->d_sketch);
    if (h->h_sketch) (void)hipHostFree(h->h_sketch);
    if (h->sample_ev) (void)hipEventDestroy(h->sample_ev);
    if (h->d_sample) (void)hipFree(h->d_sample);
    if (h->h_sample) (void)hipHostFree(h->h_sample);
    if (h->own_stream && !h->stream_borrowed) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int gndt_set_origin(gndt_handle* h, const float origin_xyz[3]) {
    if (!h || !origin_xyz) return GNDT_ERR_INVALID;
    // a build still pending belongs to the OLD origin: settle it (flags, retries) before the origin moves
    if (h->pending.active) { const int prc = partition_resolve(h); if (prc) return prc; }
    if (h->table_dirty) { h->err = "origin cannot change while the map holds points (call gndt_reset)"; return GNDT_ERR_INVALID; }
    if (memcmp(h->origin, origin_xyz, 3 * sizeof(float)) != 0) h->part.blk_state = 0;      // (a box of blocked buckets belongs to the keys of the old origin)
    memcpy(h->origin, origin_xyz, 3 * sizeof(float));
    h->origin_set = true;
    return GNDT_OK;
}

int gndt_get_origin(const gndt_handle* h, float origin_xyz[3]) {
    if (!h || !origin_xyz || !h->origin_set) return GNDT_ERR_INVALID;
    memcpy(origin_xyz, h->origin, 3 * sizeof(float));
    return GNDT_OK;
}

int gndt_reset(gndt_handle* h, void* hip_stream) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    h->pending.active = false;
    h->map_in_table = true;
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    return do_reset(h, s);
}
// Called with the stream idle.  The flags' host mirror names the recorded call that produced it (PartCounters::capture_id; 0 = a
// call the host launched).  A replay is work the host did not see: if its buffers have been reallocated since the recording, what
// it wrote is not in the handle's buffers (GNDT_ERR_CAPACITY: capture again); if it is a PARTITION build and the host's last own
// build went through the node table (or the other way round), the handle's idea of what it holds is brought in line first.
static int replay_check(gndt_handle* h) {
    auto& q = h->part;
    if (!q.h_pc || h->pending.active) return GNDT_OK;
    const uint32_t cid = q.h_pc->capture_id;
    h->replay_seen = cid;
    if (!cid) return GNDT_OK;
    const auto& rec = h->captures[cid % 32];
    const bool known = rec.id == cid;
    const bool moved = !known || rec.realloc_gen != h->realloc_gen || (!rec.partition && rec.table_gen != h->table_gen);
    if (moved) {
        h->results_valid = false;
        h->pending.replay_failed = true;
        h->err = "a hipGraph replay ran after buffers of this handle had been reallocated (an eager call of another size since the recording): "
                 "it wrote through the old pointers — run the call eagerly once (or gndt_reserve), then capture again";
        return GNDT_ERR_CAPACITY;
    }
    // What the device holds is a finished run of the recorded call, with the handle's own buffers: it can be read again even if an
    // earlier replay was reported (too many nodes for what was recorded: the flags are looked at again right after this, and say so
    // again if THIS replay does not fit either).  Until round 5 one reported replay left every later one "no finished build".
    if (rec.partition) { if (!h->results_valid) h->pending.replay_failed = true; }
    else h->results_valid = true;
    if (rec.partition && h->map_in_table) {            // a PARTITION build replayed after an eager table-path call
        auto& P = h->pending;
        h->map_in_table = false;
        h->last_strategy = rec.two_level ? GNDT_STRATEGY_PARTITION : (rec.one_level ? GNDT_STRATEGY_PARTITION_ONE_LEVEL : GNDT_STRATEGY_PARTITION_EXACT);
        h->incr_ok = false;
        h->emit_pending = false;
        h->results_valid = true;
        ++h->result_serial;
        P = gndt_handle::Pending{};                      // (what is known of the recorded build: enough for the flags' check that follows)
        P.captured = true; P.two_level = rec.two_level; P.one_level = rec.one_level;
        P.done_serial = h->result_serial;
    } else if (!rec.partition && !h->map_in_table) {   // a table-path call replayed after an eager PARTITION build
        h->map_in_table = true;
        h->last_strategy = GNDT_STRATEGY_ATOMIC;
        h->incr_ok = false;                              // (the next update relabels every column)
        h->table_dirty = true;
        h->results_valid = true;
        ++h->result_serial;
        h->pending.done_serial = 0;
    }
    return GNDT_OK;
}

int gndt_sync(gndt_handle* h, uint64_t* num_nodes, uint64_t* num_columns, uint64_t* num_slopes) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    { const int prc = partition_resolve(h); if (prc) return prc; }
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    { const int crc = replay_check(h); if (crc) return crc; }                          // (what the device holds may be a replay the host did not see)
    { const int rrc = partition_recheck_after_replay(h); if (rrc) return rrc; }      // (a replayed build that ran out of room says so)
    if (h->results_valid && (h->emit_pending || h->deferred_captured) && h->map_in_table) {      // deferred-emit mode: the rows are produced now, for all the frames since the last read
        const int erc = table_emit_pending(h);
        if (erc) return erc;
    }
    // (asked of the DEVICE's flag and of the id in the mirror, not of what the host's last own call used: round 5, tools/fuzz_graph.py —
    //  a recorded small-map build replayed on a larger cloud right after an eager full finalisation came back as a map of 0 columns)
    if (h->results_valid && h->part.h_pc && h->part.h_pc->small_fallback && (h->small_used || h->replay_seen)) {
        // the one-workgroup finalisation (k_small_finalize) met a map that is not small and wrote nothing
        if (h->replay_seen || h->small_captured) {
            h->results_valid = false;
            h->err = "a build replayed from a hipGraph has more nodes than the small-map finalisation it was captured with can hold: "
                     "build this cloud eagerly (gndt_build_device + gndt_sync), then capture again";
            return GNDT_ERR_CAPACITY;
        }
        const int frc = table_refinalize(h);
        if (frc) return frc;
    }
    if (h->results_valid) {
        h->res_nodes = h->h_cnt->num_nodes;
        h->res_columns = h->h_cnt->num_columns;
        h->res_slopes = h->h_cnt->num_slopes;
    }
    if (num_nodes) *num_nodes = h->res_nodes;
    if (num_columns) *num_columns = h->res_columns;
    if (num_slopes) *num_slopes = h->res_slopes;
    if (h->h_cnt->err_key_range) {
        h->err = std::to_string(h->h_cnt->err_key_range) +
                 " point(s) outside the key range (|nx|,|ny| <= 65535: countMorton wraps beyond, Stopwatch.h:102-110)";
        return GNDT_ERR_KEY_RANGE;
    }
    if (h->h_cnt->err_table_full) { h->err = "node table full: raise gndt_params.max_nodes_hint"; return GNDT_ERR_CAPACITY; }
    if (h->results_valid && h->part.h_pc && h->part.h_pc->stage_overflow) {
        h->err = "more nodes than result rows: raise gndt_params.max_nodes_hint";
        return GNDT_ERR_CAPACITY;
    }
    if (h->results_valid && h->part.h_pc && h->part.h_pc->index_overflow) {
        h->err = "point stream ran past the column-order bitmap (replayed graph?): raise gndt_params.max_points_hint";
        return GNDT_ERR_CAPACITY;
    }
    return GNDT_OK;
}

int gndt_export_device(gndt_handle* h, gndt_cells* out) {
    if (!h || !out) return GNDT_ERR_INVALID;
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to export"; return GNDT_ERR_INVALID; }
    int rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    out->num_nodes = h->res_nodes; out->num_columns = h->res_columns; out->num_slopes = h->res_slopes;
    out->sx = h->out.sx; out->sy = h->out.sy; out->sz = h->out.sz;
    out->count = h->out.count; out->first_idx = h->out.first_idx;
    out->mean = h->out.mean; out->cov = h->out.cov; out->rough = h->out.rough; out->normal = h->out.normal;
    out->flags = h->out.flags;
    return GNDT_OK;
}

int gndt_export(gndt_handle* h, gndt_cells* o) {
    if (!h || !o) return GNDT_ERR_INVALID;
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to export"; return GNDT_ERR_INVALID; }
    int rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    const uint64_t n = h->res_nodes;
    o->num_nodes = n; o->num_columns = h->res_columns; o->num_slopes = h->res_slopes;
    struct { void* dst; const void* src; size_t elem; } copies[] = {
        {o->sx, h->out.sx, 4}, {o->sy, h->out.sy, 4}, {o->sz, h->out.sz, 4}, {o->count, h->out.count, 4},
        {o->first_idx, h->out.first_idx, 4}, {o->mean, h->out.mean, 12}, {o->cov, h->out.cov, 24},
        {o->rough, h->out.rough, 4}, {o->normal, h->out.normal, 12}, {o->flags, h->out.flags, 4}};
    for (auto& c : copies)
        if (c.dst && n) HIP_TRY(h, hipMemcpy(c.dst, c.src, n * c.elem, hipMemcpyDeviceToHost));
    return GNDT_OK;
}

int gndt_export_host(gndt_handle* h, gndt_cells* o) {
    if (!h || !o) return GNDT_ERR_INVALID;
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to export"; return GNDT_ERR_INVALID; }
    int rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    const uint64_t n = h->res_nodes;
    const uint64_t rows = std::max<uint64_t>(n, 1);
    if (rows > h->exp_rows) {
        if (h->exp_host) (void)hipHostFree(h->exp_host);
        h->exp_host = nullptr; h->exp_rows = 0;
        const uint64_t want = rows + rows / 4;
        HIP_TRY(h, hipHostMalloc(&h->exp_host, want * 76 + 256));
        h->exp_rows = want;
    }
    // ten arrays, one after the other in the pinned mirror (every one of them 4-byte elements: no alignment gaps needed)
    char* base = static_cast<char*>(h->exp_host);
    const uint64_t cap = h->exp_rows;
    struct { const void* src; size_t elem; void** dst; } copies[] = {
        {h->out.sx, 4, (void**)&o->sx}, {h->out.sy, 4, (void**)&o->sy}, {h->out.sz, 4, (void**)&o->sz}, {h->out.count, 4, (void**)&o->count},
        {h->out.first_idx, 4, (void**)&o->first_idx}, {h->out.mean, 12, (void**)&o->mean}, {h->out.cov, 24, (void**)&o->cov},
        {h->out.rough, 4, (void**)&o->rough}, {h->out.normal, 12, (void**)&o->normal}, {h->out.flags, 4, (void**)&o->flags}};
    size_t off = 0;
    for (auto& c : copies) {
        *c.dst = base + off;
        if (n) HIP_TRY(h, hipMemcpyAsync(base + off, c.src, n * c.elem, hipMemcpyDeviceToHost, h->last_stream));
        off += cap * c.elem;
    }
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    o->num_nodes = n; o->num_columns = h->res_columns; o->num_slopes = h->res_slopes;
    return GNDT_OK;
}

int gndt_set_profiling(gndt_handle* h, int enable) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (enable)
        for (auto& set : h->ev)
            for (auto& e : set)
                if (!e) HIP_TRY(h, hipEventCreate(&e));
    for (auto& set : h->ev_recorded)
        for (auto& r : set) r = false;
    h->prof = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return GNDT_OK;
}

// Mean duration of every phase over the builds recorded since the last call (at most the last kEvSets).
int gndt_get_phase_times(gndt_handle* h, double ms_out[GNDT_NUM_PHASES]) {
    if (!h || !ms_out) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < GNDT_NUM_PHASES; ++i) {
        double sum = 0.0;
        int cnt = 0;
        for (int k = 0; k < gndt_handle::kEvSets; ++k) {
            if (!(h->ev_recorded[k][i] && h->ev_recorded[k][i + 1])) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[k][i], h->ev[k][i + 1]) == hipSuccess) { sum += ms; ++cnt; }
        }
        ms_out[i] = cnt ? sum / cnt : -1.0;
    }
    for (auto& set : h->ev_recorded)
        for (auto& r : set) r = false;
    return GNDT_OK;
}
int gndt_last_strategy(const gndt_handle* h) { return h ? h->last_strategy : GNDT_STRATEGY_AUTO; }

int gndt_device_info(int32_t device_id, char name_out[128], int32_t* compute_units, uint64_t* hbm_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return GNDT_ERR_NO_DEVICE;
    if (name_out) {
        name_out[0] = 0;
        (void)hipDeviceGetName(name_out, 128, device_id);
    }
    if (compute_units) {
        int cu = 0;
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id);
        *compute_units = cu;
    }
    if (hbm_bytes) {
        size_t total = 0;
        (void)hipDeviceTotalMem(&total, device_id);
        *hbm_bytes = total;
    }
    return GNDT_OK;
}

}  // extern "C"
