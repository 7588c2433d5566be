// gndt_api_dist.hip — one global map from a sharded cloud: shard -> statistics (PARTITION pipeline, no node table), and merged statistics (sorted by key) -> map.
#include "gndt_handle.hpp"
#include "gndt_table.hpp"
using namespace gndt;
using namespace gndt_host;

extern "C" {

int gndt_shard_stats_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint64_t first_idx_base,
                            gndt_stats* out, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!out || (!xyz_dev && n)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    if (first_idx_base + n >= 0x7FFFFFFFull) { h->err = "point index exceeds 31 bits"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    next_event_set(h);
    rc = -1;
    if (n >= (1u << 12) && h->P.strategy != GNDT_STRATEGY_ATOMIC && stride_bytes != 0) {
        auto& q = h->part;
        if ((rc = ensure_words(h, (n + 31) / 32 + 1))) return rc;
        if ((rc = ensure_part_counters(h))) return rc;
        auto& P = h->pending;
        P = gndt_handle::Pending{};
        P.xyz = xyz_dev; P.n = n; P.stride = stride_bytes; P.s = s;
        P.stats_only = true; P.first_base = (uint32_t)first_idx_base;
        P.gp = grid_params(h);
        P.nodes_est = h->P.max_nodes_hint ? h->P.max_nodes_hint : (q.nodes_learned ? q.nodes_learned : std::max<uint64_t>(n / 4, 1024));
        P.stage_want = std::max<uint64_t>(h->st_cap, std::max<uint64_t>(4096, n / 4));
        h->results_valid = false;
        const int prev = h->last_strategy;
        h->last_strategy = GNDT_STRATEGY_PARTITION;
        rc = partition_launch(h, P);
        if (rc == GNDT_OK) { P.active = true; rc = partition_resolve(h); }
        if (rc != -1) {
            if (rc) { h->last_strategy = prev; return rc; }
            if (h->h_cnt->err_key_range) {
                h->err = std::to_string(h->h_cnt->err_key_range) + " point(s) outside the key range";
                return GNDT_ERR_KEY_RANGE;
            }
            out->num_nodes = h->h_cnt->num_nodes;
            out->key = h->st_key; out->sums = h->st_sums; out->count = h->st_count; out->first_idx = h->st_first;
            return GNDT_OK;
        }
        h->last_strategy = prev;
    }
    // small shards, strategy ATOMIC, or too many nodes per bucket: through the node table
    h->pending.active = false;
    if ((rc = gndt_reset(h, hip_stream))) return rc;
    if ((rc = gndt_accumulate_device(h, xyz_dev, n, stride_bytes, first_idx_base, hip_stream))) return rc;
    return gndt_stats_export_device(h, out, hip_stream);
}

int gndt_finalize_stats_device(gndt_handle* h, const gndt_stats* in, uint64_t total_points, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!in || (in->num_nodes && (!in->key || !in->sums || !in->count || !in->first_idx))) { h->err = "null statistics"; return GNDT_ERR_INVALID; }
    if (in->num_nodes >= 0xFFFFFFFFull || total_points >= 0xFFFFFFFFull) { h->err = "too many nodes / points"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    next_event_set(h);
    auto& q = h->part;
    const uint64_t n = in->num_nodes;
    const uint64_t words = (std::max<uint64_t>(total_points, 64) + 31) / 32 + 1;
    if ((rc = ensure_part_counters(h))) return rc;
    if ((rc = ensure_stage(h, std::max<uint64_t>(1024, n)))) return rc;
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    if ((rc = ensure_words(h, words))) return rc;
    if (h->table_dirty) { if ((rc = do_reset(h, s))) return rc; }
    h->results_valid = false;
    mark(h, 0, s);
    hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                       (uint32_t*)nullptr, 0u);
    HIP_TRY(h, hipGetLastError());
    mark(h, 4, s);
    hipLaunchKernelGGL(k_stats_rows, dim3(grid_for(std::max<uint64_t>(n, 1))), dim3(kBlock), 0, s, in->key, in->sums, in->count,
                       in->first_idx, (uint32_t)n, grid_params(h), q.stage, q.ord_cf, q.ord_idx, ColumnOrder{q.bitmap, q.word_weight, q.ncol_at},
                       (uint64_t)words, h->d_cnt, q.d_pc);
    HIP_TRY(h, hipGetLastError());
    mark(h, 5, s);
    if ((rc = launch_order_and_emit(h, words, 5, s))) return rc;
    HIP_TRY(h, hipMemcpyAsync(q.h_pc, q.d_pc, sizeof(PartCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    h->results_valid = true;
    ++h->result_serial;
    h->map_in_table = false;
    h->last_strategy = GNDT_STRATEGY_PARTITION;
    h->stream_pos = total_points;
    return GNDT_OK;
}

}  // extern "C"
