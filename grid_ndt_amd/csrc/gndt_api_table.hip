// gndt_api_table.hip — strategy ATOMIC: the HBM node table (accumulate, incremental update, finalize, statistics export / merge).
#include "gndt_handle.hpp"
#include "gndt_table.hpp"
#include "gndt_tile.hpp"
using namespace gndt;
using namespace gndt_host;

namespace gndt_host {

void free_table(gndt_handle* h) {
    void* ptrs[] = {h->keys, h->acc, h->col_keys, h->col_first, h->aux, h->node_slot, h->col_slot_of_node,
                    h->col_cnt, h->col_head, h->node_next, h->ninfo, h->index_of_slot, h->touch_epoch, h->col_epoch, h->touched,
                    h->touched_cols};
    for (void* p : ptrs) release_device(h, p, false);
    h->keys = nullptr; h->acc = nullptr; h->col_keys = nullptr; h->col_first = nullptr; h->aux = nullptr;
    h->node_slot = nullptr; h->col_slot_of_node = nullptr; h->col_cnt = nullptr; h->col_head = nullptr; h->node_next = nullptr; h->ninfo = nullptr;
    h->index_of_slot = h->touch_epoch = h->col_epoch = h->touched = h->touched_cols = nullptr;
    h->incr_ok = false;
    h->cap = 0;
}

// A fresh, empty table of `cap` slots.  The device counters are NOT touched: the caller decides (reset vs growth).
int alloc_table(gndt_handle* h, uint32_t cap, hipStream_t s) {
    GNDT_NO_CAPTURE(h, "the node table");
    free_table(h);
    HIP_TRY(h, hipMalloc(&h->keys, (size_t)cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->acc, (size_t)cap * sizeof(NodeAcc)));
    HIP_TRY(h, hipMalloc(&h->col_keys, (size_t)cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->col_first, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_cnt, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_head, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->aux, (size_t)cap * sizeof(SlotAux)));
    HIP_TRY(h, hipMalloc(&h->node_slot, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_slot_of_node, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->node_next, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->ninfo, (size_t)cap * sizeof(NodeInfo)));
    for (uint32_t** a : {&h->index_of_slot, &h->touch_epoch, &h->col_epoch, &h->touched, &h->touched_cols})
        HIP_TRY(h, hipMalloc(a, (size_t)cap * sizeof(uint32_t)));
    h->cap = cap;
    ++h->table_gen;
    hipLaunchKernelGGL(k_clear_all, dim3(grid_for(cap)), dim3(kBlock), 0, s, h->keys, h->acc, h->col_keys,
                       h->col_first, h->col_cnt, h->col_head, h->touch_epoch, h->col_epoch, cap, h->d_cnt, h->table_gen);
    HIP_TRY(h, hipGetLastError());
    h->table_dirty = false;
    return GNDT_OK;
}

namespace {
TableView table_view(const gndt_handle* h) {
    TableView T;
    T.keys = h->keys; T.acc = h->acc; T.aux = h->aux; T.col_keys = h->col_keys; T.col_first = h->col_first;
    T.col_cnt = h->col_cnt; T.col_head = h->col_head; T.node_slot = h->node_slot; T.col_slot_of_node = h->col_slot_of_node;
    T.node_next = h->node_next; T.ninfo = static_cast<NodeInfo*>(h->ninfo); T.cap_mask = h->cap - 1;
    T.index_of_slot = h->index_of_slot; T.touch_epoch = h->touch_epoch; T.col_epoch = h->col_epoch;
    T.touched = h->touched; T.touched_cols = h->touched_cols;
    return T;
}

}  // namespace

int do_reset(gndt_handle* h, hipStream_t s) {
    // A reset recorded into a hipGraph must be the one that CLEARS: at capture time a fresh table is clean and the cheap variant
    // would be recorded, but the second replay finds the first one's nodes in it (round 4: a build captured on a reserved fresh
    // handle came back from its second replay with only the nodes the first cloud did not have).  On a clean table the clearing
    // kernel finds an empty node list and only zeroes the counters.
    // ... and once a call of this handle has been recorded into a hipGraph the host's idea of the table (table_dirty) is only a
    // guess — it is updated while a call is RECORDED, as if the kernels had run, and says nothing of replays — so the clearing
    // kernel always runs and decides on the device: the listed slots if the counters are the table's, the whole table if a stale
    // reset left it unclean, nothing if it is empty.  (Round 5, tools/fuzz_graph.py under -DGNDT_POISON: an eager reset right after
    // a capture took the one-thread path, the points that followed met the nodes of the map before — a memory fault in a process
    // that had freed and reallocated for a while, silently merged statistics in a fresh one.)
    if (h->cap && (h->table_dirty || h->capturing || h->ever_captured))       // (its last workgroup zeroes the counters)
        hipLaunchKernelGGL(k_clear_used, dim3(grid_for(h->cap / 8)), dim3(kBlock), 0, s, h->keys, h->acc, h->col_keys,
                           h->col_first, h->col_cnt, h->col_head, h->node_slot, h->col_slot_of_node, h->d_cnt, h->cap, h->table_gen);
    else
        hipLaunchKernelGGL(k_zero_counters, dim3(1), dim3(64), 0, s, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    h->table_dirty = false;
    h->results_valid = false;
    h->stream_pos = 0;
    h->nodes_bound = 0;
    h->incr_ok = false;
    // (deferred-emit frames that nobody read belong to the map that has just been cleared)
    h->emit_pending = false; h->pending_words = 0; h->deferred_captured = false;
    return GNDT_OK;
}

namespace {
// Grow the table to `new_cap` slots keeping its contents (export -> fresh table -> merge).
int grow_table(gndt_handle* h, uint32_t new_cap, hipStream_t s);

// `base_from_device`: first_idx base = the device-side stream position (incremental updates)
// `defer_advance`: the device-side stream position is advanced by the finalisation that follows (k_emit_rows), not here
int do_accumulate(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint64_t first_base,
                  int base_from_device, hipStream_t s, int mark = 0, bool tile = false, bool defer_advance = false, bool records = false) {
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (records && stride_bytes != 16) { h->err = "records are 16 bytes"; return GNDT_ERR_INVALID; }
    if (!records && first_base + n >= 0xFFFFFFFFull) { h->err = "point index exceeds 32 bits"; return GNDT_ERR_INVALID; }
    if (n == 0) return GNDT_OK;
    const float* p = static_cast<const float*>(xyz_dev);
    const int blocks = grid_for(n, kBlock, 256 * 16);
    if (records) {       // {x, y, z, index word}: indices and weights as the records carry them (first_base is not used)
        hipLaunchKernelGGL((k_accumulate<4, true>), dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, 0u, 0, grid_params(h), h->keys, h->acc,
                           h->cap - 1, h->node_slot, h->index_of_slot, h->touch_epoch, h->touched, mark, h->d_cnt);
    } else if (tile) {
        // strategy TILE: contiguous ranges, LDS-privatised partials (gndt_tile.hpp); two resident workgroups per CU
        const int wgs = (int)std::min<uint64_t>(512, (n + kTileCheck - 1) / kTileCheck);
        if (stride_bytes == 12)
            hipLaunchKernelGGL(k_tile_accumulate<3>, dim3(wgs), dim3(kTileT), 0, s, p, (uint64_t)n, (uint32_t)first_base, base_from_device,
                               grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot, h->touch_epoch, h->touched, mark,
                               h->d_cnt);
        else
            hipLaunchKernelGGL(k_tile_accumulate<4>, dim3(wgs), dim3(kTileT), 0, s, p, (uint64_t)n, (uint32_t)first_base, base_from_device,
                               grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot, h->touch_epoch, h->touched, mark,
                               h->d_cnt);
    } else if (stride_bytes == 12)
        hipLaunchKernelGGL(k_accumulate<3>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, (uint32_t)first_base,
                           base_from_device, grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot,
                           h->touch_epoch, h->touched, mark, h->d_cnt);
    else
        hipLaunchKernelGGL(k_accumulate<4>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, (uint32_t)first_base,
                           base_from_device, grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot,
                           h->touch_epoch, h->touched, mark, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    if (!mark) h->incr_ok = false;                    // nodes changed without being listed: the next finalisation redoes every column
    if (base_from_device && !defer_advance) {
        hipLaunchKernelGGL(k_advance_stream, dim3(1), dim3(64), 0, s, h->d_cnt, (uint32_t)n);
        HIP_TRY(h, hipGetLastError());
    }
    h->table_dirty = true;
    h->results_valid = false;
    h->nodes_bound = std::min<uint64_t>(h->nodes_bound + n, h->cap);
    return GNDT_OK;
}

// columns -> labels + staging rows -> ordering -> emit.  Everything is sized on the device; nothing waits for
// the host, so accumulate + finalize can be captured in a hipGraph once the buffers exist.
// `advance`: points by which the device-side stream position moves at the end of this finalisation (gndt_update_device)
// `raise_to`: the stream position is at least this from here on (a batch build: its point count)
int do_finalize(gndt_handle* h, hipStream_t s, bool incremental = false, uint64_t touched_bound = 0, uint32_t advance = 0, uint32_t raise_to = 0) {
    auto& q = h->part;
    int rc;
    // host-side upper bounds only: rows <= slots/2 at a healthy load; points seen so far (or the caller's hint)
    const uint64_t rows_bound = std::max<uint64_t>(1024, h->cap / 2 + 1);
    const uint64_t pts_bound = std::max<uint64_t>(std::max<uint64_t>(h->stream_pos, h->P.max_points_hint), 64);
    const uint64_t words = (pts_bound + 31) / 32 + 1;
    // The incremental form needs what the last finalisation left behind (staging rows, order keys, column order): any
    // reallocation, or anything else that touched those buffers, sends this call down the full path.
    if (!h->incr_ok || rows_bound > q.stage_cap || words > q.word_cap || !q.d_pc) incremental = false;
    if ((rc = ensure_part_counters(h))) return rc;
    if ((rc = ensure_stage(h, rows_bound))) return rc;
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    if ((rc = ensure_words(h, words))) return rc;
    const TableView T = table_view(h);
    const GridParams gp = grid_params(h);
    const ColumnOrder O{q.bitmap, q.word_weight, q.ncol_at};
    if (h->cur_capture_id) {                       // (recorded: a table-path call, with the buffers as they are now)
        auto& rec = h->captures[h->cur_capture_id % 32];
        rec.partition = false; rec.realloc_gen = h->realloc_gen; rec.table_gen = h->table_gen;
    }
    mark(h, 2, s);
    // Small maps — a few hundred nodes: a depth-camera frame at the launch cells — are finalised by ONE workgroup in ONE launch
    // (k_small_finalize) instead of the six kernels below.  The host goes by what the last resolved build of this handle had; a
    // map that turns out larger raises PartCounters::small_fallback and gndt_sync runs the regular path (table_refinalize).
    h->small_used = false;
    // (not while deferred frames wait for their rows: the read that follows would run the ordering pass over staging rows and
    //  order arrays this kernel does not write, and overwrite its rows)
    if (!incremental && advance == 0 && h->small_ok && h->table_nodes_seen && h->table_nodes_seen <= 900u && !h->emit_pending && !h->deferred_captured) {
        hipLaunchKernelGGL(k_small_finalize, dim3(1), dim3(kSmallMapNodes), 0, s, T, gp, h->out, q.row_ncol, h->d_cnt, q.d_pc, h->h_cnt, q.h_pc,
                           raise_to, (uint32_t)std::min<uint64_t>(h->out_cap, 0xFFFFFFFFull), h->cur_capture_id);
        HIP_TRY(h, hipGetLastError());
        for (int i = 3; i <= 9; ++i) mark(h, i, s);
        h->small_used = true;
        h->small_captured = h->capturing;
        h->results_valid = true;
        ++h->result_serial;
        h->last_stream = s;
        h->incr_ok = false;                  // (no staging rows, no order arrays: an update after this takes the full path)
        return GNDT_OK;
    }
    if (incremental) {
        if (words > q.words_init) {                        // the stream grew past the words the order has seen: they start empty
            HIP_TRY(h, hipMemsetAsync(q.bitmap + q.words_init, 0, (words - q.words_init) * 4, s));
            HIP_TRY(h, hipMemsetAsync(q.word_weight + q.words_init, 0, (words - q.words_init) * 4, s));
            q.words_init = words;
        }
        const uint64_t tb = std::max<uint64_t>(std::min<uint64_t>(touched_bound, rows_bound), 64);
        hipLaunchKernelGGL(k_tab_touch, dim3(grid_for(tb)), dim3(kBlock), 0, s, T, gp, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_tab_expand, dim3(grid_for(tb)), dim3(kBlock), 0, s, T, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        mark(h, 3, s);
        hipLaunchKernelGGL(k_tab_rows_touched, dim3(grid_for(4 * tb, kBlock, 4096)), dim3(kBlock), 0, s, T, gp, q.stage,
                           (uint32_t)q.stage_cap, q.ord_cf, q.ord_idx, O, (uint64_t)words, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_tab_begin, dim3(grid_for(std::max<uint64_t>(words, h->cap / 4), kBlock, 1024)), dim3(kBlock), 0, s, T,
                           h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words, raise_to);
        hipLaunchKernelGGL(k_tab_columns, dim3(grid_for(rows_bound)), dim3(kBlock), 0, s, T, gp, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        mark(h, 3, s);
        hipLaunchKernelGGL(k_tab_rows, dim3(grid_for(rows_bound)), dim3(kBlock), 0, s, T, gp, q.stage, (uint32_t)q.stage_cap,
                           q.ord_cf, q.ord_idx, O, (uint64_t)words, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        q.words_init = words;
    }
    mark(h, 4, s);
    if (h->defer_emit && incremental) {
        // Deferred-emit mode: the frame ends here — touched columns relabelled, their staging rows and the column order up to date,
        // the bookkeeping done by a one-thread launch.  The ordering + emit pass (O(map): rows move when a column in front of them
        // grows) runs when somebody reads the map (gndt_sync -> table_emit_pending).  A frame costs O(touched).
        hipLaunchKernelGGL(k_tab_end, dim3(1), dim3(64), 0, s, h->d_cnt, (const PartCounters*)q.d_pc, h->h_cnt, q.h_pc, advance, h->cur_capture_id);
        HIP_TRY(h, hipGetLastError());
        for (int i = 5; i <= 9; ++i) mark(h, i, s);
        h->emit_pending = true;
        // STICKY: a deferred frame recorded into a hipGraph may be replayed at any time later, unseen by the host, so every read from
        // here on emits — an eager frame in between must not clear it (replay, eager frame, sync, replay, export returned
        // stale rows with GNDT_OK).  Cleared by a reset and by gndt_set_deferred_emit(0).
        h->deferred_captured = h->deferred_captured || h->capturing;
        h->pending_words = std::max(h->pending_words, words);
        h->results_valid = true;
        ++h->result_serial;
        h->last_stream = s;
        h->incr_ok = true;
        return GNDT_OK;
    }
    // (k_emit_rows, the last kernel, also stores the counters and flags into the host's pinned mirrors and does the end-of-frame
    //  bookkeeping: no copy commands and no one-thread launches behind a frame)
    // (rows left un-emitted by deferred frames: everything is placed and emitted again, not only what this frame moved)
    if ((rc = launch_order_and_emit(h, words, 4, s, false, true, true, advance, incremental && !h->emit_pending && !h->deferred_captured))) return rc;
    h->emit_pending = false;
    h->pending_words = h->deferred_captured ? std::max(h->pending_words, words) : 0;      // (a captured deferred frame may still be replayed: reads keep emitting)
    h->results_valid = true;
    ++h->result_serial;
    h->last_stream = s;
    h->incr_ok = true;
    return GNDT_OK;
}

uint64_t expected_nodes_for_batch(const gndt_handle* h, uint64_t known_nodes, uint64_t n) {
    if (h->P.max_nodes_hint) return std::max<uint64_t>(h->P.max_nodes_hint, known_nodes);
    return known_nodes + n;   // worst case: every point opens a node
}

// Make room for `extra` more points (or merged nodes).  The host only tracks an upper bound of the node count;
// when that bound asks for a larger table the real count is fetched (one sync) before anything is moved.
int ensure_capacity_for(gndt_handle* h, uint64_t extra, hipStream_t s) {
    uint32_t want = cap_for_nodes(expected_nodes_for_batch(h, h->nodes_bound, extra));
    if (h->cap == 0) return alloc_table(h, want, s);
    if (want <= h->cap) return GNDT_OK;
    if (h->table_dirty) {
        int rc = fetch_counters(h, s);
        if (rc) return rc;
        h->nodes_bound = h->h_cnt->num_nodes;
        want = cap_for_nodes(expected_nodes_for_batch(h, h->nodes_bound, extra));
        if (want <= h->cap) return GNDT_OK;
    }
    return grow_table(h, want, s);
}

int grow_table(gndt_handle* h, uint32_t new_cap, hipStream_t s) {
    if (!h->table_dirty) return alloc_table(h, new_cap, s);
    // export the current contents (the node list is always valid), rebuild, merge back
    int rc = fetch_counters(h, s);
    if (rc) return rc;
    const uint32_t C = h->h_cnt->num_nodes;
    rc = ensure_stats_buffers(h, C);
    if (rc) return rc;
    if (C) {
        hipLaunchKernelGGL(k_stats_export, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->node_slot, h->d_cnt,
                           h->st_key, h->st_sums, h->st_count, h->st_first);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    rc = alloc_table(h, new_cap, s);
    if (rc) return rc;
    // the new table is empty: node list restarts, nobody owns a column entry yet (stream position is kept)
    HIP_TRY(h, hipMemsetAsync(&h->d_cnt->num_nodes, 0, sizeof(uint32_t), s));
    HIP_TRY(h, hipMemsetAsync(&h->d_cnt->prev_nodes, 0, sizeof(uint32_t), s));
    if (C) {
        hipLaunchKernelGGL(k_stats_merge, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->cap - 1, h->node_slot,
                           h->index_of_slot, h->st_key, h->st_sums, h->st_count, h->st_first, (uint64_t)C, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        h->table_dirty = true;
    }
    h->nodes_bound = C;
    return GNDT_OK;
}

}  // namespace

// gndt_reserve: a node table for `nodes` nodes (what a table that holds a map already is grown to, its contents kept)
int reserve_table(gndt_handle* h, uint64_t nodes, hipStream_t s) {
    const uint32_t want = cap_for_nodes(nodes);
    if (h->cap >= want) return GNDT_OK;
    return (h->cap && h->table_dirty) ? grow_table(h, want, s) : alloc_table(h, want, s);
}

// gndt_sync on a handle in deferred-emit mode: the ordering + emit pass over the whole map, once, for all the frames since the last read
int table_emit_pending(gndt_handle* h) {
    if (!h->emit_pending && !h->deferred_captured) return GNDT_OK;
    hipStream_t s = h->last_stream;
    const uint64_t words = std::max<uint64_t>(h->pending_words, 1);
    int rc = launch_order_and_emit(h, words, 4, s, false, true, true, 0u, false);     // (advance 0: the frames did their own bookkeeping)
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));
    h->emit_pending = false;
    if (!h->deferred_captured) h->pending_words = 0;      // (a captured frame may be replayed again: the next read emits again)
    ++h->result_serial;                                   // (a cost map of the stale rows is stale, too)
    return GNDT_OK;
}

// gndt_sync found PartCounters::small_fallback after an eager finalisation: the same table through the regular kernels
int table_refinalize(gndt_handle* h) {
    h->small_ok = false;
    hipStream_t s = h->last_stream;
    int rc = do_finalize(h, s, false, 0, 0, (uint32_t)std::min<uint64_t>(h->stream_pos, 0xFFFFFFFEull));
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));
    return GNDT_OK;
}

// the sample's three device words (zero between launches: the kernel's last workgroup leaves them so), two pinned words, its event
int ensure_sample_buffers(gndt_handle* h) {
    if (!h->d_sample) {
        GNDT_NO_CAPTURE(h, "the locality sample's buffers");
        HIP_TRY(h, hipMalloc(&h->d_sample, 3 * sizeof(unsigned long long)));
        HIP_TRY(h, hipHostMalloc(&h->h_sample, 2 * sizeof(unsigned long long)));
        { const int rc = zero_device_now(h, h->d_sample, 3 * sizeof(unsigned long long)); if (rc) return rc; }
    }
    if (!h->sample_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->sample_ev, hipEventDisableTiming));
    return GNDT_OK;
}

int locality_sample_begin(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, hipStream_t s) {
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    h->sample_pending = false;
    if (n == 0 || tiles == 0) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the locality sample");
    { const int arc = ensure_sample_buffers(h); if (arc) return arc; }
    tiles = (uint32_t)std::min<uint64_t>(tiles, (n + kTileCheck - 1) / kTileCheck);
    // (the kernel's last workgroup stores the totals into h_sample and leaves d_sample zero: nothing in front of it, nothing behind it)
    const float* p = static_cast<const float*>(xyz_dev);
    if (stride_bytes == 12) hipLaunchKernelGGL(k_tile_sample<3>, dim3(tiles), dim3(kTileT), 0, s, p, (uint64_t)n, grid_params(h), tiles, h->d_sample, h->h_sample);
    else hipLaunchKernelGGL(k_tile_sample<4>, dim3(tiles), dim3(kTileT), 0, s, p, (uint64_t)n, grid_params(h), tiles, h->d_sample, h->h_sample);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->sample_ev, s));
    h->sample_pending = true;
    h->sample_n = n;
    return GNDT_OK;
}

bool locality_sample_take(gndt_handle* h, bool wait, double* ratio) {
    if (!h->sample_pending) return false;
    if (wait) { if (hipEventSynchronize(h->sample_ev) != hipSuccess) { (void)hipGetLastError(); h->sample_pending = false; return false; } }
    else {
        const hipError_t e = hipEventQuery(h->sample_ev);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }       // (not there yet: ask again later)
    }
    h->sample_pending = false;
    *ratio = h->h_sample[1] ? (double)h->h_sample[0] / (double)h->h_sample[1] : (double)h->h_sample[0];
    return true;
}

int locality_sample(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, double* ratio, hipStream_t s) {
    *ratio = 0.0;
    const int rc = locality_sample_begin(h, xyz_dev, n, stride_bytes, tiles, s);
    if (rc) return rc;
    (void)locality_sample_take(h, true, ratio);
    return GNDT_OK;
}

// `rec`: the input is the records of a pending owner-partitioned build (two segments, indices over [0, index_range)) that did
// not fit the partition pipeline: the same map through the node table.  The table is only the means here: the handle ends as
// after a PARTITION build (rows, order arrays; no additive state to update).
int build_atomic(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, bool tile, const gndt_handle::Pending* rec) {
    int rc;
    h->last_strategy = tile ? GNDT_STRATEGY_TILE : GNDT_STRATEGY_ATOMIC;
    h->map_in_table = true;
    const uint64_t n_all = rec ? rec->n + rec->n2 : n;
    const uint64_t reach = rec ? (rec->index_range ? rec->index_range : n_all) : n;      // the point indices run below this
    if (reach >= 0xFFFFFFFFull) { h->err = "point index exceeds 32 bits"; return GNDT_ERR_INVALID; }
    uint64_t expect = h->P.max_nodes_hint ? h->P.max_nodes_hint : std::max<uint64_t>(1024, n_all / 4);
    for (int attempt = 0; attempt < 8; ++attempt) {
        const uint32_t want = cap_for_nodes(expect);
        if (h->cap < want) { rc = alloc_table(h, want, s); if (rc) return rc; }
        mark(h, 0, s);
        rc = do_reset(h, s);
        if (rc) return rc;
        mark(h, 1, s);
        if (rec) {
            rc = do_accumulate(h, rec->xyz, rec->n, 16, 0, 0, s, 0, false, false, true);
            if (!rc) rc = do_accumulate(h, rec->xyz2, rec->n2, 16, 0, 0, s, 0, false, false, true);
        } else {
            rc = do_accumulate(h, xyz_dev, n, stride_bytes, 0, 0, s, 0, tile);
        }
        if (rc) return rc;
        h->stream_pos = reach;
        rc = do_finalize(h, s, false, 0, 0, (uint32_t)reach);    // (the device-side stream position is raised by its first kernel)
        if (rc) return rc;
        // A build being captured into a hipGraph cannot wait: it is recorded once, for the table the eager builds before it
        // settled on, and a replay whose cloud outgrows that table says so at gndt_sync (GNDT_ERR_CAPACITY), as a captured
        // gndt_update does.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap != hipStreamCaptureStatusNone) return GNDT_OK;
        // a build returns with its results ready: wait once and look at the device-side flags
        HIP_TRY(h, hipStreamSynchronize(s));
        if (h->small_used && h->part.h_pc->small_fallback) {      // not a small map after all: the regular finalisation of the same table
            h->small_ok = false;
            rc = do_finalize(h, s, false, 0, 0, (uint32_t)reach);
            if (rc) return rc;
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        if (!h->h_cnt->err_table_full && !h->part.h_pc->stage_overflow) {
            h->table_nodes_seen = h->h_cnt->num_nodes;
            if (h->table_nodes_seen <= 900u) h->small_ok = true;       // (small again: the next build may take the short way)
            // what a later PARTITION build of a similar cloud should expect (a first build without a hint guesses n / 4)
            h->part.nodes_learned = (uint64_t)h->h_cnt->num_nodes + h->h_cnt->num_nodes / 5;
            if (rec) { h->map_in_table = false; h->incr_ok = false; }
            return GNDT_OK;
        }
        // table (or staging) overflowed: the build starts from empty, so simply redo it in a larger table
        expect = (uint64_t)h->cap * 2;   // cap_for_nodes doubles again -> 4x slots
        if (expect > (1ull << 30)) break;
    }
    return GNDT_ERR_CAPACITY;
}

}  // namespace gndt_host

extern "C" {

int gndt_accumulate_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes,
                           uint64_t first_idx_base, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state: create the handle "
                 "with strategy = GNDT_STRATEGY_ATOMIC for incremental updates, or call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    rc = ensure_capacity_for(h, n, s);
    if (rc) return rc;
    mark(h, 1, s);
    rc = do_accumulate(h, xyz_dev, n, stride_bytes, first_idx_base, 0, s);
    if (rc) return rc;
    mark(h, 2, s);
    h->stream_pos = std::max<uint64_t>(h->stream_pos, first_idx_base + n);
    hipLaunchKernelGGL(k_raise_stream, dim3(1), dim3(64), 0, s, h->d_cnt, (uint32_t)h->stream_pos);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}

int gndt_finalize_device(gndt_handle* h, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table) {
        h->err = "nothing accumulated in the node table: the current map was built by a PARTITION strategy (call gndt_reset, "
                 "then gndt_accumulate_device / gndt_stats_merge_device)";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    if (h->cap == 0) {
        rc = alloc_table(h, cap_for_nodes(1024), s);
        if (rc) return rc;
    }
    return do_finalize(h, s);
}

int gndt_update_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state: create the handle "
                 "with strategy = GNDT_STRATEGY_ATOMIC for incremental updates, or call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    next_event_set(h);
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    // Host side: only upper bounds, so that buffers exist (allocation happens outside any graph capture: run one
    // frame eagerly first, or give max_nodes_hint / max_points_hint).  The first_idx base is the DEVICE-side
    // stream position, which k_advance_stream bumps, so a captured update can be replayed frame after frame.
    rc = ensure_capacity_for(h, n, s);
    if (rc) return rc;
    mark(h, 1, s);
    // The frame's points list the nodes they touch; if the staging rows and the column order of the last finalisation
    // are still in place, only the columns holding a touched node are relabelled (the ordering and the emit pass
    // still cover the whole map: rows move when a column in front of them grows).
    const bool incr = h->incr_ok;
    bool tile = n >= 4096 && (h->P.strategy == GNDT_STRATEGY_TILE || (h->P.strategy == GNDT_STRATEGY_AUTO && h->tile_choice == 1));
    rc = do_accumulate(h, xyz_dev, n, stride_bytes, h->stream_pos, 1, s, incr ? 1 : 0, tile, /*defer_advance=*/true);
    if (rc) return rc;
    h->stream_pos += n;
    return do_finalize(h, s, incr, n, (uint32_t)n);
}

int gndt_set_deferred_emit(gndt_handle* h, int on) {
    if (!h) return GNDT_ERR_INVALID;
    h->defer_emit = on != 0;
    if (!on && h->deferred_captured) {      // frames a graph replayed since the last read get their rows at the next read; after that, none are deferred
        h->emit_pending = true;
        h->deferred_captured = false;
    }
    return GNDT_OK;
}

int gndt_update(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    rc = stage_host_input(h, xyz_host, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    rc = gndt_update_device(h, h->stage, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

int gndt_stats_export_device(gndt_handle* h, gndt_stats* out, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!out) return GNDT_ERR_INVALID;
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state (use strategy ATOMIC)";
        return GNDT_ERR_INVALID;
    }
    if (h->cap == 0) { rc = alloc_table(h, cap_for_nodes(1024), s); if (rc) return rc; }
    rc = fetch_counters(h, s);
    if (rc) return rc;
    if (h->h_cnt->err_table_full) { h->err = "node table full: raise gndt_params.max_nodes_hint"; return GNDT_ERR_CAPACITY; }
    const uint32_t C = h->h_cnt->num_nodes;
    rc = ensure_stats_buffers(h, C);
    if (rc) return rc;
    if (C) {
        hipLaunchKernelGGL(k_stats_export, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->node_slot, h->d_cnt,
                           h->st_key, h->st_sums, h->st_count, h->st_first);
        HIP_TRY(h, hipGetLastError());
    }
    out->num_nodes = C;
    out->key = h->st_key; out->sums = h->st_sums; out->count = h->st_count; out->first_idx = h->st_first;
    return GNDT_OK;
}

int gndt_stats_merge_device(gndt_handle* h, const gndt_stats* in, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!in) return GNDT_ERR_INVALID;
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table) {
        h->err = "the current map was built by a PARTITION strategy and does not live in the node table: call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    rc = ensure_capacity_for(h, in->num_nodes, s);
    if (rc) return rc;
    if (in->num_nodes) {
        hipLaunchKernelGGL(k_stats_merge, dim3(grid_for(in->num_nodes)), dim3(kBlock), 0, s, h->keys, h->acc, h->cap - 1,
                           h->node_slot, h->index_of_slot, in->key, in->sums, in->count, in->first_idx, (uint64_t)in->num_nodes,
                           h->d_cnt);
        h->incr_ok = false;
        HIP_TRY(h, hipGetLastError());
        h->table_dirty = true;
        h->results_valid = false;
        // the merged first indices tell how far the point stream reaches (sizes the column-order bitmap);
        // the exchange path may wait for the host, and learns the exact node count on the way
        rc = fetch_counters(h, s);
        if (rc) return rc;
        h->stream_pos = std::max<uint64_t>(h->stream_pos, h->h_cnt->stream_pos);
        h->nodes_bound = h->h_cnt->num_nodes;
    }
    return GNDT_OK;
}


int gndt_remove_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if (!h->map_in_table || h->cap == 0) {
        h->err = "gndt_remove needs the additive node table: build with strategy ATOMIC / TILE or through gndt_update*";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    next_event_set(h);
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    if (n) {
        const float* p = static_cast<const float*>(xyz_dev);
        const int blocks = grid_for(n, kBlock, 256 * 16);
        if (stride_bytes == 12) hipLaunchKernelGGL(k_remove<3>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, grid_params(h), h->keys, h->acc, h->cap - 1, h->d_cnt);
        else hipLaunchKernelGGL(k_remove<4>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, grid_params(h), h->keys, h->acc, h->cap - 1, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
    }
    h->incr_ok = false;                                // every row is redone
    h->results_valid = false;
    if ((rc = fetch_counters(h, s))) return rc;        // (waits) misses and deaths are only known on the device
    if (h->h_cnt->err_remove) {
        h->err = std::to_string(h->h_cnt->err_remove) + " point(s) to remove were never added (or their node is already empty): the map is "
                 "partly updated, call gndt_reset";
        HIP_TRY(h, hipMemsetAsync(&h->d_cnt->err_remove, 0, sizeof(uint32_t), s));
        return GNDT_ERR_INVALID;
    }
    if (h->h_cnt->n_dead) {
        // drop the empty nodes: statistics out, fresh table, merge back (k_stats_merge skips entries without points)
        HIP_TRY(h, hipMemsetAsync(&h->d_cnt->n_dead, 0, sizeof(uint32_t), s));
        if ((rc = grow_table(h, h->cap, s))) return rc;
    }
    return do_finalize(h, s);
}

int gndt_remove(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    rc = stage_host_input(h, xyz_host, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    rc = gndt_remove_device(h, h->stage, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

int gndt_locality_sample(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, double* points_per_partial,
                         void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!points_per_partial || (!xyz_dev && n)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    return locality_sample(h, xyz_dev, n, stride_bytes, tiles, points_per_partial, s);
}

}  // extern "C"
