// gndt_api_build.hip — strategy PARTITION (gndt_partition.hpp, gndt_bucket3.hpp): launch, pending-build resolution, gndt_build*.
#include <atomic>

#include "gndt_handle.hpp"
#include "gndt_bucket3.hpp"
#include "gndt_blocked.hpp"
using namespace gndt;
using namespace gndt_host;

namespace gndt_host {

// prefix of the per-word column weights -> row of every staged node -> SoA rows (marks m0+1 .. m0+5)
// `grouped`: the staging rows of a column are adjacent (k_bucket_direct): the destination pass works per column
int launch_order_and_emit(gndt_handle* h, uint64_t words, int m0, hipStream_t s, bool grouped, bool counters_to_host, bool tab_end, uint32_t advance,
                          bool partial) {
    auto& q = h->part;
    // (small clouds, staging rows grouped by column: the destination pass scans the word weights itself — gndt_partition.hpp)
    const bool dest_scans = grouped && !partial && !tab_end && words <= kDestScanMax && tuning().dest_scans;
    if (dest_scans) {
    } else if (words <= kScanSmallMax) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, (const uint32_t*)q.word_weight, (uint32_t)words, q.word_base);
    } else {
        const uint32_t nbw = (uint32_t)((words + kScanChunk - 1) / kScanChunk);
        // (two launches: a single-launch chained scan — every workgroup waiting for the tagged sums of those in front of it — was
        //  measured in round 4: 11.8 against 11.4 us on the bench scene, 17.7 against 13 on 32 M points.  Launches that follow
        //  each other on a stream overlap their start-up; what a fused kernel saves is not there to be saved.)
        hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr,
                           (uint32_t)words, q.bsum_words);
        hipLaunchKernelGGL(k_scan_apply<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr,
                           (uint32_t)words, q.bsum_words, q.word_base);
    }
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 1, s);
    mark(h, m0 + 2, s);       // (the column-rank and column-scan passes of earlier versions: phases kept for the ABI, empty)
    mark(h, m0 + 3, s);
    if (grouped && !partial && !tab_end && words <= (uint64_t)tuning().place_emit_words) {
        // small clouds: the ordering pass and the emit pass as one kernel in scatter form (gndt_partition.hpp)
        mark(h, m0 + 4, s);
        if (dest_scans)
            hipLaunchKernelGGL(k_place_emit_rows<true>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, (const RawNode*)q.raw, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                               q.ncol_at, h->out, q.row_ncol, h->d_cnt, q.d_pc, counters_to_host ? h->h_cnt : (Counters*)nullptr,
                               counters_to_host ? q.h_pc : (PartCounters*)nullptr, h->cur_capture_id, h->pending.gp, (const uint32_t*)q.word_weight, (uint32_t)words);
        else
            hipLaunchKernelGGL(k_place_emit_rows<false>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, (const RawNode*)q.raw, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                               q.ncol_at, h->out, q.row_ncol, h->d_cnt, q.d_pc, counters_to_host ? h->h_cnt : (Counters*)nullptr,
                               counters_to_host ? q.h_pc : (PartCounters*)nullptr, h->cur_capture_id, h->pending.gp, (const uint32_t*)q.word_weight, (uint32_t)words);
        HIP_TRY(h, hipGetLastError());
        mark(h, m0 + 5, s);
        return GNDT_OK;
    }
    if (grouped && dest_scans)
        hipLaunchKernelGGL(k_order_dest_columns<true>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                           q.ncol_at, q.inv, h->d_cnt, q.d_pc, (const uint32_t*)q.word_weight, (uint32_t)words);
    else if (grouped)
        hipLaunchKernelGGL(k_order_dest_columns<false>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                           q.ncol_at, q.inv, h->d_cnt, q.d_pc, (const uint32_t*)q.word_weight, (uint32_t)words);
    else
        hipLaunchKernelGGL(k_order_dest, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                           q.ncol_at, q.inv, h->d_cnt, q.d_pc, tab_end ? q.row_of : (uint32_t*)nullptr, partial ? 1u : 0u);
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 4, s);
    if (grouped)       // (the PARTITION strategies' RawNode rows: moments and eigen-solve in the emit pass)
        hipLaunchKernelGGL(k_emit_rows<true>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, reinterpret_cast<const StageRow*>(q.raw), q.inv, h->out, q.row_ncol, h->d_cnt,
                           q.d_pc, counters_to_host ? h->h_cnt : (Counters*)nullptr, counters_to_host ? q.h_pc : (PartCounters*)nullptr,
                           (Counters*)nullptr, 0u, EmitPartial{nullptr, nullptr, nullptr}, h->cur_capture_id, h->pending.gp);
    else
    hipLaunchKernelGGL(k_emit_rows<false>, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.stage, q.inv, h->out, q.row_ncol, h->d_cnt,
                       q.d_pc, counters_to_host ? h->h_cnt : (Counters*)nullptr, counters_to_host ? q.h_pc : (PartCounters*)nullptr,
                       tab_end ? h->d_cnt : (Counters*)nullptr, advance,
                       partial ? EmitPartial{q.word_base, q.row_of, h->touched} : EmitPartial{nullptr, nullptr, nullptr}, h->cur_capture_id, GridParams{});
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 5, s);
    return GNDT_OK;
}

namespace {
// ---------------------------------------------------------------------------------------------
// strategy PARTITION (gndt_partition.hpp): the build
// ---------------------------------------------------------------------------------------------
// Bucket count for `nodes` expected nodes.  Few, large buckets are better for both the scatter (longer runs per
// workgroup and bucket) and the bucket kernel (fixed costs per bucket): as many points per bucket as two chunks of
// the bucket kernel take (2800 leaves room for the spread of a hash partition), unless the LDS node table says
// otherwise: average load <= 0.5 of `slots` against an estimate that already carries a 20 % margin, i.e. ~0.42 of
// the slots really used, far below the overflow limit (a table holds `slots` nodes).  (An overflow is not an
// error: the build is re-run with the larger table / more buckets.)
uint64_t buckets_for(uint64_t n, uint64_t nodes, int slots, int load_pct, bool wide = false) {
    const int pts_target = tuning().bucket_points;
    if (pts_target) return std::max<uint64_t>(n / (uint64_t)pts_target, 16);
    // Clouds of up to a few million points do not fill the chip with 2800-point buckets (200 k points: 71 workgroups for 256
    // CUs): down to 700 points per bucket below a million points, measured 5-13 % faster there (campus / bridge / terrain / uniform
    // clouds of 0.1-1 M points) and slower from 2 M points on.
    // Round 4: 3600 (was 2800) for clouds that fill the chip either way — with the fingerprint index a bucket's accumulate phase
    // is shorter and the per-bucket phases weigh more: bench scene 2000 / 2400 / 2800 / 3072 / 3600 / 4000 points per bucket:
    // 0.418 / 0.399 / 0.393 / 0.383 / 0.372 / 0.385 ms per build (node-heavy clouds get their bucket count from the nodes, below).
    // Round 6: 4000 for the three-workgroups-per-CU kernel (`wide`: one record per thread, 512 records per iteration — bench scene 2600 /
    // 3000 / 3255 / 3600 / 4000 / 4340 / 5000 points per bucket: bucket kernel 173 / 149 / 136 / 138 / 133 / 131 / 160 us; beyond 4340
    // too many tables overflow into the second pass).
    const uint64_t small_cloud = std::min<uint64_t>(wide ? 4000 : 3600, std::max<uint64_t>(700, n / 1024));
    const uint64_t per_bucket = slots >= 1024 ? 6400 : small_cloud;
    const uint64_t node_room = (uint64_t)slots * load_pct;
    const uint64_t want = std::max<uint64_t>(n / per_bucket, (nodes * 100) / node_room);
    return std::max<uint64_t>(want, 16);
}
constexpr uint64_t kMaxBuckets = 32768;   // 4-byte LDS cursor per bucket in the partition passes

}  // namespace

// cursors of both partition levels, the sample votes, the buckets' ranges and the bucket kernel's retry list, for B buckets
int ensure_cursors(gndt_handle* h, uint64_t B) {
    auto& q = h->part;
    if (B <= q.cur_cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the partition cursors");
    for (uint32_t** a : {&q.cursors, &q.cursors_alt, &q.range_lo, &q.range_hi, &q.range_cap}) { release_device(h, *a); *a = nullptr; }
    q.cur_cap = 0;
    q.alt_clean = false;
    const uint64_t c = B + B / 4;
    HIP_TRY(h, hipMalloc(&q.cursors, ((size_t)kCursor1Words + 2 * c) * 4));  // [kCursor1Words] level 1 (a line each), [c] level 2, [c] samples
    HIP_TRY(h, hipMalloc(&q.cursors_alt, ((size_t)kCursor1Words + 2 * c) * 4));      // (the set the next eager build takes: FoldClear)
    HIP_TRY(h, hipMalloc(&q.range_lo, c * 4));
    HIP_TRY(h, hipMalloc(&q.range_hi, c * 4));                        // (the retry list)
    HIP_TRY(h, hipMalloc(&q.range_cap, c * 4));
    q.cur_cap = c;
    // The other set starts clean (FoldClear): a handle's first eager one-level build already goes without the clear launch — after
    // gndt_reserve / gndt_warmup that is the first build of the process, the one the reference's user waits for.
    if (q.d_pc_alt && !h->capturing) {
        HIP_TRY(h, hipMemsetAsync(q.cursors_alt, 0, ((size_t)kCursor1Words + 2 * c) * 4, h->own_stream));
        HIP_TRY(h, hipMemsetAsync(q.d_pc_alt, 0, sizeof(PartCounters), h->own_stream));
        HIP_TRY(h, hipStreamSynchronize(h->own_stream));
        q.alt_clean = true;
    }
    return GNDT_OK;
}

// FoldClear (gndt_partition.hpp, Part::cursors_alt): an eager one-level PARTITION build of points (a frame of < 1 M points) takes the
// cursors / partition counters the level-1 kernel of the build before it zeroed and launches no k_part_clear; its own level-1 kernel
// zeroes the other set, the bitmap and the Counters.  200 k-point campus frame: 0.0491 -> 0.0472 ms back to back, 0.068 -> 0.0635
// awaited; bridge_ground 0.0586 -> 0.0574, 0.078 -> 0.074 (profiles/r06_ablation.txt 9).  Returns what that kernel is to zero (words == 0: a k_part_clear launch still prepares THIS build — the first build of a
// handle, buffers that grew; alt_cursors == nullptr: nothing on the side — records, recorded builds, handles that ever recorded one:
// a replay works on the set it was recorded with and the host does not see it).
static FoldClear fold_clear_begin(gndt_handle* h, const gndt_handle::Pending& P, uint64_t words) {
    auto& q = h->part;
    if (!tuning().fold_clear || P.records || h->capturing || h->ever_captured || !q.cursors_alt || !q.d_pc_alt) return FoldClear{};
    const bool clean = q.alt_clean;
    if (clean) { std::swap(q.cursors, q.cursors_alt); std::swap(q.d_pc, q.d_pc_alt); }
    q.alt_clean = false;                   // (the other set is the last build's now; partition_launch raises the flag once the level-1 kernel that zeroes it is launched)
    // (what a one-level build uses of a cursor set, and all "clean" means: [0, kMaxFan) dense level-1 cursors + the sample votes behind them)
    return FoldClear{q.cursors_alt, (uint32_t)(kMaxFan + 2 * q.cur_cap), q.d_pc_alt, q.bitmap, q.word_weight, clean ? words : 0ull};
}

// One attempt of the PARTITION build: every launch plus the asynchronous read-back of the counters and overflow
// flags; no host wait.  Returns GNDT_OK, an error, or -1 when the partition path cannot hold this input.
int partition_launch(gndt_handle* h, gndt_handle::Pending& P) {
    auto& q = h->part;
    const size_t n = P.n + P.n2, stride_bytes = P.stride;       // (n2: the second segment of a records build)
    hipStream_t s = P.s;
    const int attempt = P.attempt;
    uint64_t& nodes_est = P.nodes_est;
    uint64_t& stage_want = P.stage_want;
    const int env_slots = tuning().bucket_slots, part_wgs = kPartWgs;
    const uint32_t nwg = (uint32_t)std::min<uint64_t>((uint64_t)part_wgs, std::max<uint64_t>(1, n / 8192));
    const uint64_t words = ((P.index_range ? (size_t)P.index_range : n) + 31) / 32 + 1;
    int rc;
    // identical consecutive points travel as weighted records whose index word gives up two bits (gndt_partition.hpp);
    // records from an owner split were compressed where they came from and keep their index words
    const uint32_t compress = (!P.records && (uint64_t)P.first_base + n < (uint64_t)kWeightIndexLimit) ? 1u : 0u;
    const uint32_t part_mode = P.records ? kPartModeRecords : 0u;
    GridParams gp = P.gp;                              // as they were at launch (a retry must not pick up a new origin)
    // Blocked buckets (gndt_blocked.hpp): the map of the last build on this handle was a dense, evenly filled box and this cloud has
    // its size — spatial blocks of 512 nodes as buckets, the bucket kernel addresses its table directly.  Only with the two-level
    // partition (large clouds), never for records / statistics / a re-run after a blocked attempt failed.
    // (not under hipGraph capture either: a recorded build is replayed on OTHER clouds, and one that leaves the box cannot be re-run)
    const bool blocked = tuning().blocked && q.blk_state == 1 && !P.no_block && !P.records && !P.stats_only && attempt == 0 && !h->capturing && !P.captured &&
                         !tuning().bucket_slots && q.two_level_ok && n >= (1u << 20) &&
                         h->P.strategy != GNDT_STRATEGY_PARTITION_EXACT && n <= q.blk_n + q.blk_n / 4 && n + n / 4 >= q.blk_n;
    P.blocked = blocked;
    if (blocked) gp.blk = q.blk_map;
    const float* p = static_cast<const float*>(P.xyz);
    const float* p2 = static_cast<const float*>(P.xyz2);
    // table size and bucket count for this attempt: 512-slot tables unless that needs too many buckets
    // attempt 0: 512-slot tables; an overflow first doubles the table (same estimate), then raises the estimate
    int bslots = env_slots ? env_slots : (attempt == 0 ? 512 : 1024);
    // Node-heavy clouds (a few points per node) get their bucket count from the NODES: estimate / (slots x load).  From 2 M points
    // on any such count fills the chip, and fuller tables with fewer buckets are faster (size sweep, round 3: load 75 against 60
    // gains 3-11 % on 3-30 M-point clouds of 0.05-0.2 m voxels, S2z 0.84 -> 0.79 ms, S3 32 M 1.50 -> 1.47; a million points lose
    // 4-13 % with it: fewer buckets than CUs want).  The estimate carries a 20 % margin, so 75 % is ~62 % of the slots really used.
    int load_pct = P.load_pct ? P.load_pct : q.load_pct;
    if (!P.load_pct && n >= (1u << 21)) load_pct = std::max(load_pct, tuning().bucket_load_large);
    // (which bucket kernel this build gets — gndt_bucket3.hpp: three workgroups per CU for clouds without locality — also sizes its buckets)
    const bool wide_hint = tuning().bucket_three_wgs && !P.stats_only && tuning().interleave < 0 && !(q.pair_ratio < 0.0 || q.pair_ratio > 0.02);
    // (its buckets are sized by the POINTS — 4000 each — unless the nodes really fill the tables: the estimate carries a 20-30 % margin
    //  and the second pass takes the tables that overflow all the same; with the 75 % rule the bench scene's hint made 2 730 buckets
    //  of 3 660 points where 2 500 of 4 000 are 4 % faster)
    if (wide_hint && bslots == 512 && !P.load_pct && tuning().retry_pass) load_pct = std::max(load_pct, 90);
    uint64_t Bw = buckets_for(n, nodes_est, bslots, load_pct, wide_hint && bslots == 512);
    // Clouds with few points per node (a million points in half a million nodes): the NODES size the bucket count, and 512-slot
    // tables — which since round 3 hold 512 nodes and no longer overflow on such clouds — would get buckets of a few hundred
    // points, all per-bucket overhead (size sweep: 0.217 against 0.186 ms).  Below 900 points per bucket the 1024-slot tables
    // are taken from the start, if they still fill the chip.
    if (!env_slots && attempt == 0 && P.est_reliable && n / std::max<uint64_t>(Bw, 1) < 900) {
        const uint64_t b1024 = buckets_for(n, nodes_est, 1024, load_pct);
        if (b1024 >= 384) { bslots = 1024; Bw = b1024; }
    }
    if (blocked) { bslots = 512; Bw = q.blk_buckets; }       // (one bucket per block of the box)
    // Two-level partition (no counting passes) for large builds; the exact single-level counting partition for small
    // ones, when asked for (GNDT_STRATEGY_PARTITION_EXACT), and after a region overflowed once on this handle.
    const int env_two = tuning().two_level;
    bool two = h->P.strategy != GNDT_STRATEGY_PARTITION_EXACT && q.two_level_ok &&
               (n >= (1u << 20) || h->P.strategy == GNDT_STRATEGY_PARTITION_TWO_LEVEL);
    if (env_two == 0 || n == 0) two = false;
    if (two && Bw > (uint64_t)kMaxFan * kMaxFan) {         // more buckets than two levels address: larger tables, fewer buckets
        if (!env_slots) { bslots = 1024; Bw = buckets_for(n, nodes_est, bslots, load_pct); }
        if (Bw > (uint64_t)kMaxFan * kMaxFan) two = false;
    }
    if (!two) {
        if (!env_slots && Bw > kMaxBuckets) { bslots = 1024; Bw = buckets_for(n, nodes_est, bslots, load_pct); }
        if (Bw > kMaxBuckets) return -1;                   // too many nodes for one partition level: atomic path
    }
    // (a cloud just above kMaxFan buckets of the small-cloud size still fits kMaxFan larger ones, if its nodes do)
    if (!two && tuning().one_level && q.one_level_ok && h->P.strategy != GNDT_STRATEGY_PARTITION_EXACT && Bw > (uint64_t)kMaxFan &&
        n <= (uint64_t)kMaxFan * 2800 && (nodes_est * 100) / ((uint64_t)bslots * load_pct) <= (uint64_t)kMaxFan)
        Bw = kMaxFan;
    const uint32_t B = (uint32_t)Bw;
    // Small clouds (at most kMaxFan buckets): ONE tile-sort level writes the buckets themselves, each with a fixed room of 4 x
    // the mean fill — a third of the counting partition's time (no histogram pass, no offsets pass, coalesced copy-out).  A
    // bucket that overflows its room sends the build (and this handle from then on) to the counting partition.
    const bool one = !two && tuning().one_level && q.one_level_ok && h->P.strategy != GNDT_STRATEGY_PARTITION_EXACT && B <= (uint32_t)kMaxFan &&
                     n >= (1u << 14);
    P.two_level = two;
    P.one_level = one;
    if (blocked && !two) { P.no_block = true; return partition_launch(h, P); }      // (blocked buckets ride on the two partition levels only)
    h->last_strategy = two ? (blocked ? GNDT_STRATEGY_PARTITION_BLOCKED : GNDT_STRATEGY_PARTITION)
                           : (one ? GNDT_STRATEGY_PARTITION_ONE_LEVEL : GNDT_STRATEGY_PARTITION_EXACT);
    stage_want = std::max<uint64_t>(stage_want, nodes_est + nodes_est / 8);
    if (P.stats_only) {
        if ((rc = ensure_stats_buffers(h, stage_want))) return rc;
    } else {
        if ((rc = ensure_stage(h, stage_want, false))) return rc;       // (the order arrays; the rows are RawNode records)
        if ((rc = ensure_raw(h, std::max<uint64_t>(P.rows_floor, nodes_est + nodes_est / 8)))) return rc;
        if ((rc = ensure_out(h, q.stage_cap))) return rc;
    }
    BucketRanges ranges{nullptr, nullptr, nullptr, 0u};
    const float4* bucket_recs = nullptr;
    if (two) {
        uint32_t F2_shift = 1;                             // fan-out ~ sqrt(B) per level, F2 a power of two, both <= kMaxFan
        while (F2_shift < 9 && (1ull << (2 * F2_shift)) < B) ++F2_shift;
        const uint32_t F2 = 1u << F2_shift;
        const uint32_t F1 = (B + F2 - 1) / F2;
        // Level-1 regions are large and hash-balanced: a fixed capacity of 2x the mean (or 1.25x the fullest one an
        // earlier build on this handle saw).  The buckets' regions are laid out on the device from a 1-in-64 sample
        // level 1 takes (k_part2_layout: 2x the estimate + 2048 each), so LiDAR clouds' hot columns get the room they
        // need and the records take 2 n + 2048 B slots in all.  If a region overflows all the same, the build is
        // re-run; a second failure sends this handle to the exact counting partition.
        const uint32_t R = 1;                  // sub-regions per coarse region (cursor replicas): 2 / 4 measured 0.404 / 0.425 against 0.403 ms per step
        const uint32_t V = F1 * R;
        constexpr uint64_t kTile1 = (uint64_t)kTileThreads * kTilePer1, kTile2 = (uint64_t)kTileThreads * kTilePer2;
        const double r1 = std::max(2.0, q.fill1_ratio * 1.25);
        const uint64_t cap1w = (uint64_t)(r1 * (double)(n / V)) + 2 * kTile1;
        const uint64_t recs_want = 2 * (uint64_t)n + n / 8 + 2048ull * B + 4096;     // 2 n + 2048 B, and sampling slack
        if ((uint64_t)V * cap1w > 4 * (uint64_t)n + (1u << 24) || recs_want >= 0xF0000000ull || q.two_level_failures >= 2) {
            q.two_level_ok = false;
            return partition_launch(h, P);                 // (re-enters on the exact path)
        }
        const uint32_t cap1 = (uint32_t)cap1w;
        P.mean1 = (double)(n / V);
        if ((rc = grow_buf(h, q.recs1, q.rec1_cap, (uint64_t)V * cap1))) return rc;
        if ((rc = grow_buf(h, q.recs, q.rec_cap, recs_want))) return rc;
        if ((rc = ensure_cursors(h, B))) return rc;
        // (no FoldClear here: measured on the bench scene, one call, A B A B: 0.3129 | 0.3143 ms per step with it, 0.3116 | 0.3117 with the
        //  k_part_clear launch — level 1 and level 2 each a microsecond slower; the one-level partition of a small frame is where it pays)
        const FoldClear fold{};
        uint32_t* cursor1 = q.cursors;
        uint32_t* cursor2 = q.cursors + kCursor1Words;
        uint32_t* est2 = cursor2 + B;
        mark(h, 0, s);
        // (a build recorded into a hipGraph always records the reset: its replays must not depend on what the table held at capture time)
        if (h->table_dirty || h->capturing) { if ((rc = do_reset(h, s))) return rc; }
        h->results_valid = false;
        if (!fold.words) {
            hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                               q.cursors, (uint32_t)(kCursor1Words + 2 * B));
            HIP_TRY(h, hipGetLastError());
        }
        mark(h, 1, s);
        const uint32_t tiles1 = (uint32_t)((P.n + kTile1 - 1) / kTile1), tiles1b = (uint32_t)((P.n2 + kTile1 - 1) / kTile1);
        const uint32_t l1_wgs = tuning().l1_wgs;   // persistent workgroups (2 resident per CU)
        const bool wide = std::max(V, F2) > 256;           // LDS arrays for a fan-out of 512 (fewer resident tiles) only when needed
        const dim3 g1(std::max<uint32_t>(1, std::min<uint32_t>(tiles1, l1_wgs))), g1b(std::max<uint32_t>(1, std::min<uint32_t>(tiles1b, l1_wgs))),
            g2((uint32_t)((cap1 + kTile2 - 1) / kTile2), V);
        // (the last level-1 workgroup to finish lays out the buckets' regions: range_lo / range_cap)
#define GNDT_L1(SF_, FAN_)                                                                                                  \
    hipLaunchKernelGGL((k_part2_level1<SF_, FAN_>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, P.first_base, gp, B, F1, F2_shift, \
                       R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, q.range_lo,     \
                       q.range_cap, (uint64_t)q.rec_cap, fold, kCursor1Shift)
        // records: the two segments one after the other into the same regions (the cursors carry on; the last launch lays out)
#define GNDT_L1R(FAN_)                                                                                                      \
    do {                                                                                                                    \
        if (P.n) hipLaunchKernelGGL((k_part2_level1<4, FAN_, true>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)P.n, P.first_base, gp, B, F1, \
                                    F2_shift, R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, \
                                    P.n2 ? (uint32_t*)nullptr : q.range_lo, q.range_cap, (uint64_t)q.rec_cap, FoldClear{}, kCursor1Shift);  \
        if (P.n2) hipLaunchKernelGGL((k_part2_level1<4, FAN_, true>), g1b, dim3(kTileThreads), 0, s, p2, (uint64_t)P.n2, P.first_base, gp, B, \
                                     F1, F2_shift, R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, \
                                     q.range_lo, q.range_cap, (uint64_t)q.rec_cap, FoldClear{}, kCursor1Shift);              \
    } while (0)
        if (P.records) { if (wide) GNDT_L1R(512); else GNDT_L1R(256); }
        else if (stride_bytes == 12) { if (wide) GNDT_L1(3, 512); else GNDT_L1(3, 256); }
        else { if (wide) GNDT_L1(4, 512); else GNDT_L1(4, 256); }
#undef GNDT_L1
#undef GNDT_L1R
        HIP_TRY(h, hipGetLastError());
        mark(h, 2, s);
        mark(h, 3, s);
        if (wide)
            hipLaunchKernelGGL(k_part2_level2<512>, g2, dim3(kTileThreads), 0, s, q.recs1, cursor1, cap1, R, gp, B, F2, cursor2,
                               q.range_lo, q.range_cap, q.recs, q.d_pc);
        else
            hipLaunchKernelGGL(k_part2_level2<256>, g2, dim3(kTileThreads), 0, s, q.recs1, cursor1, cap1, R, gp, B, F2, cursor2,
                               q.range_lo, q.range_cap, q.recs, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        mark(h, 4, s);
        ranges = BucketRanges{q.range_lo, q.range_cap, cursor2, 0u};
    } else if (one) {
        constexpr uint64_t kTile1 = (uint64_t)kTileThreads * kTilePer1;
        const uint64_t mean = n / B + 1;
        const uint32_t cap1 = (uint32_t)std::max<uint64_t>(4 * mean, mean + 4096);
        if ((rc = grow_buf(h, q.recs1, q.rec1_cap, (uint64_t)B * cap1))) return rc;
        if ((rc = ensure_cursors(h, B))) return rc;
        const FoldClear fold = fold_clear_begin(h, P, words);     // (may swap q.cursors / q.d_pc with their clean twins)
        uint32_t* cursor1 = q.cursors;                     // [B <= kMaxFan] the buckets' fills
        uint32_t* est2 = q.cursors + kMaxFan;              // (the kernel's sample votes: not used here, but counted; one-level: dense cursors)
        const uint32_t F1 = B, F2_shift = 0, R = 1;
        mark(h, 0, s);
        // (a build recorded into a hipGraph always records the reset: its replays must not depend on what the table held at capture time)
        if (h->table_dirty || h->capturing) { if ((rc = do_reset(h, s))) return rc; }
        h->results_valid = false;
        if (!fold.words) {
            hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                               q.cursors, (uint32_t)(kMaxFan + B));
            HIP_TRY(h, hipGetLastError());
        }
        mark(h, 1, s);
        // small clouds: 1024-point tiles (gndt_partition.hpp: kTilePerSmall), so that the frame is a launch of a few hundred workgroups
        const bool small_tiles = !P.records && n < (1u << 20) && tuning().small_tiles;
        const uint64_t tile1 = small_tiles ? (uint64_t)kTileThreads * kTilePerSmall : kTile1;
        const uint32_t tiles1 = (uint32_t)((P.n + tile1 - 1) / tile1), tiles1b = (uint32_t)((P.n2 + tile1 - 1) / tile1);
        const uint32_t l1_wgs = tuning().l1_wgs;
        const bool wide = B > 256;
        const dim3 g1(std::max<uint32_t>(1, std::min<uint32_t>(tiles1, l1_wgs))), g1b(std::max<uint32_t>(1, std::min<uint32_t>(tiles1b, l1_wgs)));
#define GNDT_L1(SF_, FAN_)                                                                                                  \
    hipLaunchKernelGGL((k_part2_level1<SF_, FAN_>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, P.first_base, gp, B, F1, F2_shift, \
                       R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, (uint32_t*)nullptr, \
                       (uint32_t*)nullptr, 0ull, fold, 0)
#define GNDT_L1S(SF_, FAN_)                                                                                                 \
    hipLaunchKernelGGL((k_part2_level1<SF_, FAN_, false, false, kTilePerSmall>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, P.first_base, gp, B, F1, \
                       F2_shift, R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, (uint32_t*)nullptr, \
                       (uint32_t*)nullptr, 0ull, fold, 0)
#define GNDT_L1R(FAN_)                                                                                                      \
    do {                                                                                                                    \
        if (P.n) hipLaunchKernelGGL((k_part2_level1<4, FAN_, true>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)P.n, P.first_base, gp, B, F1, \
                                    F2_shift, R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, \
                                    (uint32_t*)nullptr, (uint32_t*)nullptr, 0ull, FoldClear{}, 0);                              \
        if (P.n2) hipLaunchKernelGGL((k_part2_level1<4, FAN_, true>), g1b, dim3(kTileThreads), 0, s, p2, (uint64_t)P.n2, P.first_base, gp, B, \
                                     F1, F2_shift, R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc, compress, OwnerMap{nullptr, nullptr, 0u}, \
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, 0ull, FoldClear{}, 0);                             \
    } while (0)
        if (small_tiles) {
            if (stride_bytes == 12) { if (wide) GNDT_L1S(3, 512); else GNDT_L1S(3, 256); }
            else { if (wide) GNDT_L1S(4, 512); else GNDT_L1S(4, 256); }
        } else
        if (P.records) { if (wide) GNDT_L1R(512); else GNDT_L1R(256); }
        else if (stride_bytes == 12) { if (wide) GNDT_L1(3, 512); else GNDT_L1(3, 256); }
        else { if (wide) GNDT_L1(4, 512); else GNDT_L1(4, 256); }
#undef GNDT_L1
#undef GNDT_L1S
#undef GNDT_L1R
        HIP_TRY(h, hipGetLastError());
        if (fold.alt_cursors) q.alt_clean = true;          // (in stream order: the next build's level 1 runs behind this one)
        mark(h, 2, s);
        mark(h, 3, s);
        mark(h, 4, s);
        ranges = BucketRanges{nullptr, nullptr, cursor1, cap1};
    } else {
    if (P.n2) {                    // one array for the counting partition: the first segment goes into the room in front of the second
        p = p2 - 4 * P.n;
        if (P.n) HIP_TRY(h, hipMemcpyAsync(const_cast<float*>(p), P.xyz, P.n * 16, hipMemcpyDeviceToDevice, s));
    }
    if ((rc = grow_buf(h, q.recs, q.rec_cap, n))) return rc;
    if ((rc = ensure_cursors(h, B))) return rc;            // (for the bucket kernel's retry list)
    if ((rc = grow_buf(h, q.hist, q.hist_cap, (uint64_t)nwg * B))) return rc;
    if (B > q.bucket_cap) {
        GNDT_NO_CAPTURE(h, "the bucket totals");
        release_device(h, q.totals);
        release_device(h, q.bucket_base);
        q.totals = q.bucket_base = nullptr; q.bucket_cap = 0;
        HIP_TRY(h, hipMalloc(&q.totals, (size_t)B * 4));
        HIP_TRY(h, hipMalloc(&q.bucket_base, ((size_t)B + 1) * 4));
        q.bucket_cap = B;
    }
    mark(h, 0, s);
    // the HBM node table is not used by this strategy, but a previous atomic build may sit in it
    if (h->table_dirty || h->capturing) { if ((rc = do_reset(h, s))) return rc; }     // (as above: a captured build always records the reset)
    h->results_valid = false;
    hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                       (uint32_t*)nullptr, 0u);
    HIP_TRY(h, hipGetLastError());
    mark(h, 1, s);
    const size_t lds = (size_t)B * 4;
    if (lds > 48 * 1024) {   // beyond the default dynamic-LDS limit the kernels must be told (gfx950: 160 KiB/CU)
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_hist<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_hist<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_scatter<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_scatter<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (stride_bytes == 12)
        hipLaunchKernelGGL(k_part_hist<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, B, q.hist, h->d_cnt, compress, part_mode, OwnerMap{nullptr, nullptr, 0u});
    else
        hipLaunchKernelGGL(k_part_hist<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, B, q.hist, h->d_cnt, compress, part_mode, OwnerMap{nullptr, nullptr, 0u});
    HIP_TRY(h, hipGetLastError());
    mark(h, 2, s);
    hipLaunchKernelGGL(k_part_offsets, dim3((B + 31) / 32), dim3(256), 0, s, q.hist, q.totals, B, nwg);
    HIP_TRY(h, hipGetLastError());
    mark(h, 3, s);
    if (stride_bytes == 12)
        hipLaunchKernelGGL(k_part_scatter<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, P.first_base, gp, B, q.hist,
                           q.totals, q.bucket_base, q.recs, compress, part_mode, OwnerMap{nullptr, nullptr, 0u});
    else
        hipLaunchKernelGGL(k_part_scatter<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, P.first_base, gp, B, q.hist,
                           q.totals, q.bucket_base, q.recs, compress, part_mode, OwnerMap{nullptr, nullptr, 0u});
    HIP_TRY(h, hipGetLastError());
    mark(h, 4, s);
    ranges = BucketRanges{q.bucket_base, nullptr, nullptr, 0u};
    }
    bucket_recs = (two || !one) ? q.recs : q.recs1;
    if (tuning().stamps && q.dbg_buckets < B) {
        GNDT_NO_CAPTURE(h, "the stamp buffer");
        release_device(h, q.dbg);
        q.dbg = nullptr; q.dbg_buckets = 0;
        HIP_TRY(h, hipMalloc(&q.dbg, (size_t)B * 16 * sizeof(unsigned long long)));
        { const int zrc = zero_device_now(h, q.dbg, (size_t)B * 16 * sizeof(unsigned long long)); if (zrc) return zrc; }
        q.dbg_buckets = B;
    }
    q.last_buckets = B;
    // one workgroup per bucket by default: persistent workgroups (GNDT_BUCKET_WGS=512) measured 8 % slower, the
    // hardware's dynamic workgroup scheduling balances uneven buckets better than a static stride
    const uint32_t bucket_wgs = tuning().bucket_wgs;
    const dim3 bgrid(std::min<uint32_t>(B, bucket_wgs));
    const ColumnOrder order{q.bitmap, q.word_weight, q.ncol_at};
    const StatsOut stats_out{h->st_key, h->st_sums, h->st_count, h->st_first};
    unsigned long long* dbg = tuning().stamps ? q.dbg : nullptr;
    const bool grouped = true;          // k_bucket_direct stages a column's rows next to each other
    const uint32_t fp_mask = (1u << tuning().fp_bits) - 1u;
    // record pairs interleaved over the waves (gndt_bucket3.hpp) unless the last build of this handle counted next to no adjacent
    // records of one node: on by default — a cloud with locality gains 11-14 % of its bucket kernel, one without loses 2 %
    // (which mapping a cloud with locality gets: 1 = pairs interleaved inside 512-pair blocks, 2 = a contiguous stretch per lane;
    //  bucket kernel S3 32 M 595 | 585 us, S3 100 M 1.691 | 1.679 ms, S5 317 | 306 us: 2)
#ifndef GNDT_LOCALITY_MODE
#define GNDT_LOCALITY_MODE 2
#endif
    const uint32_t interleave = tuning().interleave >= 0 ? (uint32_t)tuning().interleave : (q.pair_ratio < 0.0 || q.pair_ratio > 0.02 ? (uint32_t)GNDT_LOCALITY_MODE : 0u);
    {
        // k_bucket_direct (gndt_bucket3.hpp): 512-slot tables with two workgroups per CU, 1024-slot tables on a retry
        // A bucket whose 512-slot table overflows (a few wall columns; the spread of a hash partition over tall columns) is queued and
        // done again by a SECOND PASS with 1024-slot tables over the queued buckets only (round 5) — it used to send the whole build
        // round again.  The pass is launched behind a handle's first builds, behind builds recorded into a hipGraph (a replay cannot
        // be re-run) and for as long as the last build used it; a handle whose clouds never overflow does not pay for the launch.
        // Round 6: a RECORDED build carries the pass only if nothing is known about this cloud size (a capture on a reserved fresh
        // handle) or the eager build before it used it: every replay of a recorded pass over an empty list is a launch of 256 x 1024
        // threads that all leave again (~4 us of a 52 us frame).  A replay whose tables overflow without it is
        // reported like any replay that does not fit (GNDT_ERR_CAPACITY: build that cloud eagerly, capture again).
        const bool known_size = q.last_n == n && q.last_est != 0;
        const bool retry = !blocked && bslots == 512 && tuning().retry_pass && (q.retry_pass || (h->capturing && !known_size));
        uint32_t* const rlist = retry ? q.range_hi : (uint32_t*)nullptr;
        P.retry_pass = retry;
        // one record per thread at three workgroups per CU for clouds without locality (interleave == 0: the last build counted next
        // to no adjacent records of one node), two records at two workgroups per CU — with the pair folding — for the others
        const bool wide = (interleave == 0u || tuning().bucket_three_wgs == 2) && !P.stats_only && tuning().bucket_three_wgs;
        const uint64_t rows_cap = std::min<uint64_t>(q.stage_cap, P.stats_only ? q.stage_cap : q.raw_cap);      // staging rows the order arrays AND the records hold
#define GNDT_LAUNCH_DIRECT(T_, H_, S_, GRID_, RL_, TODO_)                                                                       \
    hipLaunchKernelGGL((k_bucket_direct<T_, H_, S_>), GRID_, dim3(T_), 0, s, bucket_recs, ranges, B, gp, q.raw,    \
                       (uint32_t)(S_ ? h->st_cap : rows_cap), q.ord_cf, q.ord_idx, order, h->d_cnt, q.d_pc, dbg, stats_out, fp_mask, interleave, \
                       RL_, TODO_)
        if (blocked)        // (gndt_blocked.hpp: a block of the box per bucket, the table addressed by the key)
            hipLaunchKernelGGL(k_bucket_blocked<512>, bgrid, dim3(512), 0, s, bucket_recs, ranges, B, gp, q.raw, (uint32_t)rows_cap, q.ord_cf, q.ord_idx, order,
                               h->d_cnt, q.d_pc, dbg);
        else if (bslots == 1024) { if (P.stats_only) GNDT_LAUNCH_DIRECT(1024, 1024, true, bgrid, (uint32_t*)nullptr, (const uint32_t*)nullptr); else GNDT_LAUNCH_DIRECT(1024, 1024, false, bgrid, (uint32_t*)nullptr, (const uint32_t*)nullptr); }
        else if (wide)
            hipLaunchKernelGGL((k_bucket_direct<GNDT_DIRECT_THREADS, 512, false, 1, 6>), bgrid, dim3(GNDT_DIRECT_THREADS), 0, s, bucket_recs, ranges, B, gp, q.raw,
                               (uint32_t)rows_cap, q.ord_cf, q.ord_idx, order, h->d_cnt, q.d_pc, dbg, stats_out, fp_mask, interleave, rlist, (const uint32_t*)nullptr);
        else { if (P.stats_only) GNDT_LAUNCH_DIRECT(GNDT_DIRECT_THREADS, 512, true, bgrid, rlist, (const uint32_t*)nullptr); else GNDT_LAUNCH_DIRECT(GNDT_DIRECT_THREADS, 512, false, bgrid, rlist, (const uint32_t*)nullptr); }
        if (retry) {
            const dim3 rgrid(std::min<uint32_t>(B, 256u));       // (one 1024-thread workgroup per CU; all of them leave at once when nothing is queued)
            if (P.stats_only) GNDT_LAUNCH_DIRECT(1024, 1024, true, rgrid, (uint32_t*)nullptr, (const uint32_t*)rlist);
            else GNDT_LAUNCH_DIRECT(1024, 1024, false, rgrid, (uint32_t*)nullptr, (const uint32_t*)rlist);
        }
#undef GNDT_LAUNCH_DIRECT
    }
    HIP_TRY(h, hipGetLastError());
    mark(h, 5, s);
    // (the counters and overflow flags come back with the last kernel: k_emit_rows stores them into the host's pinned mirrors)
    if (!P.stats_only && (rc = launch_order_and_emit(h, words, 5, s, grouped, true))) return rc;
    if (P.stats_only) {
        HIP_TRY(h, hipMemcpyAsync(q.h_pc, q.d_pc, sizeof(PartCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    }
    P.bslots = bslots;
    if (!P.captured) { q.last_n = n; q.last_est = nodes_est; q.last_stage_want = stage_want; q.last_rows_floor = P.rows_floor; q.last_attempt = attempt; q.last_load = P.load_pct; }
    return GNDT_OK;
}

// Distinct nodes of a cloud, estimated in one pass (k_node_sketch, gndt_partition.hpp); waits for the answer.
int sketch_nodes(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, uint64_t* estimate) {
    *estimate = 0;
    if (n == 0) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "the node-count sketch");
    if (!h->d_sketch) {
        HIP_TRY(h, hipMalloc(&h->d_sketch, (size_t)kSketchRegs * 4));
        HIP_TRY(h, hipHostMalloc(&h->h_sketch, (size_t)kSketchRegs * 4));
    }
    HIP_TRY(h, hipMemsetAsync(h->d_sketch, 0, (size_t)kSketchRegs * 4, s));
    const float* p = static_cast<const float*>(xyz_dev);
    const uint32_t wgs = (uint32_t)std::min<uint64_t>(512, (n + 16383) / 16384);      // two resident per CU (64 KB of LDS each)
    if (stride_bytes == 12) hipLaunchKernelGGL(k_node_sketch<3>, dim3(wgs), dim3(kSketchThreads), 0, s, p, (uint64_t)n, grid_params(h), h->d_sketch);
    else hipLaunchKernelGGL(k_node_sketch<4>, dim3(wgs), dim3(kSketchThreads), 0, s, p, (uint64_t)n, grid_params(h), h->d_sketch);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_sketch, h->d_sketch, (size_t)kSketchRegs * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    // HyperLogLog (Flajolet et al. 2007): E = alpha m^2 / sum 2^-M[j]; linear counting while many registers are still empty
    const double m = (double)kSketchRegs;
    double sum = 0.0;
    uint32_t zeros = 0;
    for (int j = 0; j < kSketchRegs; ++j) { sum += std::ldexp(1.0, -(int)h->h_sketch[j]); zeros += h->h_sketch[j] == 0u; }
    double e = 0.7213 / (1.0 + 1.079 / m) * m * m / sum;
    if (e <= 2.5 * m && zeros) e = m * std::log(m / (double)zeros);
    *estimate = (uint64_t)(e + 0.5);
    return GNDT_OK;
}

// Start a PARTITION build (attempt 0) and leave it pending.  `records`: the input is 16-B records {x, y, z, index word} whose
// index words (point indices below `index_range`, weight flags included) are taken as they are (gndt_build_records_device).
int partition_begin(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, bool records, uint64_t index_range,
                    const void* records2, size_t n2) {
    auto& q = h->part;
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (records && stride_bytes != 16) { h->err = "records are 16 bytes"; return GNDT_ERR_INVALID; }
    if (n2 && !records) { h->err = "a second segment needs records"; return GNDT_ERR_INVALID; }
    const size_t n_first = n;
    n += n2;                           // (sizes below are those of the whole input)
    if (n >= 0x7FFFFFFFull || index_range >= 0x7FFFFFFFull) return -1;   // bit 31 of the record index word carries the weight flag: atomic path beyond 2^31 points
    int rc;
    if ((rc = ensure_words(h, ((index_range ? (size_t)index_range : n) + 31) / 32 + 1))) return rc;
    if ((rc = ensure_part_counters(h))) return rc;
    auto& P = h->pending;
    P = gndt_handle::Pending{};
    P.xyz = xyz_dev; P.n = n_first; P.stride = stride_bytes; P.s = s; P.attempt = 0;
    P.records = records; P.index_range = index_range; P.xyz2 = records2; P.n2 = n2;
    P.gp = grid_params(h);
    // What the last build of a cloud of this size needed (larger tables, a doubled estimate) is where this one starts: without
    // it every build of such a cloud would first fail with the small tables and be run twice.
    const bool similar = q.good_n && n <= 2 * q.good_n && 2 * n >= q.good_n;
    if (similar && q.good_slots == 1024) P.attempt = 1;
    if (similar && q.good_load) P.load_pct = q.good_load;        // (a small cloud that needed a lower table load)
    // expected node count: the caller's hint, else what the previous build of this handle found, else n/4
    // A first build without a hint guesses n / 4 nodes — unless strategy AUTO has just sampled this cloud's locality (64 tiles of
    // 2048 consecutive points: points per distinct node and tile): with locality, n / ratio bounds the node count from above (a node
    // met in two tiles counts twice) and is the better first guess (round 4: the 200 k-point campus frame has 0.37 nodes per point;
    // its first build on a fresh handle ran three times, 1.56 ms; a uniform cloud measures ratio 1 and keeps n / 4).
    uint64_t first_guess = std::max<uint64_t>(n / 4, 1024);
    if (h->tile_ratio_seen >= 2.0 && h->tile_choice_n && n <= h->tile_choice_n + h->tile_choice_n / 4 && n + n / 4 >= h->tile_choice_n)
        first_guess = std::max<uint64_t>(1024, (uint64_t)((double)n / h->tile_ratio_seen * 1.15));
    P.nodes_est = h->P.max_nodes_hint ? h->P.max_nodes_hint : (q.nodes_learned ? q.nodes_learned : first_guess);
    P.est_reliable = h->P.max_nodes_hint != 0 || q.nodes_learned != 0;      // (not the n / 4 guess of a first build)
    // A fresh handle without a hint: count the nodes instead of guessing them (one pass, a HyperLogLog sketch: ~0.8 % standard
    // error; the estimate carries 6 % on top).  Not under hipGraph capture (the answer is awaited), not for records (their
    // producer knows), not for small clouds (a re-run costs them less than the wait).
    if (!P.est_reliable && !records && n >= (1u << 20) && tuning().sketch) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            uint64_t est = 0;
            if ((rc = sketch_nodes(h, xyz_dev, n_first, stride_bytes, s, &est))) return rc;
            if (est) {
                P.nodes_est = std::max<uint64_t>(1024, est + est / 16);
                P.est_reliable = true;
                if (tuning().verbose) fprintf(stderr, "[gndt] node sketch: ~%llu nodes in %zu points\n", (unsigned long long)est, n);
            }
        }
    }
    if (similar && q.good_slots == 1024) P.nodes_est = std::max<uint64_t>(P.nodes_est, q.good_est);   // (an estimate that had to be doubled)
    P.est0 = P.nodes_est;
    P.rows_floor = h->P.max_nodes_hint ? h->P.max_nodes_hint + h->P.max_nodes_hint / 8
                                       : std::max<uint64_t>(4096, P.est_reliable ? P.nodes_est + P.nodes_est / 8 : n / 4);
    P.stage_want = std::max<uint64_t>(q.stage_cap, P.rows_floor);
    const int prev_strategy = h->last_strategy;
    h->last_strategy = GNDT_STRATEGY_PARTITION;
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        P.captured = cap != hipStreamCaptureStatusNone;
        P.captured_gen = h->realloc_gen;
    }
    if (P.captured && q.last_n == n && q.last_est) {      // (recorded: sized like the eager build of this cloud size before it)
        P.nodes_est = q.last_est; P.est0 = q.last_est; P.attempt = q.last_attempt; P.load_pct = q.last_load;
        P.stage_want = std::max<uint64_t>(q.stage_cap, q.last_stage_want);
        P.rows_floor = q.last_rows_floor;
    }
    rc = partition_launch(h, P);
    if (rc) { h->last_strategy = prev_strategy; return rc; }
    if (h->cur_capture_id) {                       // (recorded: what the replays of this call will have run, and with which buffers)
        auto& rec = h->captures[h->cur_capture_id % 32];
        rec.partition = true; rec.two_level = P.two_level; rec.one_level = P.one_level;
        rec.realloc_gen = h->realloc_gen; rec.table_gen = h->table_gen;
    }
    P.active = true;
    h->results_valid = false;
    h->map_in_table = false;
    h->stream_pos = index_range ? index_range : n;
    h->last_stream = s;
    return GNDT_OK;
}

// A PARTITION build captured in a hipGraph is replayed without the host: nobody starts a pending build, nobody looks at its flags.
// They come home all the same (k_emit_rows stores them into the pinned mirrors), so the call that waits for the stream can see
// that the build it finds on the device — necessarily a replay of the last one resolved here, while the handle's serial is still
// that build's — ran out of room in an LDS table, a partition region or the staging rows.  It cannot be run again here as a
// build the host launched would be: more room means new buffers, and the graph holds the old ones.  So it is REPORTED
// (GNDT_ERR_CAPACITY, what a captured gndt_update does): the caller builds that cloud eagerly and captures again.  Without this
// such a replay came back SHORT and unreported (tools/fuzz_graph.py, round 3).  Called with the stream idle.
int partition_recheck_after_replay(gndt_handle* h) {
    auto& P = h->pending;
    auto& q = h->part;
    // Keyed on the serial alone: while the handle still shows the build resolved here, whatever is on the device is that build or
    // a replay of it.  (Round 3 also required results_valid — after ONE reported replay every later one went unchecked, a short map
    // came back as GNDT_OK with the counts of the last good build, and a good replay after a bad one could not be exported.)
    if (P.active || !P.done_serial || P.done_serial != h->result_serial || P.stats_only || P.records || h->map_in_table || !q.h_pc ||
        (!h->results_valid && !P.replay_failed))
        return GNDT_OK;
    if (!(q.h_pc->part_overflow | q.h_pc->lds_overflow | q.h_pc->stage_overflow)) {
        if (P.replay_failed) { P.replay_failed = false; h->results_valid = true; }      // this replay fitted: its map is the handle's map again
        return GNDT_OK;
    }
    h->results_valid = false;
    P.replay_failed = true;
    h->err = "a build replayed from a hipGraph ran out of room (LDS tables " + std::to_string(q.h_pc->lds_overflow) + ", partition regions " +
             std::to_string(q.h_pc->part_overflow) + ", staging rows " + std::to_string(q.h_pc->stage_overflow) +
             "): the capture is sized for the cloud it was recorded on — build this cloud eagerly (gndt_build_device + gndt_sync), then capture again";
    return GNDT_ERR_CAPACITY;
}

// Wait for the pending build and look at its flags; re-run it with more room while they ask for it (the input
// must still be valid: it is the caller's until gndt_sync / gndt_export returns).
// Does the map this build has just finished say that the next cloud of its size may take BLOCKED buckets (gndt_blocked.hpp)?  Looked at
// once per cloud size and handle (one small kernel over the result keys and a host wait), for clouds whose neighbouring records rarely
// share a node.  Yes if the occupied key box is at most 32 levels high and blocks of 512 nodes (2^shz levels x the columns that leave)
// hold a few thousand points each, none of them dominated by one node.  A blocked build that meets a record outside the box or an
// overflowing region says no for good (partition_resolve).
static void blocked_decide(gndt_handle* h, const gndt_handle::Pending& P) {
    auto& q = h->part;
    const uint64_t n = P.n + P.n2;
    if (!tuning().blocked || P.stats_only || P.records || P.captured || !P.two_level || P.blocked || h->ever_captured) return;
    if (h->P.strategy != GNDT_STRATEGY_AUTO && h->P.strategy != GNDT_STRATEGY_PARTITION && h->P.strategy != GNDT_STRATEGY_PARTITION_TWO_LEVEL) return;
    if (q.blk_state != 0 && n <= q.blk_n + q.blk_n / 4 && n + n / 4 >= q.blk_n) return;      // (decided for clouds of this size)
    q.blk_state = -1; q.blk_n = n;
    if (!(q.pair_ratio >= 0.0 && q.pair_ratio <= 0.02)) return;                                // (clouds with locality: hot buckets, the pair folding)
    const uint64_t nodes = h->h_cnt->num_nodes;
    if (nodes < 1024 || nodes > h->out_cap) return;
    hipStream_t s = P.s;
    if (!q.d_extent) {
        if (hipMalloc(&q.d_extent, 8 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipHostMalloc(&q.h_extent, 8 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return; }
    }
    for (int k = 0; k < 8; ++k) q.h_extent[k] = k < 3 ? 0xFFFFFFFFu : 0u;
    if (hipMemcpyAsync(q.d_extent, q.h_extent, 8 * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); return; }
    hipLaunchKernelGGL(k_key_extent, dim3(64), dim3(1024), 0, s, (const int32_t*)h->out.sx, (const int32_t*)h->out.sy, (const int32_t*)h->out.sz,
                       (const uint32_t*)h->out.count, (uint32_t)nodes, q.d_extent);
    if (hipMemcpyAsync(q.h_extent, q.d_extent, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    const int64_t bias = 1ll << 30;
    const int64_t x0 = (int64_t)q.h_extent[0] - bias, y0 = (int64_t)q.h_extent[1] - bias, z0 = (int64_t)q.h_extent[2] - bias;
    const int64_t X = (int64_t)q.h_extent[3] - bias - x0 + 1, Y = (int64_t)q.h_extent[4] - bias - y0 + 1, Z = (int64_t)q.h_extent[5] - bias - z0 + 1;
    const uint64_t biggest = q.h_extent[6];
    if (X <= 0 || Y <= 0 || Z <= 0 || Z > 32) return;
    int shz = 0;
    while ((1 << shz) < Z) ++shz;
    // taller blocks (more level bits than the box needs) mean fewer columns per block and more blocks: the first height whose blocks
    // hold at most ~5 000 points
    for (; shz <= 5; ++shz) {
        const int shx = (9 - shz + 1) / 2, shy = 9 - shz - shx;
        const uint64_t nx = (uint64_t)((X + (1ll << shx) - 1) >> shx), ny = (uint64_t)((Y + (1ll << shy) - 1) >> shy);
        const uint64_t B = nx * ny;
        if (B < 64 || B > (uint64_t)kMaxFan * kMaxFan) continue;
        const uint64_t per_block = n / B;
        if (per_block > 5000) continue;
        if (per_block < 1500) break;                    // (sparse boxes: mostly empty blocks)
        if (biggest * 4 > per_block) break;            // (a node that holds a quarter of an average block's points: hot blocks)
        // how full the box is: nodes per slot of the blocks (an evenly filled box fills its blocks alike)
        if (nodes * 8 < B * 512) break;                 // (less than an eighth of the slots used: the box is mostly air)
        // one block of margin on every side and the levels centred in the block's height: the next cloud of the same scene (other
        // noise, a few points further out) still fits the box
        const int64_t zpad = ((1ll << shz) - Z) / 2;
        q.blk_map = BlockMap{1, (int)(x0 - (1ll << shx)), (int)(y0 - (1ll << shy)), (int)(z0 - zpad), shx, shy, shz, (int)nx + 2, (int)ny + 2};
        q.blk_buckets = (uint32_t)((nx + 2) * (ny + 2));
        if (q.blk_buckets > (uint64_t)kMaxFan * kMaxFan) { q.blk_map.on = 0; continue; }
        q.blk_state = 1;
        if (tuning().verbose)
            fprintf(stderr, "[gndt] blocked buckets for clouds of ~%llu points: box %lld x %lld x %lld, blocks of %d x %d columns x %d levels, %llu buckets of ~%llu points\n",
                    (unsigned long long)n, (long long)X, (long long)Y, (long long)Z, 1 << shx, 1 << shy, 1 << shz, (unsigned long long)B, (unsigned long long)per_block);
        return;
    }
}

int partition_resolve(gndt_handle* h) {
    auto& P = h->pending;
    if (!P.active) return GNDT_OK;
    auto& q = h->part;
    const int env_slots = tuning().bucket_slots;
    int rc = GNDT_OK;
    for (;;) {
        if (hipStreamSynchronize(P.s) != hipSuccess) { P.active = false; h->err = "hipStreamSynchronize failed"; return GNDT_ERR_HIP; }
        bool again = false;
        if (P.two_level && P.mean1 > 0) {                      // remember how uneven the level-1 regions of the latest cloud were
            q.fill1_ratio = q.h_pc->max_fill1 / P.mean1;
            if (tuning().verbose)
                fprintf(stderr, "[gndt] two-level partition: fullest level-1 region %.2fx the mean, overflow %u\n",
                        q.h_pc->max_fill1 / P.mean1, q.h_pc->part_overflow);
        }
        if (tuning().verbose && (q.h_pc->part_overflow | q.h_pc->lds_overflow | q.h_pc->stage_overflow))
            fprintf(stderr, "[gndt] build of %zu points, attempt %d (%d-slot tables, %llu nodes expected): region overflow %u, table overflow %u, "
                            "staging overflow %u -> re-run\n", P.n, P.attempt, P.bslots, (unsigned long long)P.nodes_est, q.h_pc->part_overflow,
                    q.h_pc->lds_overflow, q.h_pc->stage_overflow);
        if (P.blocked && (q.h_pc->blk_miss | q.h_pc->part_overflow)) {
            // a record outside the box the blocks were laid out for, or blocks so uneven that a region overflowed: this cloud is
            // not the dense box the last one was — the same build with hashed buckets, and the handle forgets the box
            q.blk_state = -1;
            P.no_block = true;
            --P.attempt;
            again = true;
        } else if (q.h_pc->part_overflow) {                    // a region of the two-level partition was too small: same table
            if (P.one_level) q.one_level_ok = false;           // (one-level: a bucket outgrew its fixed room: counting partition from now on)
            else ++q.two_level_failures;                       // size and estimate again (level-1 regions sized from the fullest
            --P.attempt;                                       // one seen; after two failures the exact counting partition)
            again = true;
        } else if (q.h_pc->lds_overflow) {                            // some bucket holds too many nodes for its LDS table:
            const bool without_retry = P.bslots == 512 && !P.retry_pass && tuning().retry_pass != 0 && !env_slots;
            q.retry_pass = true;                                      // (from now on overflowing buckets of this handle's clouds are done again locally)
            // (Lowering the average table load instead — more, smaller buckets — was measured: 1.19 ms against 0.80 ms with the
            // 1024-slot tables on the 3 M-node variant of the bench scene, and it does not help a hot column at all.)
            // Small clouds are the exception: a 200 k-point frame fills 1024-slot tables (one workgroup per CU) with ~140
            // buckets for 256 CUs; more 512-slot buckets keep the chip busy, so they first get a lower table load — as long as
            // the 1024-slot tables would leave CUs idle (under 384 buckets): a million points with half a million nodes get
            // 1 000 of them, and 3 600 tiny 512-slot buckets instead cost 30-45 % (size sweep, round 3).
            const uint64_t b1024 = buckets_for(P.n + P.n2, P.nodes_est, 1024, P.load_pct ? P.load_pct : q.load_pct);
            if (without_retry) {
                --P.attempt;                                     // the same tables once more, this time with the second pass behind them
            } else if (P.bslots == 512 && !env_slots && P.n + P.n2 <= (1u << 20) && b1024 < 384 && (P.load_pct ? P.load_pct : q.load_pct) > 35) {
                P.load_pct = 35;
                --P.attempt;                                     // (stay on the 512-slot tables)
            } else if (P.attempt >= 1 || env_slots) {            // (attempt 0 -> 1 only switches to the 1024-slot table)
                // More buckets: by 1.4x while only a few tables overflow (a cloud of tall columns — 10 M points at z = 0.1 m: one
                // table in 5 500 — runs 15 % faster on 7 700 buckets than on the 11 000 a doubling gave it: fewer, fuller buckets
                // amortise the per-bucket phases), doubling when many do.
                const bool few = q.h_pc->lds_overflow * 64u <= q.last_buckets;
                P.nodes_est = few ? P.nodes_est + (P.nodes_est * 2) / 5 : P.nodes_est * 2;
            }
            again = true;
        } else if (q.h_pc->stage_overflow) {                   // num_nodes kept counting: it is the true total
            const uint64_t true_nodes = h->h_cnt->num_nodes;
            const uint64_t want = true_nodes + true_nodes / 8 + 1024;
            // More rows than the staging arrays held is not a failure of the TABLES: the same attempt once more, with room for what
            // was counted — and with the count as the node estimate where the estimate fell short of it (round 6: a first build that
            // guessed n / 4 nodes for a cloud of n / 2 went on to 1024-slot tables with the same wrong estimate and ran a third time)
            if (want > P.stage_want) --P.attempt;
            P.stage_want = want;
            P.rows_floor = want;
            if (true_nodes > P.nodes_est) { P.nodes_est = true_nodes + true_nodes / 16; P.est_reliable = true; }
            again = true;
        }
        if (again && P.captured) {
            // The build was recorded into a hipGraph and what ran was its first replay.  Running it again here with more room would
            // re-allocate the buffers the graph holds (round 3: exactly that happened — the re-run gave the right map and the NEXT
            // replay wrote through freed pointers: a GPU memory fault; tools/fuzz_graph.py).  Reported instead, like a later replay.
            P.active = false;
            h->results_valid = false;
            // (round 5: the capture itself is sound — only THIS cloud did not fit — so the handle is left as after any reported replay:
            //  a later replay that fits is validated by partition_recheck_after_replay and can be exported.  Until then this first
            //  failure left later, fitting replays "no finished build": DESIGN §8.7 of round 4.)
            ++h->result_serial;
            P.done_serial = h->result_serial;
            P.replay_failed = true;
            h->table_dirty = false;
            h->err = "a build replayed from a hipGraph ran out of room (LDS tables " + std::to_string(q.h_pc->lds_overflow) + ", partition regions " +
                     std::to_string(q.h_pc->part_overflow) + ", staging rows " + std::to_string(q.h_pc->stage_overflow) +
                     "): the capture is sized for the cloud it was recorded on — build this cloud eagerly (gndt_build_device + gndt_sync), then capture again";
            return GNDT_ERR_CAPACITY;
        }
        if (!again) {
            q.nodes_learned = (uint64_t)h->h_cnt->num_nodes + h->h_cnt->num_nodes / 5;
            // the second pass stays on for as long as it has work (and comes back through the re-run below when it is missed);
            // many queued buckets mean the tables are too full for this cloud's columns: the next build takes more, emptier ones
            q.retry_seen = P.retry_pass ? q.h_pc->lds_retry : 0u;
            if (P.retry_pass) q.retry_pass = q.h_pc->lds_retry != 0u;
            if (P.retry_pass && (uint64_t)q.h_pc->lds_retry * 8u > q.last_buckets) q.nodes_learned += q.nodes_learned / 4;
            if (tuning().verbose && P.retry_pass && q.h_pc->lds_retry)
                fprintf(stderr, "[gndt] %u of %u buckets went through the 1024-slot second pass\n", q.h_pc->lds_retry, q.last_buckets);
            if ((P.n + P.n2) && !P.blocked) q.pair_ratio = 2.0 * (double)q.h_pc->pairs / (double)(P.n + P.n2);      // (the blocked kernel does not count pairs)
            // the larger tables are remembered only if the small ones failed although the estimate was adequate (a first build
            // without a hint guesses n / 4 nodes: its failure says nothing about the cloud)
            const bool est_was_fine = P.est_reliable && P.est0 >= (uint64_t)h->h_cnt->num_nodes;
            q.good_slots = (P.bslots == 1024 && !est_was_fine) ? 0 : P.bslots;
            q.good_est = P.nodes_est; q.good_n = P.n + P.n2; q.good_load = P.load_pct;
            if (!P.stats_only) {
                h->results_valid = true;
                ++h->result_serial;
                P.done_serial = h->result_serial;
            }
            h->table_dirty = false;
            P.active = false;
            blocked_decide(h, P);
            return GNDT_OK;
        }
        rc = -1;
        ++q.retries_total;
        if (++P.attempt < 5) rc = partition_launch(h, P);
        if (rc == GNDT_OK) continue;
        P.active = false;
        if (rc != -1 || P.stats_only) return rc;   // (a statistics-only run reports -1: its caller falls back)
        // does not fit the LDS-resident pipeline (too many nodes per bucket): same result via the atomic path
        if (P.records) { const gndt_handle::Pending rec = P; return build_atomic(h, nullptr, 0, 16, rec.s, false, &rec); }   // (a copy: the build resets h->pending)
        return build_atomic(h, P.xyz, P.n, P.stride, P.s);
    }
}


}  // namespace gndt_host

extern "C" {

int gndt_build_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;          // a build still pending is being replaced: nobody will ask for its result
    { const int urc = use_stream(h, s); if (urc) return urc; }
    next_event_set(h);
    int strategy = h->P.strategy;
    if (strategy == GNDT_STRATEGY_AUTO) {
        strategy = (n >= (1u << 16)) ? GNDT_STRATEGY_PARTITION : GNDT_STRATEGY_ATOMIC;
        // Clouds that keep their scan order can be built in ONE pass (strategy TILE) when a flush of a workgroup's private
        // table carries many points per node.  That ratio is MEASURED on a sample of this cloud (64 tiles of 2048
        // consecutive points, ~20 us + one host wait); the answer is kept for the handle and re-measured when the cloud
        // size changes by a quarter or after 64 builds.
        if (n >= (1u << 16)) {             // (a 200 k-point depth-camera frame at 0.5 m cells: 400 points per node, TILE 0.088 ms against PARTITION 0.145)
            // (the sample ends in a host wait, which a stream under graph capture cannot take: a captured build keeps the
            //  handle's last answer, PARTITION if there is none yet)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s, &cap);
            auto adopt = [&](double ratio, uint64_t n_of) {
                h->tile_choice = ratio >= tuning().tile_ratio ? 1 : 0;
                h->tile_choice_n = n_of; h->tile_choice_age = 0; h->tile_ratio_seen = ratio;
                if (tuning().verbose) fprintf(stderr, "[gndt] locality sample: %.1f points per partial -> %s\n", ratio, h->tile_choice ? "TILE" : "PARTITION");
            };
            // (the answer of a sample an earlier build did not wait for, if it is there by now)
            { double r0 = 0.0; if (h->sample_pending && locality_sample_take(h, false, &r0)) adopt(r0, h->sample_n); }
            const bool stale = cap == hipStreamCaptureStatusNone && !h->sample_pending &&
                               (h->tile_choice < 0 || ++h->tile_choice_age >= 64 ||
                                n > h->tile_choice_n + h->tile_choice_n / 4 || n + n / 4 < h->tile_choice_n);
            if (stale) {
                // A RE-sample (the cloud size changed by a quarter, or 64 builds have passed) on a handle that has room for whatever the
                // cloud turns out to be does not WAIT for the answer (~50 us): the kernel and its copy ride in front of this build,
                // which keeps the handle's last choice, and the next build finds the answer.
                // (A handle whose node estimate does not come from the sample — a hint, an earlier build — never needs to wait either.)
                // ... and not for a handle's FIRST sample (round 6, measured: a depth-camera frame — configs[0]'s own kind of cloud — built
                // by PARTITION because nobody had looked yet took 0.31 ms with a re-run where the awaited sample + TILE take 0.10)
                const bool defer = h->tile_choice >= 0 &&
                                   ((h->part.stage_cap >= n / 2 && h->part.stage_cap > 0) || h->part.nodes_learned != 0 || h->P.max_nodes_hint != 0);
                rc = locality_sample_begin(h, xyz_dev, n, stride_bytes, 64, s);
                if (rc) return rc;
                if (!defer) { double ratio = 0.0; if (locality_sample_take(h, true, &ratio)) adopt(ratio, n); }
                else { h->tile_choice_age = 0; h->tile_choice_n = n; }                  // (the last choice until the answer is in)
            }
            if (h->tile_choice == 1) strategy = GNDT_STRATEGY_TILE;
        }
    }
    if (strategy == GNDT_STRATEGY_TILE) return build_atomic(h, xyz_dev, n, stride_bytes, s, true);
    if (strategy == GNDT_STRATEGY_PARTITION || strategy == GNDT_STRATEGY_PARTITION_EXACT || strategy == GNDT_STRATEGY_PARTITION_TWO_LEVEL) {
        // launched, not awaited: gndt_sync / gndt_export* (or whatever needs the result next) waits, checks the
        // overflow flags and re-runs with more room if needed.  xyz_dev stays the caller's until then.
        rc = partition_begin(h, xyz_dev, n, stride_bytes, s);
        if (rc != -1) return rc;
        // does not fit the LDS-resident pipeline: same result via the atomic path
    }
    return build_atomic(h, xyz_dev, n, stride_bytes, s);
}

int gndt_build(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    rc = stage_host_input(h, xyz_host, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    rc = gndt_build_device(h, h->stage, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

// Every buffer a build of up to max_points points and max_nodes nodes can ask for, whatever the strategy and whatever a re-run
// would want (both LDS table sizes, the lowest table load, node estimates up to twice max_nodes; level-1 regions at their 4 n
// ceiling; the exact partition's histogram; the node table of strategies ATOMIC / TILE; staging rows, result rows, order arrays).
// After it a build of that size allocates nothing: it can be captured into a hipGraph without an eager warm-up, and a replay
// never finds its buffers moved (7.7 % of round 3's captures were refused — root cause: ONE capture in
// which a buffer had to grow; hipFree / hipMalloc on a capturing stream invalidates the capture and HIP then refuses every later
// capture of the process).
int gndt_reserve(gndt_handle* h, uint64_t max_points, uint64_t max_nodes) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    h->capturing = false;
    if (max_points == 0) { h->err = "gndt_reserve: max_points must be > 0"; return GNDT_ERR_INVALID; }
    if (max_points >= 0x7FFFFFFFull) { h->err = "gndt_reserve: more than 2^31 points"; return GNDT_ERR_INVALID; }
    if (h->pending.active) { const int prc = partition_resolve(h); if (prc) return prc; }
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));              // (buffers may move: nothing may still use them)
    const uint64_t n = max_points;
    // No node count given: a first build sizes itself from n / 4 nodes — or, when strategy AUTO's locality sample finds two or more
    // points per node and tile, from 1.15 n / ratio <= 0.575 n (partition_begin): what is reserved has to cover that too (round 6:
    // the campus frame, 0.37 nodes per point, grew its staging arrays in the first build after gndt_reserve(n, 0): 0.26 ms of a
    // 0.45 ms first call)
    if (!max_nodes) max_nodes = std::max<uint64_t>(std::max<uint64_t>(n / 4, (n * 3) / 10), 1024);
    max_nodes = std::max<uint64_t>(max_nodes, h->P.max_nodes_hint);      // (the hint is what sizes the first attempt of every build)
    auto& q = h->part;
    int rc;
    if ((rc = ensure_words(h, (std::max<uint64_t>(n, h->P.max_points_hint) + 31) / 32 + 1))) return rc;
    if ((rc = ensure_part_counters(h))) return rc;
    // rows: what a PARTITION build stages (estimate + 1/8, after two 1.4x raises) and what the table path bounds by its slots
    const uint64_t est_hi = 2 * max_nodes;
    const uint64_t rows = std::max<uint64_t>(std::max<uint64_t>(est_hi + est_hi / 8 + 1024, (uint64_t)cap_for_nodes(max_nodes) / 2 + 1),
                                             4096 + n / 4 + n / 32);      // (a build without a hint stages max(4096, n / 4) rows at least)
    if ((rc = ensure_stage(h, rows))) return rc;
    if ((rc = ensure_raw(h, q.stage_cap))) return rc;           // (the PARTITION strategies' staging rows)
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    // bucket count: the largest any attempt can ask for
    uint64_t Bmax = 16;
    for (int slots : {512, 1024})
        for (int load : {35, 60, 75}) Bmax = std::max(Bmax, buckets_for(n, est_hi, slots, load));
    Bmax = std::min<uint64_t>(Bmax, (uint64_t)kMaxFan * kMaxFan);
    if ((rc = ensure_cursors(h, Bmax))) return rc;
    constexpr uint64_t kTile1 = (uint64_t)kTileThreads * kTilePer1;
    if ((rc = grow_buf(h, q.recs, q.rec_cap, 2 * n + n / 8 + 2048ull * Bmax + 4096))) return rc;                    // two-level (covers the exact partition's n)
    if ((rc = grow_buf(h, q.recs1, q.rec1_cap, 4 * n + (1u << 24) + (uint64_t)kMaxFan * (4096 + 2 * kTile1)))) return rc;   // level-1 regions at their ceiling / one-level rooms
    const uint64_t Bexact = std::min<uint64_t>(Bmax, kMaxBuckets);
    if ((rc = grow_buf(h, q.hist, q.hist_cap, (uint64_t)kPartWgs * Bexact))) return rc;
    if (Bexact > q.bucket_cap) {
        release_device(h, q.totals);
        release_device(h, q.bucket_base);
        q.totals = q.bucket_base = nullptr; q.bucket_cap = 0;
        HIP_TRY(h, hipMalloc(&q.totals, (size_t)Bexact * 4));
        HIP_TRY(h, hipMalloc(&q.bucket_base, ((size_t)Bexact + 1) * 4));
        q.bucket_cap = (uint32_t)Bexact;
    }
    // strategies ATOMIC / TILE (and gndt_update*): the node table
    if ((rc = reserve_table(h, std::max<uint64_t>(max_nodes, std::max<uint64_t>(1024, n / 4)), h->own_stream))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    // (rows_bound of a table-path finalisation follows the table: cap / 2 + 1)
    if ((rc = ensure_stage(h, std::max<uint64_t>(1024, h->cap / 2 + 1)))) return rc;
    if ((rc = ensure_raw(h, q.stage_cap))) return rc;
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    return GNDT_OK;
}

int gndt_debug_retry_count(gndt_handle* h, uint64_t* retries) {
    if (!h || !retries) return GNDT_ERR_INVALID;
    *retries = h->part.retries_total;
    return GNDT_OK;
}

int gndt_debug_second_pass_buckets(gndt_handle* h, uint64_t* buckets) {
    if (!h || !buckets) return GNDT_ERR_INVALID;
    *buckets = h->part.retry_seen;
    return GNDT_OK;
}

int gndt_debug_enable_stamps(int on) { tuning_force_stamps(on != 0); return GNDT_OK; }
int gndt_debug_set_option(int option, double value) { return tuning_set_option(option, value); }

int gndt_debug_set_fp_bits(int bits) { tuning_force_fp_bits(bits); return GNDT_OK; }

int gndt_debug_fp_clashes(gndt_handle* h, uint64_t* buckets) {
    if (!h || !buckets) return GNDT_ERR_INVALID;
    *buckets = h->part.h_pc ? h->part.h_pc->fp_clashes : 0;
    return GNDT_OK;
}

int gndt_debug_bucket_phases(gndt_handle* h, double cycles_out[10], uint32_t* buckets_out) {
    if (!h || !cycles_out) return GNDT_ERR_INVALID;
    auto& q = h->part;
    if (!q.dbg || !q.last_buckets) { h->err = "no stamps: call gndt_debug_enable_stamps(1) before a PARTITION build"; return GNDT_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    std::vector<unsigned long long> t((size_t)q.last_buckets * 16);
    HIP_TRY(h, hipMemcpy(t.data(), q.dbg, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 10; ++k) cycles_out[k] = 0.0;
    for (uint32_t b = 0; b < q.last_buckets; ++b) {
        const unsigned long long* s = &t[(size_t)b * 16];
        for (int k = 0; k < 6; ++k) if (s[k + 1] > s[k]) cycles_out[k] += (double)(s[k + 1] - s[k]);   // (a kernel stamps only the boundaries it has)
        // sub-phases of the accumulate phase, summed over the bucket's chunks: load wait, classify, scan+scatter, reduce
        for (int k = 0; k < 4; ++k) cycles_out[6 + k] += (double)s[8 + k];
    }
    for (int k = 0; k < 10; ++k) cycles_out[k] /= q.last_buckets;
    if (buckets_out) *buckets_out = q.last_buckets;
    return GNDT_OK;
}

}  // extern "C"

// ---- gndt_warmup ----
namespace {
// A synthetic cloud with the statistics of nothing in particular: 100 x 100 columns of four levels around the origin, shuffled
// (a counter-based hash per coordinate), so that every kernel of every strategy has nodes, columns and slopes to work on.
__global__ void k_warm_fill(float* __restrict__ xyz, uint32_t n, uint32_t stride_floats, float gl, float zl) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t a = i * 0x9E3779B1u + 0x7F4A7C15u; a ^= a >> 15; a *= 0x85EBCA77u; a ^= a >> 13;
        uint32_t b = a * 0xC2B2AE3Du + 0x165667B1u; b ^= b >> 16;
        uint32_t c = b * 0x27D4EB2Fu + 0x9E3779B1u; c ^= c >> 15;
        float* p = xyz + (size_t)i * stride_floats;
        p[0] = ((float)(a >> 8) * (1.0f / 16777216.0f) - 0.5f) * 100.0f * gl;
        p[1] = ((float)(b >> 8) * (1.0f / 16777216.0f) - 0.5f) * 100.0f * gl;
        p[2] = ((float)(c >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.0f * zl;
        if (stride_floats == 4) p[3] = 0.f;
    }
}
std::atomic<uint64_t> g_warm_points[64];      // per device: the largest cloud size the process has warmed up with (0: none yet)
}  // namespace

int gndt_warmup(gndt_handle* h, uint64_t expected_points) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(h->own_stream, &cap);
        if (cap != hipStreamCaptureStatusNone) { h->err = "gndt_warmup: not under stream capture"; return GNDT_ERR_INVALID; }
    }
    uint64_t n = expected_points ? expected_points : (h->P.max_points_hint ? h->P.max_points_hint : 200000);
    const uint64_t reserve_points = expected_points ? expected_points : h->P.max_points_hint;
    n = std::min<uint64_t>(std::max<uint64_t>(n, 1u << 16), 4u << 20);     // (>= 2^16: strategy AUTO's partition side; <= 4 M: the two-level kernels)
    const bool known_device = h->device >= 0 && h->device < 64;
    int rc = GNDT_OK;
    // (sizes within a quarter of one another make the same choices — partition levels, fan-outs, tile sizes, scans: what AUTO's own
    //  "similar cloud" rule uses — so a second handle of the process, or a second call, finds the code loaded and skips the runs)
    const uint64_t seen = known_device ? g_warm_points[h->device].load(std::memory_order_acquire) : 0;
    const bool loaded = seen && n <= seen + seen / 4 && n + n / 4 >= seen;
    if (!loaded) {
        hipStream_t s = h->own_stream;
        float* cloud = nullptr;
        HIP_TRY(h, hipMalloc(&cloud, (size_t)n * 16));
        auto run = [&](int strategy, uint32_t stride_floats, bool update, bool flood) -> int {
            gndt_params tp = h->P;
            tp.strategy = strategy; tp.max_points_hint = 0; tp.max_nodes_hint = 0;
            gndt_handle* t = nullptr;
            int r = create_handle(&tp, &t, s);
            if (r) { h->err = std::string("gndt_warmup: ") + gndt_last_error(nullptr); return r; }
            const float origin[3] = {0.f, 0.f, 0.f};
            hipLaunchKernelGGL(k_warm_fill, dim3(512), dim3(256), 0, s, cloud, (uint32_t)n, stride_floats, tp.grid_len, tp.z_len);
            r = gndt_set_origin(t, origin);
            // two builds: the first sizes its tables from a guess or a sketch, the second from what the first learnt — both code paths
            for (int k = 0; k < 2 && !r; ++k) {
                r = gndt_build_device(t, cloud, (size_t)n, stride_floats * 4u, nullptr);
                if (!r) r = gndt_sync(t, nullptr, nullptr, nullptr);
            }
            if (!r && update) {             // a frame on top (the table path of gndt_update*), eager and with deferred rows
                r = gndt_update_device(t, cloud, (size_t)std::min<uint64_t>(n, 1u << 17), stride_floats * 4u, nullptr);
                if (!r) r = gndt_sync(t, nullptr, nullptr, nullptr);
            }
            if (!r) { gndt_cells cells; r = gndt_export_device(t, &cells); }
            if (!r && flood) {
                const float goal[3] = {0.3f * tp.grid_len, 0.3f * tp.grid_len, 0.f};
                r = gndt_compute_cost(t, goal, nullptr, nullptr);
                if (r == GNDT_ERR_INVALID) r = GNDT_OK;      // (a goal that is no slope of the synthetic map: the kernels up to there have run)
            }
            if (r) h->err = std::string("gndt_warmup: ") + gndt_last_error(t);
            gndt_destroy(t);
            return r;
        };
        // what a build on THIS handle can take: its own strategy first (AUTO samples the locality, then partitions), then the
        // fallbacks it may be sent to (exact partition after an overflowing region, ATOMIC when a bucket does not fit), both input strides
        const int own = h->P.strategy;
        // (gndt_update* adds to a map of the table strategies only: the frame on top rides with the ATOMIC run)
        rc = run(own, 3, own == GNDT_STRATEGY_ATOMIC, true);
        if (!rc) rc = run(own, 4, false, false);
        for (int st : {GNDT_STRATEGY_PARTITION, GNDT_STRATEGY_PARTITION_EXACT, GNDT_STRATEGY_ATOMIC, GNDT_STRATEGY_TILE})
            if (!rc && st != own) rc = run(st, 3, st == GNDT_STRATEGY_ATOMIC, false);
        (void)hipStreamSynchronize(s);
        (void)hipFree(cloud);
        if (rc) return rc;
        if (known_device) g_warm_points[h->device].store(n, std::memory_order_release);
    }
    if (reserve_points) rc = gndt_reserve(h, reserve_points, h->P.max_nodes_hint);
    // (what a first build of THIS handle would still allocate: the locality sample's counters, their pinned mirror, its event)
    if (!rc) rc = ensure_sample_buffers(h);
    return rc;
}
