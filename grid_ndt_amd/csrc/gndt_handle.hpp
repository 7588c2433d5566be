// gndt_handle.hpp — the handle behind include/gndt.h and the host-side helpers its translation units share.
// Host code only owns memory, orders launches on a HIP stream and maps errors to status codes; the entry points are
// grouped by what they drive:
//   gndt_api_core.hip   handle life cycle, origin, sync / export, profiling, shared buffers
//   gndt_api_table.hip  strategy ATOMIC: the HBM node table (accumulate, update, finalize, statistics export / merge)
//   gndt_api_build.hip  strategy PARTITION: launch, pending-build resolution, gndt_build*
//   gndt_api_dist.hip   one global map from a sharded cloud (shard statistics, exchange, finalize from statistics)
//   gndt_api_cost.hip   cost-map flood over the finished grid
//   gndt_api_io.hip     input side (record unpack + NaN strip, gndt_build_cloud)
// There is NO CPU fallback: without a HIP device every compute entry point fails with GNDT_ERR_NO_DEVICE.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "gndt.h"

// Debug build (-DGNDT_POISON): every device allocation is filled with 0xA5 before it is handed out.  A fresh process gets zeroed
// pages from hipMalloc, which hides reads of memory the library never initialised; a process that has freed and reallocated for a
// while does not (tools/fuzz_graph.py found two such reads only after ~100 handles).  The tests run under this build as well.
#ifdef GNDT_POISON
namespace gndt_host {
inline hipError_t poison_malloc(void** p, size_t bytes) {
    const hipError_t e = (hipMalloc)(p, bytes);
    if (e == hipSuccess && bytes) { (void)hipMemset(*p, 0xA5, bytes); (void)hipDeviceSynchronize(); }
    return e;
}
}  // namespace gndt_host
#define hipMalloc(p, bytes) gndt_host::poison_malloc((void**)(p), (bytes))
#endif
#include "gndt_kernels.hpp"
#include "gndt_partition.hpp"
#include "gndt_cost.hpp"

using namespace gndt;   // host translation units of libgndt only: nothing else includes this header

struct gndt_handle {
    gndt_params P{};
    float origin[3] = {0, 0, 0};
    bool origin_set = false;
    int device = 0;
    hipStream_t own_stream = nullptr;
    bool stream_borrowed = false;      // gndt_warmup's temporary handles run on the stream of the handle they warm up (never destroyed by them)
    hipStream_t last_stream = nullptr;
    hipEvent_t xstream_ev = nullptr;   // orders work on a new stream behind what the previous one still runs (use_stream)
    // A build recorded into a hipGraph keeps the device pointers of its day.  Once a stream of this handle has been seen under
    // capture, buffers that are outgrown are not freed but RETIRED (kept until gndt_destroy): a replay of the old graph then writes
    // into memory that is still the handle's — its result is reported as stale (realloc_gen), but it cannot fault or corrupt
    // someone else's allocation (round 5, tools/fuzz_graph.py: "write access to a read-only page" 60 s into the first seed that
    // put eager table-family builds between the replays).
    bool ever_captured = false;  std::vector<void*> retired;
    uint32_t table_gen = 0;            // allocations of the node table so far (Counters::table_gen on the device)
    // Calls recorded into a hipGraph: each gets an id that the last kernel of the recorded work stores into the host's mirror of the
    // flags (PartCounters::capture_id) — so the call that next waits for the stream can tell that what it finds on the device is a
    // REPLAY (the host saw none of it), of which recorded call, and whether that call's buffers are still the handle's.
    struct CaptureRec { uint32_t id = 0; uint64_t realloc_gen = 0; uint32_t table_gen = 0; bool partition = false; bool two_level = false, one_level = false; };
    CaptureRec captures[32];           // (a ring: a handle with more than 32 live graphs reports the oldest ones as stale)
    uint32_t capture_seq = 0, cur_capture_id = 0;
    uint32_t replay_seen = 0;          // the capture id the last gndt_sync found in the mirror (0: the host's own call)
    uint64_t realloc_gen = 0;          // bumped whenever a device buffer of this handle is freed / reallocated: a build recorded into a hipGraph
                                       //   before that writes through the old pointers when replayed — noticed and reported (Pending::captured_gen)
    bool capturing = false;            // the stream of the call in progress is under hipGraph capture (use_stream): nothing may allocate,
                                       //   free or wait — such a call would not only fail, it INVALIDATES the capture (GNDT_NO_CAPTURE)

    // node table
    uint32_t cap = 0;  // slots, power of two
    uint64_t* keys = nullptr;
    NodeAcc* acc = nullptr;
    uint64_t* col_keys = nullptr;
    uint32_t* col_first = nullptr;
    SlotAux* aux = nullptr;
    // node list + ordering buffers (cap entries each)
    uint32_t* node_slot = nullptr;
    uint32_t* col_slot_of_node = nullptr;
    uint32_t* col_cnt = nullptr;
    uint32_t* col_head = nullptr;
    uint32_t* node_next = nullptr;
    void* ninfo = nullptr;          // gndt::NodeInfo[cap]: per node {first-seen, z level, mean z, flags} for the column walks (gndt_table.hpp)
    // incremental updates: slot -> node index, touch marks, the lists of touched nodes / columns
    uint32_t *index_of_slot = nullptr, *touch_epoch = nullptr, *col_epoch = nullptr, *touched = nullptr, *touched_cols = nullptr;
    bool incr_ok = false;       // the persistent staging rows / order of the table path describe the current map

    Counters* d_cnt = nullptr;
    Counters* h_cnt = nullptr;  // pinned

    // results
    uint64_t out_cap = 0;
    OutView out{};
    uint64_t res_nodes = 0, res_columns = 0, res_slopes = 0;
    bool results_valid = false;

    // stats export buffers
    uint64_t st_cap = 0;
    uint64_t* st_key = nullptr;
    double* st_sums = nullptr;
    uint32_t* st_count = nullptr;
    uint32_t* st_first = nullptr;

    // staging for host input
    void* stage = nullptr;
    size_t stage_bytes = 0;
    void* exp_host = nullptr;  uint64_t exp_rows = 0;   // gndt_export_host: pinned mirror of the result arrays
    void* bounce[2] = {nullptr, nullptr};          // pinned bounce buffers for large pageable inputs (stage_host_input)
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    // gndt_build_cloud: packed xyz of the unpacked, NaN-stripped cloud; pinned word for the valid count
    float* packed = nullptr;
    uint64_t packed_cap = 0;
    uint32_t* d_nvalid = nullptr;
    uint32_t* h_nvalid = nullptr;

    uint64_t stream_pos = 0;    // points accumulated since the last reset (host mirror of the device-side first_idx base)
    uint64_t nodes_bound = 0;   // host-side upper bound of the nodes in the table (no sync needed to size buffers)
    bool table_dirty = false;   // table holds nodes
    // small maps (k_small_finalize, gndt_table.hpp): the node count the last resolved table-path build had; cleared when the
    // one-workgroup finalisation met a map that was not small; whether its last launch was recorded under hipGraph capture
    uint32_t table_nodes_seen = 0;  bool small_ok = true;  bool small_captured = false;  bool small_used = false;
    // deferred-emit mode (gndt_set_deferred_emit): gndt_update* relabels the touched columns and stops; the rows in the reference's
    // dense order are produced by the next call that READS the map.  emit_pending: frames have been added since the rows were emitted
    bool defer_emit = false;  bool emit_pending = false;  uint64_t pending_words = 0;
    bool deferred_captured = false;    // the last deferred frame was recorded under hipGraph capture: replays add frames the host does not see

    // strategy PARTITION buffers (gndt_partition.hpp)
    struct Part {
        uint64_t rec_cap = 0;      float4* recs = nullptr;
        // two-level partition: level-1 regions, cursors of both levels, record ranges of the fine buckets
        uint64_t rec1_cap = 0;     float4* recs1 = nullptr;
        uint64_t cur_cap = 0;      uint32_t *cursors = nullptr, *range_lo = nullptr, *range_hi = nullptr, *range_cap = nullptr;
        // (range_hi doubles as the bucket kernel's retry list: buckets whose 512-slot table overflowed, done again with 1024 slots)
        bool retry_pass = true;    // launch that second pass behind the bucket kernel (first builds, captured builds, clouds that needed it)
        uint32_t retry_seen = 0;   // buckets the last resolved build sent through it
        int two_level_failures = 0;   // builds whose regions overflowed although sized from the sample
        bool one_level_ok = true;  // cleared when a bucket of the one-level tile partition (small clouds) overflowed its fixed room
        bool two_level_ok = true;  // cleared when the regions a cloud needs are too large: exact path from then on
        double fill1_ratio = 0.0;   // fullest level-1 region / mean seen on this handle (0 = unknown)
        uint64_t hist_cap = 0;     uint32_t* hist = nullptr;
        uint32_t bucket_cap = 0;   uint32_t* totals = nullptr; uint32_t* bucket_base = nullptr;
        uint64_t stage_cap = 0;    StageRow* stage = nullptr;    // stage_cap: staging rows the order arrays below can hold ...
        uint64_t stage_rows_cap = 0;                             // ... and how many StageRow records `stage` holds (the table / TILE / statistics paths;
                                                                 //     the PARTITION strategies stage RawNode records in `raw` instead and never grow it)
        RawNode* raw = nullptr;  uint64_t raw_cap = 0;          // k_bucket_direct's staging rows (dense, a column's rows adjacent)
        uint32_t *ord_cf = nullptr, *ord_idx = nullptr, *inv = nullptr;
        uint32_t* row_of = nullptr;        // [stage_cap] row of every staged node (table path: the incremental finalisation emits in place)
        uint32_t* row_ncol = nullptr; uint64_t row_ncol_cap = 0;     // per result row: its column's node count on the column's first row, else 0
        // column order (gndt_partition.hpp ColumnOrder): per bitmap word, and per point index for ncol_at
        uint64_t words_init = 0;   // bitmap / word_weight words the table path's column order has initialised
        uint64_t word_cap = 0;     uint32_t *bitmap = nullptr, *word_weight = nullptr, *word_base = nullptr, *bsum_words = nullptr,
                                            *ncol_at = nullptr;
        PartCounters* d_pc = nullptr;
        // Round 6: what level 1 needs zeroed BEFORE it starts (the cursors, the partition counters) exists twice; the level-1 kernel of a
        // build zeroes the set the NEXT build takes (and this build's bitmap / Counters, which nothing reads before the bucket kernel),
        // so that an eager build is not preceded by a k_part_clear launch (~4.5 us of a 49 us frame).  Not for handles that ever
        // recorded a hipGraph: a replay uses the set it was recorded with, unseen by the host.
        uint32_t* cursors_alt = nullptr;
        PartCounters* d_pc_alt = nullptr;
        bool alt_clean = false;         // cursors_alt / d_pc_alt are zero (a level-1 kernel launched on this stream left them so)
        PartCounters* h_pc = nullptr;   // pinned
        unsigned long long* dbg = nullptr;  uint32_t dbg_buckets = 0;   // diagnostic phase stamps (gndt_debug_enable_stamps)
        uint32_t last_buckets = 0;
        uint64_t nodes_learned = 0;   // node count of the last successful PARTITION build (+20 %)
        // blocked buckets (gndt_blocked.hpp): 0 = not looked at yet, 1 = the map of the last build is a dense box: the next builds of
        // clouds of that size take them, -1 = no (or a blocked build failed)
        int blk_state = 0;  uint64_t blk_n = 0;  BlockMap blk_map{};  uint32_t blk_buckets = 0;
        uint32_t* d_extent = nullptr;  uint32_t* h_extent = nullptr;      // k_key_extent's seven words and their pinned mirror
        int good_slots = 0; uint64_t good_est = 0, good_n = 0;   // table size / estimate that worked last time
        int good_load = 0;          //   ... and the table load (percent) if it had to be lowered (0: the default)
        int load_pct = 60;          // average LDS-table load (percent) the bucket count aims at
        // how the last PARTITION attempt of this handle was sized: a build RECORDED into a hipGraph right after it is sized the same
        // way (what the eager build allocated is then enough — what it has learnt since, e.g. "more buckets next time", would ask
        // for buffers a capture cannot allocate)
        uint64_t last_n = 0, last_est = 0, last_stage_want = 0, last_rows_floor = 0;  int last_attempt = 0, last_load = 0;
        double pair_ratio = -1.0;   // share of the last resolved build's records that sat next to one of their own node inside a bucket (< 0: unknown)
        uint64_t retries_total = 0; // builds re-run because a table / region / staging area was too small (gndt_debug_retry_count)
    } part;
    // cost-map flood over the finished grid (gndt_cost.hpp)
    struct Cost {
        uint64_t node_cap = 0;     uint32_t *h_bits = nullptr, *state = nullptr, *f[2] = {nullptr, nullptr};
        uint32_t ctab_size = 0;    uint64_t* ctab_key = nullptr; uint32_t* ctab_val = nullptr;
        uint32_t* nbr = nullptr;       // per-flood tables, node_cap rows each: neighbour columns (8 words), own column + collision verdict (2), ring step masks (4), ring extremes (4)
        CostEdge* edges = nullptr;     // node_cap x 4 records: what a slope does to each of its neighbour cells (gndt_cost.hpp)
        CostCounters* d_cc = nullptr;
        CostCounters* h_cc = nullptr;   // pinned
        uint64_t serial = 0;            // result_serial the flood was computed for (0 = none)
        uint64_t tables_serial = 0;     // result_serial the per-map tables (column index, nbr / self / edges, ring verdicts) belong to (0 = none) ...
        float tables_robot[4] = {0, 0, 0, 0};   // ... and the robot they were worked out for: the next goal on the same map reuses them
        int ring_n = 0, ring_store = 0;
    } cost;
    // statistics exchange of a sharded build (gndt_exchange.hpp, gndt_api_dist.hip)
    struct Exchange {
        unsigned long long* d_counts = nullptr; uint64_t counts_cap = 0; unsigned long long* h_counts = nullptr;
        uint64_t *keys_in = nullptr, *keys_all = nullptr, *keys_sorted = nullptr, *canon = nullptr;
        uint64_t keys_in_cap = 0, keys_all_cap = 0, keys_sorted_cap = 0, canon_cap = 0;
        unsigned int* d_unique = nullptr;  uint32_t* d_missing = nullptr;
        char* scratch = nullptr; uint64_t scratch_cap = 0;
        double* packed = nullptr; uint64_t packed_cap = 0;  uint32_t* pfirst = nullptr; uint64_t pfirst_cap = 0;
        double* r_sums = nullptr; uint64_t r_sums_cap = 0;  uint32_t* r_count = nullptr; uint64_t r_count_cap = 0;
        double* red_tmp = nullptr; uint64_t red_tmp_cap = 0;      // thread ranks: where an in-place all-reduce is summed up
        // owner-partitioned build (gndt_build_owned_device): records grouped by owner, the records this rank owns, its
        // columns as (first-seen index, node count) pairs, everybody's pairs, the global row of every local row
        float4* send_recs = nullptr; uint64_t send_cap = 0;  float4* own_recs = nullptr; uint64_t own_cap = 0;
        uint32_t* d_matrix = nullptr; uint64_t matrix_cap = 0;  uint32_t* h_matrix = nullptr;     // [W x W] send counts
        gndt::Counters* d_split_cnt = nullptr;  gndt::Counters* h_split_cnt = nullptr;
        unsigned long long* pairs = nullptr; uint64_t pairs_cap = 0;  unsigned long long* pairs_all = nullptr; uint64_t pairs_all_cap = 0;
        uint32_t* d_npairs = nullptr;  uint32_t* global_row = nullptr; uint64_t global_row_cap = 0;
        unsigned long long* gw = nullptr; uint64_t gw_cap = 0;                                   // bitmap word + weight, packed (k_pairs_note)
        unsigned long long* d_colmsg = nullptr;  unsigned long long* h_colmsg = nullptr;         // [2 x ranks] column count, build failed
        unsigned long long* d_totals = nullptr;  unsigned long long* h_totals = nullptr;          // [4] nodes, columns, slopes, points
        hipEvent_t ev[5] = {};          // stage stamps of the sharded builds (created once, reused)
        uint32_t* h_bad = nullptr;      // pinned: the "pair beyond the index range" counter comes back here
        uint64_t send_off[1025] = {}, send_cnt[1025] = {};   // host: start and length of every owner's run in send_recs (after the split)
        bool split_one_pass = false;  uint64_t split_cap = 0;   // one-pass split: run r at r * split_cap
        // the assembled map (gndt_gather_owned_map_device): this rank's rows packed for travel, everybody's rows, the adopt tally
        uint32_t* grec = nullptr; uint64_t grec_cap = 0;  uint32_t* grec_all = nullptr; uint64_t grec_all_cap = 0;
        uint32_t* d_tally = nullptr;  uint32_t* h_tally = nullptr;
        uint64_t owned_serial = 0;  uint32_t owned_world = 0;   // result_serial / ranks of the owned build global_row describes (0: none)
        bool gathered = false;                                  // gndt_gather_owned_map_device has run for that build (every rank: once per build)
        uint32_t* d_status = nullptr;                           // scratch word for the status kernels
        // agreement rounds (gndt_api_dist.hip `agree`): one word per rank, device and pinned; allocated with the handle, so that a
        // rank can ALWAYS say "I cannot go on" — whatever else it failed to allocate
        unsigned long long* d_agree = nullptr;  unsigned long long* h_agree = nullptr;
        int inject_site = 0;                                    // tests: the allocation site that fails next (gndt_debug_fail_next_alloc)
        // sliced global rows: the first row of every pair of this rank, everybody's pair places, this rank's, slice totals / rows
        uint32_t* row_of_pair = nullptr; uint64_t row_of_pair_cap = 0;  uint32_t* place_all = nullptr; uint64_t place_all_cap = 0;
        uint32_t* place_mine = nullptr; uint64_t place_mine_cap = 0;  unsigned long long* d_slice = nullptr;   // [2 mine | 2W all | W rows]
        // locality-aware ownership (gndt_exchange.hpp): this rank's sample message, everybody's, the block table
        uint32_t* owner_msg = nullptr;  uint32_t* owner_msgs_all = nullptr; uint64_t owner_msgs_cap = 0;
        uint32_t* bkey = nullptr;  uint32_t* bcnt = nullptr; uint64_t bcnt_cap = 0;  uint8_t* bown = nullptr;  uint32_t* d_owner_full = nullptr;
        uint32_t owner_map_world = 0;   // ranks the block table in bkey / bown was made for (0: none — hash ownership)
    } exch;
    uint64_t result_serial = 0;         // bumped whenever a build / finalize produces new result rows
    int last_strategy = GNDT_STRATEGY_ATOMIC;
    // strategy AUTO: what the locality sample said last time, for which cloud size, and how many builds ago
    int tile_choice = -1;  uint64_t tile_choice_n = 0;  int tile_choice_age = 0;  double tile_ratio_seen = 0.0;
    unsigned long long* d_sample = nullptr;  unsigned long long* h_sample = nullptr;   // k_tile_sample's counters {points, nodes, ticket} and the pinned words its last workgroup fills
    // a locality sample whose answer has not been looked at yet (round 6: a build on a handle that already has room for any answer does
    // not wait for it — the build it rides in front of takes the handle's last choice, the next one finds the answer)
    bool sample_pending = false;  uint64_t sample_n = 0;  hipEvent_t sample_ev = nullptr;
    uint32_t* d_sketch = nullptr;  uint32_t* h_sketch = nullptr;                        // k_node_sketch's HyperLogLog registers (pinned copy)
    bool map_in_table = true;   // false after a PARTITION build: the HBM node table does not hold the map

    // optional phase timing (bench / profiling): events recorded on the launch stream
    int prof = 0;               // 0 off, 1 every phase, 2 only the dominant phase of the strategy in use
    // one event set per build in a ring, so that back-to-back (un-synchronised) builds can all be timed
    static constexpr int kEvSets = 32;
    hipEvent_t ev[kEvSets][GNDT_NUM_PHASES + 1] = {};
    bool ev_recorded[kEvSets][GNDT_NUM_PHASES + 1] = {};
    int ev_set = 0;

    // A PARTITION build is launched without waiting for it; its overflow flags are looked at (and the build
    // re-run with more room if they are set) by the next call that needs the result.
    struct Pending {
        bool active = false;
        const void* xyz = nullptr; size_t n = 0, stride = 0;
        hipStream_t s = nullptr;
        int attempt = 0, bslots = 0;
        int load_pct = 0;               // average LDS-table load this build aims at (0: the handle's default)
        uint64_t nodes_est = 0, stage_want = 0, est0 = 0;   // est0: the estimate the first attempt used
        uint64_t rows_floor = 0;        // staging rows THIS build asks for whatever the handle already has (stage_want also covers what it has)
        bool est_reliable = false;      //   ... and whether it came from a hint / an earlier build rather than the n / 4 guess
        bool two_level = false;         // this attempt used the two-level partition
        bool one_level = false;         //   ... the one-level tile partition (small clouds)
        bool retry_pass = false;        //   ... and the bucket kernel's second pass behind the first (overflowing 512-slot tables done again with 1024)
        double mean1 = 0.0;             // its mean level-1 region fill (to turn the fullest region into a ratio)
        bool stats_only = false;        // gndt_shard_stats_device: statistics out, no labels / ordering / rows
        bool blocked = false;           // this attempt used blocked buckets (gndt_blocked.hpp)
        bool no_block = false;          //   ... a blocked attempt of this build failed: hashed buckets from here on
        uint64_t captured_gen = 0;      // Handle::realloc_gen when this build was recorded (captured builds only)
        bool captured = false;          // launched on a stream under hipGraph capture: it runs when the graph is replayed, with the
                                        //   buffers it was recorded with — never re-run here with more room (new buffers: the graph holds the old)
        bool replay_failed = false;     // the last replay of this build ran out of room (results_valid is false because of that, and only that)
        uint64_t done_serial = 0;       // result_serial when this build was resolved: while the handle still shows that serial, the map
                                        //   on the device can only have been replaced by a REPLAY of this build (hipGraph)
        gndt::GridParams gp{};          // origin and grid parameters AS THEY WERE when the build was launched: a retry
                                        // re-runs the same build even if the handle's origin has moved on since
        uint32_t first_base = 0;        // global index of xyz[0] (shards of a global cloud)
        bool records = false;           // the input is 16-B records {x, y, z, index word} (owner-partitioned build): the index
        uint64_t index_range = 0;       //   word is taken as it is; point indices then run over [0, index_range) (0: n)
        const void* xyz2 = nullptr; size_t n2 = 0;   // records only: a second segment, PRECEDED in its allocation by room for the
                                        //   first one (the exact partition reads one array: the first segment is copied in front)
    } pending;

    std::string err;
};

namespace gndt_host {
using namespace gndt;

// Diagnostic / tuning knobs from the environment, parsed ONCE per process (DESIGN.md "Diagnostic and tuning knobs").
// (Folded into constants in round 4, each after losing its A/B: level-1 cursor replicas (1), workgroups of the counting partition
//  (256), incremental updates through the tile kernel (by strategy), every rank ordering all columns instead of its slice (no).)
constexpr int kPartWgs = 256;    // workgroups of the exact counting partition
struct Tuning {
    // constants (each with the measurement that set it) ...
    int bucket_load = 60;        // average LDS-table load (percent) that sizes the bucket count
    int bucket_load_large = 75;  // the same from 2 M points on (the chip is full either way: fuller tables, fewer buckets; 85 / 92 with the second pass behind them: noise, r05 §4)
#ifndef GNDT_BUCKET_POINTS
#define GNDT_BUCKET_POINTS 0
#endif
    int bucket_points = GNDT_BUCKET_POINTS;       // points per bucket (0 = derived: 700 .. 3600 with the cloud's size)
    int bucket_slots = 0;        // LDS table of the first attempt (0 = 512, 1024 on a retry)
    int two_level = -1;          // 0 = never use the two-level partition
    uint32_t l1_wgs = 2048;      // level-1 workgroups (512 are resident: 2048 of them, ~1.2 tiles each, 69 us against 75 with 1024 and 72 with 512 on the bench scene, r05 §5)
    uint32_t bucket_wgs = 0xFFFFFFFFu;   // persistent bucket workgroups (default: one per bucket; 512 persistent ones measured +8 %)
    int one_level = 1;           // small clouds: level 1 writes the buckets directly (0: counting partition)
    int owner_locality = 1;      // owner-partitioned build: sampled block ownership (1) or hash ownership only (0)
    int interleave = -1;         // bucket kernel: which records a lane takes — consecutive pairs (0), pairs interleaved over the waves (1), a contiguous
                                 //   stretch of the bucket per lane (2) — or by the last build's locality (-1: 0 without, 2 with; r05 §3)
    int sketch = 1;              // a fresh handle without a hint counts its first cloud's nodes (HyperLogLog pass) instead of guessing n / 4
    int retry_pass = 1;          // 0: never launch the bucket kernel's second pass (an overflowing 512-slot table re-runs the build, as before round 5)
#ifndef GNDT_PLACE_EMIT_WORDS
#define GNDT_PLACE_EMIT_WORDS 16384
#endif
    int place_emit_words = GNDT_PLACE_EMIT_WORDS;    // PARTITION builds of clouds with up to this many bitmap words (32 points each): ordering + emit as one kernel
                                                     //   in scatter form (gndt_partition.hpp k_place_emit_rows; 0: never)
    int dest_scans = 1;          // small clouds: the destination pass scans the word weights itself instead of one or two scan launches in front of it
                                 //   (200 k-point campus frame 0.0549 -> 0.0535 ms, bridge_ground 0.0647 -> 0.0646: r05 ablation 6h)
    int small_tiles = 1;         // one-level partition of < 1 M points: 1024-point level-1 tiles (a few hundred workgroups instead of a few dozen:
                                 //   campus frame 0.0535 -> 0.0520 ms, bridge_ground 0.0648 -> 0.0608; r05 ablation 6i)
#ifndef GNDT_BLOCKED
#define GNDT_BLOCKED 1
#endif
    int blocked = GNDT_BLOCKED;  // clouds whose map is a dense, evenly filled box: spatial blocks as buckets, directly addressed tables (round 6)
#ifndef GNDT_THREE_WGS
#define GNDT_THREE_WGS 1
#endif
    int bucket_three_wgs = GNDT_THREE_WGS;    // clouds without locality: the bucket kernel with one record per thread at three workgroups per CU (round 6)
#ifndef GNDT_FOLD_CLEAR
#define GNDT_FOLD_CLEAR 1
#endif
    int fold_clear = GNDT_FOLD_CLEAR;    // eager PARTITION builds: no k_part_clear launch — level 1 prepares the next build's cursors / counters (Part::cursors_alt)
    int fp_bits = 21;            // bits of the bucket kernel's index fingerprint (tests narrow it through gndt_debug_set_fp_bits to force clashes)
    // ... and what gndt_debug_set_option / gndt_debug_enable_stamps can set (process-wide; the library reads no environment variable)
    double tile_ratio = 48.0;    // GNDT_DEBUG_TILE_RATIO   AUTO takes strategy TILE from this many points per partial on (sampled; the
                                 //                         measured crossover, profiles/r02_tile_calibration.json; tools/calibrate_tile.py sweeps it)
    bool stamps = false;         // gndt_debug_enable_stamps: in-kernel phase stamps of the bucket kernel
    bool verbose = false;        // GNDT_DEBUG_VERBOSE      stderr line per resolved two-level build
    bool cost_one_workgroup = true;   // GNDT_DEBUG_COST_ONE_WORKGROUP   0: every layer of the flood its own launch
};
const Tuning& tuning();
void tuning_force_stamps(bool on);   // bench.py --stamps flips this after the timed run
void tuning_force_fp_bits(int bits); // tests: narrow the fingerprint so that clashes happen
int tuning_set_option(int option, double value);

#define HIP_TRY(h, expr)                                                                                 \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                               \
            return GNDT_ERR_HIP;                                                                         \
        }                                                                                                \
    } while (0)

// Allocating, freeing or waiting while a stream of the process is being captured makes HIP refuse the call AND invalidate the
// capture (hipErrorStreamCaptureInvalidated at hipStreamEndCapture; round 3's fuzz: one such capture and every later capture of
// the process was refused).  Code that is about to do one of these asks first and reports GNDT_ERR_CAPACITY — a clean error the
// caller answers with gndt_reserve() (or an eager build of the same size) before capturing.
#define GNDT_NO_CAPTURE(h, what)                                                                                              \
    do {                                                                                                                      \
        {   /* asked of the stream itself: entry points that enqueue nothing (gndt_sync, gndt_export*) do not pass use_stream, and a flag left by the last captured call would refuse them */ \
            hipStreamCaptureStatus c__ = hipStreamCaptureStatusNone;                                                          \
            (void)hipStreamIsCapturing((h)->last_stream, &c__);                                                               \
            (h)->capturing = c__ != hipStreamCaptureStatusNone;                                                               \
        }                                                                                                                     \
        if ((h)->capturing) {                                                                                                 \
            (h)->err = std::string(what) + " must be (re)allocated or waited for, which a stream under hipGraph capture cannot do: "  \
                       "call gndt_reserve(max_points, max_nodes) — or build a cloud of this size eagerly — before capturing";  \
            return GNDT_ERR_CAPACITY;                                                                                         \
        }                                                                                                                     \
    } while (0)

inline void mark(gndt_handle* h, int i, hipStream_t s) {
    if (!h->prof || !h->ev[h->ev_set][i]) return;
    if (h->prof == 2) {   // the bucket kernel sits between marks 4 and 5, k_accumulate between 1 and 2
        const int lo = h->last_strategy != GNDT_STRATEGY_ATOMIC ? 4 : 1;
        if (i != lo && i != lo + 1) return;
    }
    (void)hipEventRecord(h->ev[h->ev_set][i], s);
    h->ev_recorded[h->ev_set][i] = true;
}

// a new build / update starts: next event set of the ring
inline void next_event_set(gndt_handle* h) {
    if (!h->prof) return;
    h->ev_set = (h->ev_set + 1) % gndt_handle::kEvSets;
    for (auto& r : h->ev_recorded[h->ev_set]) r = false;
}

inline int grid_for(uint64_t work, int block = kBlock, int max_blocks = 256 * 8) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

inline uint32_t pow2_ceil(uint64_t v) {
    uint64_t p = 1024;
    while (p < v && p < (1ull << 31)) p <<= 1;
    return (uint32_t)p;
}

inline GridParams grid_params(const gndt_handle* h) {
    GridParams g{};            // (blk.on = 0: buckets by column hash; partition_launch fills the block map in when it takes the blocked build)
    g.ox = h->origin[0]; g.oy = h->origin[1]; g.oz = h->origin[2];
    g.grid_len = h->P.grid_len; g.z_len = h->P.z_len; g.slope_interval = h->P.slope_interval;
    g.demand = h->P.demand; g.min_points = h->P.min_points;
    g.inv_grid = 1.0f / g.grid_len; g.inv_z = 1.0f / g.z_len;     // correctly rounded on the host
    return g;
}

// slots wanted for `nodes` occupied entries (load factor <= 1/2)
inline uint32_t cap_for_nodes(uint64_t nodes) { return pow2_ceil(std::max<uint64_t>(2048, nodes * 2)); }

// NULL = the handle's own stream; hipStreamLegacy (the null stream by its explicit name) = the null stream handle itself,
// which every HIP call accepts (some do not accept the named constant)
inline hipStream_t stream_of(gndt_handle* h, void* hip_stream) {
    if (!hip_stream) return h->own_stream;
    return (hipStream_t)hip_stream == hipStreamLegacy ? (hipStream_t) nullptr : (hipStream_t)hip_stream;
}

// a device buffer of the handle goes out of use (see Handle::retired)
inline void release_device(gndt_handle* h, void* p, bool bump = true) {     // (bump = false: the node table, which has a generation of its own)
    if (!p) return;
    if (bump) ++h->realloc_gen;
    if (h->ever_captured) h->retired.push_back(p); else (void)hipFree(p);
}

template <typename T>
int grow_buf(gndt_handle* h, T*& p, uint64_t& cap, uint64_t want) {
    if (want <= cap) return GNDT_OK;
    GNDT_NO_CAPTURE(h, "a work buffer");
    // Once a graph has been recorded an outgrown buffer is kept until gndt_destroy (release_device): a handle whose eager clouds
    // grow slowly would keep the sum of all earlier sizes.  Growing by half bounds what is retired to twice the final size.
    if (h->ever_captured && cap) want = std::max<uint64_t>(want, cap + cap / 2);
    release_device(h, p);
    p = nullptr; cap = 0;
    HIP_TRY(h, hipMalloc(&p, want * sizeof(T)));
    cap = want;
    return GNDT_OK;
}

// ---- gndt_api_core.hip: shared buffers ----
int create_handle(const gndt_params* params, gndt_handle** out, hipStream_t borrowed_stream);   // gndt_create's body
int check_ready(gndt_handle* h);
int use_stream(gndt_handle* h, hipStream_t s);     // work moves to stream s: it waits for what the handle's last stream still runs
int ensure_out(gndt_handle* h, uint64_t n);
int ensure_stats_buffers(gndt_handle* h, uint64_t n);
int ensure_stage(gndt_handle* h, uint64_t nodes, bool rows = true);   // rows = false: the order arrays only (PARTITION: RawNode records, ensure_raw)
int ensure_raw(gndt_handle* h, uint64_t records);
int ensure_words(gndt_handle* h, uint64_t words);
int ensure_part_counters(gndt_handle* h);
int ensure_cursors(gndt_handle* h, uint64_t buckets);   // gndt_api_build.hip
int fetch_counters(gndt_handle* h, hipStream_t s);   // read the device counters (synchronises the stream)
int stage_host_input(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes, hipStream_t s);
void free_part(gndt_handle* h);
// ---- gndt_api_table.hip ----
void free_table(gndt_handle* h);
int alloc_table(gndt_handle* h, uint32_t cap, hipStream_t s);
int do_reset(gndt_handle* h, hipStream_t s);
int zero_device_now(gndt_handle* h, void* p, size_t bytes);
int partition_recheck_after_replay(gndt_handle* h);
int reserve_table(gndt_handle* h, uint64_t nodes, hipStream_t s);
int table_refinalize(gndt_handle* h);      // the regular finalisation after the small-map one gave up (gndt_sync)
int table_emit_pending(gndt_handle* h);    // deferred-emit mode: the ordering + emit pass the frames since the last read left out (gndt_sync)
int build_atomic(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, bool tile = false,
                 const gndt_handle::Pending* rec = nullptr);
int locality_sample(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, double* ratio, hipStream_t s);
int locality_sample_begin(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, hipStream_t s);   // launched, not awaited
int ensure_sample_buffers(gndt_handle* h);
bool locality_sample_take(gndt_handle* h, bool wait, double* ratio);                                                          // its answer, if there is one
int sketch_nodes(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, uint64_t* estimate);   // HyperLogLog over the node keys (waits)
// ---- gndt_api_build.hip ----
// tab_end: the table path's end-of-frame bookkeeping (k_tab_end + `advance` points of stream position) done by k_emit_rows
// partial: incremental finalisation — only rows from the first changed column on are placed and gathered again, touched rows in
//          front of it are emitted where they are (k_order_dest / k_emit_rows, gndt_partition.hpp)
int launch_order_and_emit(gndt_handle* h, uint64_t words, int m0, hipStream_t s, bool grouped = false, bool counters_to_host = false,
                          bool tab_end = false, uint32_t advance = 0, bool partial = false);
int partition_launch(gndt_handle* h, gndt_handle::Pending& P);
int partition_begin(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s, bool records = false, uint64_t index_range = 0,
                    const void* records2 = nullptr, size_t n2 = 0);
int partition_resolve(gndt_handle* h);
// ---- gndt_api_cost.hip ----
void free_cost(gndt_handle* h);

}  // namespace gndt_host
