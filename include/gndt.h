/* gndt.h — C ABI of the MI355X-native NDT grid builder (libgndt.so).
 *
 * This is the drop-in boundary for ONE path of daysun/grid_ndt: the two statements of
 * chatterCallback that build the map,
 *     for (i = 1 .. n-1) uniformDivision(points[i], false);     src/receiver.cpp:150-154
 *     map2D.create2DMap(demand);                                 src/receiver.cpp:160
 * i.e. point -> (xy Morton key, z level) binning (include/map2D.h:950-976, src/receiver.cpp:41-93),
 * the per-node mean / un-normalised scatter (map2D.h:611-627), the min-eigenpair that gives roughness
 * and normal (map2D.h:110-133) and the order-dependent slope label (map2D.h:66-108, 630-643).
 *
 * The reference has no FFI of its own (the path is a free function plus methods of a header-only
 * class working on a global `daysun::TwoDmap map2D`, receiver.cpp:35), so the entry points below are
 * what a binding for this path would need; each names the reference statement it replaces.
 * Plain pointers and sizes only; no C++ or torch types cross this boundary; nothing throws.
 *
 * Conventions kept from the reference:
 *   - point 0 of a cloud is the ORIGIN and is not binned (receiver.cpp:145, 150): call
 *     gndt_set_origin(h, &cloud[0]) and pass cloud+1, n-1 to gndt_build*.
 *   - inputs hold no NaN/Inf (the publisher strips them, src/publisher.cpp:24-26).
 *   - errors are status codes (the reference returns bool + prints, map2D.h:602-604).
 *   - one caller per handle at a time (ros::spin() is single-threaded, receiver.cpp:283).
 */
#ifndef GNDT_H
#define GNDT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gndt_handle gndt_handle;

enum {
    GNDT_OK = 0,
    GNDT_ERR_INVALID = 1,      /* bad argument / call order */
    GNDT_ERR_NO_DEVICE = 2,    /* no usable HIP device: the product path never falls back to the CPU */
    GNDT_ERR_HIP = 3,          /* a HIP runtime call failed; see gndt_last_error */
    GNDT_ERR_KEY_RANGE = 4,    /* |nx| or |ny| > 65535 (countMorton wraps, Stopwatch.h:102-110) or |nz| >= 2^21 */
    GNDT_ERR_CAPACITY = 5,     /* node table full and growth disabled */
    GNDT_ERR_NOMEM = 6,
    GNDT_ERR_PEER = 7          /* sharded builds: another rank of the communicator reported a failure; every rank left the
                                  collective sequence at the same point (this rank's own input was fine) */
};

enum { GNDT_DEMAND_SLOPE = 0, GNDT_DEMAND_TRUE = 1 };   /* create2DMap(demand), map2D.h:630, 644 */

/* gndt_cells.flags */
enum {
    GNDT_FLAG_HAS_STATS = 1u,  /* node reached min_points: mean/cov/N are set (map2D.h:611) */
    GNDT_FLAG_SLOPE = 2u,      /* a Slope object exists for the node (map2D.h:632 / :648) */
    GNDT_FLAG_DOWN = 4u        /* Slope::down (map2D.h:636) */
};

/* accumulate strategy (how points reach the per-node sufficient statistics) */
enum {
    GNDT_STRATEGY_AUTO = 0,
    GNDT_STRATEGY_ATOMIC = 1,     /* one pass, wave-aggregated fp64 atomics into the HBM node table */
    GNDT_STRATEGY_PARTITION = 2,  /* partition by column hash, then LDS-resident accumulation; large builds use the
                                     two-level partition without counting passes and fall back to the exact one */
    GNDT_STRATEGY_PARTITION_EXACT = 3,      /* always the single-level counting partition (histogram + offsets + scatter) */
    GNDT_STRATEGY_PARTITION_TWO_LEVEL = 4,  /* the two-level partition whatever the size (PARTITION picks it from 2^20 points) */
    GNDT_STRATEGY_PARTITION_ONE_LEVEL = 6,  /* (reported by gndt_last_strategy only) small clouds: one tile-sort level writes the
                                             * buckets directly; PARTITION / AUTO pick it when the cloud needs at most 512 buckets */
    GNDT_STRATEGY_PARTITION_BLOCKED = 7,    /* (reported by gndt_last_strategy only) the two partition levels with SPATIAL blocks of 512 nodes as
                                             * buckets and a bucket kernel that addresses its table directly (no index, no search): taken by AUTO /
                                             * PARTITION for clouds whose map is a dense, evenly filled box of bounded height (what the previous
                                             * build on the handle found), abandoned — the build re-run with hashed buckets — when a cloud outgrows it */
    GNDT_STRATEGY_TILE = 5        /* one pass for clouds that keep their scan order: contiguous ranges of the cloud, node table
                                     privatised in LDS per workgroup, ONE partial per distinct node and flush into the HBM node
                                     table (gndt_tile.hpp).  AUTO takes it when a sample of the cloud shows enough points per
                                     partial (gndt_locality_sample); like ATOMIC it keeps additive state (updates, statistics) */
};

typedef struct {
    float grid_len;         /* TwoDmap::gridLen  (setLen, map2D.h:493)   */
    float z_len;            /* TwoDmap::zLen     (setZLen, map2D.h:496)  */
    float slope_interval;   /* setInterval, map2D.h:499 */
    int32_t demand;         /* GNDT_DEMAND_* */
    int32_t min_points;     /* MINPOINTSIZE = 3, map2D.h:28 */
    int32_t device_id;      /* HIP device ordinal */
    int32_t strategy;       /* GNDT_STRATEGY_* */
    uint64_t max_points_hint;  /* largest batch expected (0 = grow on demand) */
    uint64_t max_nodes_hint;   /* occupied (xy,z) nodes expected (0 = derive from the batch size) */
} gndt_params;

/* Structure-of-arrays view of the finished map: one entry per occupied (xy,z) node — every OcNode of
 * map_xy, including those with fewer than min_points points (they keep zero mean/cov, map2D.h:54-56).
 * Order: columns in first-seen order (= morton_list, receiver.cpp:70), nodes of a column in
 * first-seen order (= insertion order among equal keys of the multimap, receiver.cpp:88).
 * 76 bytes per node.  Pointers are device or host memory depending on the call that filled them. */
typedef struct {
    uint64_t num_nodes;
    uint64_t num_columns;    /* = morton_list.size() */
    uint64_t num_slopes;
    int32_t* sx;             /* signed x index: +nx in quadrants A,B (px > ox), -nx in C,D */
    int32_t* sy;             /* signed y index: +ny in quadrants A,C (py > oy), -ny in B,D */
    int32_t* sz;             /* morton_z: signed z level, never 0 (map2D.h:963-973) */
    uint32_t* count;         /* points binned into the node */
    uint32_t* first_idx;     /* index (in the accumulated stream) of the node's first point */
    float* mean;             /* [num_nodes][3] OcNode::xyz_centroid / Slope::mean */
    float* cov;              /* [num_nodes][6] upper triangle xx,xy,xz,yy,yz,zz of covariance_matrix (un-normalised) */
    float* rough;            /* Slope::rough: min eigenvalue (0 -> 0.01, map2D.h:131) */
    float* normal;           /* [num_nodes][3] Slope::normal: its eigenvector, unit length, sign free */
    uint32_t* flags;         /* GNDT_FLAG_* */
} gndt_cells;

/* Per-node sufficient statistics (cell-local coordinates), the additive state exchanged between GPUs.
 * v = p - centre(node); centre is a pure function of (key, origin, grid_len, z_len). */
typedef struct {
    uint64_t num_nodes;
    uint64_t* key;           /* packed (sx,sy,sz), see gndt_pack_key */
    double* sums;            /* [num_nodes][9]: Sum v (3), Sum v v^T upper triangle (6) */
    uint32_t* count;
    uint32_t* first_idx;
} gndt_stats;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* `TwoDmap map2D(res, zres)` + setInterval (receiver.cpp:35, 267-269) */
int gndt_create(const gndt_params* params, gndt_handle** out);
void gndt_destroy(gndt_handle* h);
const char* gndt_last_error(const gndt_handle* h);   /* h may be NULL: last error of a failed create */

/* `map2D.setCloudFirst(points[0])` (receiver.cpp:145, map2D.h:490) */
int gndt_set_origin(gndt_handle* h, const float origin_xyz[3]);
/* The origin in use (gndt_build_cloud takes it from the first valid point of the cloud). */
int gndt_get_origin(const gndt_handle* h, float origin_xyz[3]);

/* ---- build: replaces receiver.cpp:150-154 + :160 --------------------------------------------- */
/* Host memory in (e.g. pcl::PointCloud<PointXYZ>::points.data()+1, stride 16).  Synchronous. */
int gndt_build(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes);
/* Device memory in; all work is enqueued on `hip_stream` (a hipStream_t, may be NULL = the handle's own
 * NON-BLOCKING stream: work the caller has on the null stream is not ordered with it — a caller that fills its input on the
 * null stream passes hipStreamLegacy, the null stream's explicit name, instead).  stride_bytes is 12 (packed) or 16 (PointXYZ).
 * Partition strategies: the call returns once everything is enqueued.  The device-side overflow flags are read by
 * the next call that needs the result (gndt_sync, gndt_export*, gndt_compute_cost, ...), which waits for the
 * stream and, if a table or region was too small, re-runs the build with more room — so `xyz_dev` must stay valid
 * and unchanged until then (a later gndt_build* / gndt_reset on the handle abandons the pending build instead).
 * A PARTITION build captured in a hipGraph and replayed is not watched by the host: if the replayed cloud needs more room than
 * the captured one (LDS tables, partition regions, staging rows), gndt_sync reports GNDT_ERR_CAPACITY — build that cloud
 * eagerly, then capture again.
 * Strategies ATOMIC and TILE wait for the stream once before returning (and grow the table themselves) — except on a stream
 * under hipGraph capture, where they are recorded once for the table as it stands and a replay that outgrows it reports
 * GNDT_ERR_CAPACITY at gndt_sync.  AUTO picks per cloud (ATOMIC below 65 536 points; above, TILE when a sample of the
 * cloud shows dense scan-ordered cells, PARTITION otherwise); under capture it keeps the choice of the last eager build. */
int gndt_build_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream);
/* Allocate NOW every buffer a build / update of up to max_points points and max_nodes nodes (0: max_points / 4) can ask for,
 * whatever the strategy and whatever a re-run with more room would want.  After it such a build allocates nothing: it can be
 * captured into a hipGraph on a FRESH handle (no eager warm-up), and no replay finds its buffers moved.  Without it a captured
 * call that has to grow a buffer fails with GNDT_ERR_CAPACITY before touching the allocator (allocating on a capturing stream
 * would invalidate the capture — and, on this runtime, every later capture of the process).  Waits for the handle's stream. */
int gndt_reserve(gndt_handle* h, uint64_t max_points, uint64_t max_nodes);
/* The reference builds ONE map per process (receiver.cpp:137-160): what its user sees is the FIRST build of a process, not the
 * steady state.  HIP resolves a kernel's code at its first launch and every buffer is allocated on first use; on a 200 k-point frame
 * that made the first build 0.4-1.0 ms against 0.05 ms for the next one.  gndt_warmup takes both out of the first build:
 *   1. it runs a synthetic cloud of `expected_points` points (0: params.max_points_hint, else 200 000; at most 4 M) through every
 *      strategy family a build on this handle can take — AUTO's locality sample, the partition pipeline at that size, its exact
 *      fallback, ATOMIC, TILE, an incremental update, the export and a cost flood — on a TEMPORARY handle with this handle's
 *      parameters, device and stream, so that the kernels' code is loaded (process-wide, per device) and this handle's state
 *      stays that of a fresh handle;
 *   2. if params.max_points_hint (or expected_points) is set, it calls gndt_reserve(h, that, params.max_nodes_hint).
 * Waits for the device.  Not under stream capture.  Costs 5-40 ms, once; calling it again is cheap (the code is loaded). */
int gndt_warmup(gndt_handle* h, uint64_t expected_points);

/* Incremental add (the intent of changeCallback/change2DMap, receiver.cpp:179-212, map2D.h:672-822;
 * semantics defined in SURVEY.md Appendix A.7): after update(F1) .. update(Fk) the map equals
 * build(F1 || .. || Fk).  first_idx continues counting across calls.  Needs a handle of strategy ATOMIC (the node
 * table keeps the additive state).  gndt_update_device never waits for the host: every size it needs lives on the
 * device, so once the buffers exist (one eager frame, or max_nodes_hint + max_points_hint) the call can be captured
 * in a hipGraph and replayed per frame; table / row / index overflow is reported by gndt_sync.
 * Point indices (first_idx) are 32-bit: a map holds at most 2^32 - 2 points between two gndt_reset calls (about an hour of
 * a 10 Hz, 131 072-point stream); the next update then fails with GNDT_ERR_INVALID instead of wrapping. */
int gndt_update(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes);
int gndt_update_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream);
/* Stream-friendly mode (off by default).  The dense rows in the reference's order move whenever a column in front of them grows, so
 * producing them costs O(map) per frame however little the frame touched.  With deferred emit on, gndt_update* relabels the touched
 * columns and stops — O(touched) — and the rows are ordered and emitted by the next call that READS the map (gndt_sync,
 * gndt_export*, gndt_compute_cost): a node that integrates frames at 10 Hz and plans once a second pays for the dense form once a
 * second.  Results are those of the default mode (tests: a 100-frame stream equals one build of its points in both). */
int gndt_set_deferred_emit(gndt_handle* h, int on);

/* Incremental delete: the intent of del2DMap (include/map2D.h:826-915; its caller delCallback is commented out at
 * src/receiver.cpp:214-248, and the shipped merge formula is inconsistent: SURVEY row a10), DEFINED here as the inverse of
 * gndt_update: the given points — which must have been added before, with the same coordinates — leave their nodes.
 *   - counts and the additive statistics are exactly those of the stream without the points (sums up to fp64 rounding);
 *   - a node whose last point leaves is deleted (map2D.h:861-873) and takes its slope with it;
 *   - a node that falls below min_points keeps its key and count but no statistics, like a node that never reached them;
 *   - every surviving node KEEPS its first-seen index, i.e. its place in the order — the reference's containers do not
 *     re-sort either (multimap::erase, morton_list untouched).  The result therefore equals a fresh build of the remaining
 *     points whenever no surviving node lost its first point; otherwise the order (and with it the order-dependent slope
 *     label, map2D.h:66-108) is that of the ORIGINAL stream.
 * Needs the additive state (strategy ATOMIC / TILE or a map built by gndt_update*).  A point that was never added is an
 * error (GNDT_ERR_INVALID) and leaves the handle to be reset.  Waits for the device (the node table is compacted when
 * nodes died); every row is re-finalised. */
int gndt_remove(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes);
int gndt_remove_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream);

/* Split form of build for a cloud sharded over several GPUs:
 *   accumulate (binning + sufficient statistics only)  ->  [exchange stats]  ->  finalize.
 * `first_idx_base` is the global index of xyz_dev[0] so that first-seen order is global. */
int gndt_reset(gndt_handle* h, void* hip_stream);
int gndt_accumulate_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes,
                           uint64_t first_idx_base, void* hip_stream);
int gndt_finalize_device(gndt_handle* h, void* hip_stream);

/* ---- results --------------------------------------------------------------------------------- */
/* Waits for the handle's pending work, reports counts and any deferred device-side error. */
int gndt_sync(gndt_handle* h, uint64_t* num_nodes, uint64_t* num_columns, uint64_t* num_slopes);
/* Device-resident SoA of the last build/finalize; valid until the next build/update/destroy. */
int gndt_export_device(gndt_handle* h, gndt_cells* out);
/* Copies into caller-allocated host arrays sized from gndt_sync's num_nodes (NULL arrays skipped). */
int gndt_export(gndt_handle* h, gndt_cells* out_host);
/* The same rows in host memory the HANDLE owns (pinned, one D2H pass, no copy into caller arrays): the pointers stored in *out
 * stay valid until the next gndt_export_host / gndt_destroy on the handle.  What gndt_compat::TwoDmap reads the map from
 * (the reference's containers are filled from it, or — lazy mode — the consumers are served from it directly). */
int gndt_export_host(gndt_handle* h, gndt_cells* out);

/* ---- statistics exchange (multi-GPU; additive over any partition of the points) --------------- */
/* Device-resident compact list of this handle's occupied nodes. */
int gndt_stats_export_device(gndt_handle* h, gndt_stats* out, void* hip_stream);
/* Adds `in` (device memory) into the handle's table: sums add, counts add, first_idx takes the min. */
int gndt_stats_merge_device(gndt_handle* h, const gndt_stats* in, void* hip_stream);

/* One global map from a cloud sharded over several GPUs, without the node table:
 *   gndt_shard_stats_device   this rank's shard -> compact statistics of its occupied nodes (the counting
 *                             partition + LDS bucket pipeline of a normal build, statistics out instead of rows);
 *                             `first_idx_base` = global index of xyz_dev[0].  Returns with `out` filled.
 *   [exchange: union of keys, sum of sums/counts, min of first_idx — grid_ndt_amd/dist.py over RCCL]
 *   gndt_finalize_stats_device  the merged statistics -> the map.  `in` must hold every node once, SORTED BY KEY
 *                             (any total order of the packed keys: a column's nodes are then adjacent);
 *                             `total_points` = number of points of the whole cloud (first_idx < total_points). */
int gndt_shard_stats_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint64_t first_idx_base,
                            gndt_stats* out, void* hip_stream);
int gndt_finalize_stats_device(gndt_handle* h, const gndt_stats* in, uint64_t total_points, void* hip_stream);

/* The whole sharded build behind the C ABI, RCCL called from C++ (resolved with dlopen: single-GPU users do not need it):
 * one process per GPU, every rank calls gndt_build_global_device with ITS contiguous range of the cloud; on return every
 * rank's handle holds the map of the whole cloud (origin = the global cloud's point 0 on every rank).
 *   shard -> statistics | all-gather of the occupied keys -> canonical order | ONE packed all-reduce (9 sums + count, fp64)
 *   + one min all-reduce (first-seen) | finalize from the reduced statistics.
 * The communicator: rank 0 calls gndt_comm_unique_id and hands the 128 bytes to the other ranks by its own means (a file,
 * MPI, torch.distributed ...); every rank then calls gndt_comm_create. */
#define GNDT_COMM_ID_BYTES 128
typedef struct gndt_comm gndt_comm;
typedef struct {
    float shard_ms, exchange_ms, finalize_ms;     /* device time of the three stages (HIP events on the stream) */
    uint32_t ranks;
    uint64_t local_nodes, global_nodes, bytes_reduced;
} gndt_exchange_times;
int gndt_comm_unique_id(char id_out[GNDT_COMM_ID_BYTES]);
int gndt_comm_create(const char id[GNDT_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device_id, gndt_comm** out);
/* `world` communicators whose ranks are THREADS of this process (one handle and one calling thread per rank, any devices
 * that can copy to each other — normally one): gndt_build_owned_device then hands the runs over with device copies and host
 * barriers instead of RCCL.  One GPU can play a whole node (the tests do), and a process that drives several handles from
 * several threads needs no RCCL.  Every rank must make every call; destroy each communicator. */
int gndt_comm_create_threads(int32_t world, int32_t device_id, gndt_comm** out /* [world] */);
void gndt_comm_destroy(gndt_comm* c);
const char* gndt_comm_last_error(void);
/* One tiny, VERIFIED round of every collective the sharded builds use — all-gather, the all-to-all of runs (ncclSend / ncclRecv in
 * one group), reduce-scatter, all-reduce (f64 sum, u32 min) — on communicator `c`, all ranks calling together.  ok_mask: bit 0
 * all-gather, 1 send/recv, 2 reduce-scatter, 3 all-reduce; times include the host copies that carry the test data.  Returns GNDT_OK
 * only if every primitive gave the right answer on this rank (GNDT_ERR_PEER and gndt_last_error(h) name the wrong one).  Run it
 * once before the first sharded build of a job: that build is otherwise the first test of the transport. */
typedef struct {
    float all_gather_ms, exchange_ms, reduce_scatter_ms, all_reduce_ms;
    uint32_t ok_mask, ranks;
} gndt_comm_selftest_report;
int gndt_comm_selftest(gndt_handle* h, gndt_comm* c, gndt_comm_selftest_report* out, void* hip_stream);
/* first_idx_base = index of shard[0] in the whole cloud's binned points; total_points = binned points of the whole cloud.
 * `times` may be NULL (it makes the call wait for the stream). */
int gndt_build_global_device(gndt_handle* h, gndt_comm* c, const void* shard_xyz_dev, size_t n, size_t stride_bytes,
                             uint64_t first_idx_base, uint64_t total_points, gndt_exchange_times* times, void* hip_stream);

/* ---- owner-partitioned build of a sharded cloud: the POINTS travel, statistics do not ----------------------------
 * No reference counterpart (the reference is single-process).  Every column of the grid has an owner rank (a hash of its
 * xy cell); each rank sends every point to the owner of its column (all-to-all, 16 B per point, consecutive identical points
 * folded into weighted records first), builds the columns it owns with the ordinary pipeline — statistics, labels and rows
 * are final — and learns, from an all-gather of 8 B per column, which row each of its rows has in the map of the WHOLE
 * cloud.  The map stays sharded by owner:
 *     gndt_export* on the handle   = this rank's columns, in the reference's order among themselves
 *     global_row[r]                = the row that local row r has in the single-process map of the whole cloud
 * so the union of all ranks' rows scattered by global_row IS that map (tests/test_gpu_owner.py assembles it and compares it
 * with the oracle).  Per rank a build moves (W-1)/W of its points out and as many in; gndt_build_global_device moves 84 B per
 * node of the global map through an all-reduce and leaves the whole map on every rank.
 *
 * gndt_build_owned_device does all of it over RCCL.  The four steps are also exported one by one for hosts that bring their
 * own transport (MPI, a ROS bridge) and for the single-GPU tests that play W ranks on one device:
 *   gndt_owner_split_device    shard -> records grouped by owner: rank r's run is counts_host[r] records of 16 bytes starting
 *                              offsets_host[r] records into *records_dev (the runs need not be adjacent; valid until the
 *                              next call on the handle); waits for the stream
 *   gndt_build_records_device  the records a rank owns (runs of all ranks, concatenated in any order) -> its map; like
 *                              gndt_build_device it is launched, not awaited.  total_points = binned points of the whole cloud
 *   gndt_owned_columns_device  (first-seen index << 32 | node count) per column of the local map; waits
 *   gndt_owned_global_rows_device  everybody's pairs (any order, entries of ~0 are skipped) -> global_row of every local row
 *                              (device pointer, valid until the next build), node / column count of the whole map; waits */
typedef struct gndt_owned_info {
    uint64_t owned_points;          /* records this rank received (weighted records count once) */
    uint64_t local_nodes, local_columns;
    uint64_t global_nodes, global_columns, global_slopes;
    uint64_t bytes_sent, bytes_received;   /* over the links (the run a rank keeps for itself is not counted) */
    float split_ms, exchange_ms, build_ms, order_ms;   /* device time of the stages (HIP events on the stream) */
    uint32_t ranks;
} gndt_owned_info;
/* owner rank of the columns (sx[i], sy[i]) among `world` ranks: the hash gndt_owner_split_device uses (host helper, no GPU) */
int gndt_owner_of_columns(const int32_t* sx, const int32_t* sy, size_t n, uint32_t world, uint32_t* owner_out);
/* Optional, before the split — locality-aware ownership (gndt_build_owned_device does it by itself): a contiguous range of a
 * scan-ordered cloud covers a patch of ground, so a block of 32 x 32 columns is given to the rank that already holds most
 * of its points and those points never cross a link.
 *   gndt_owner_sample_device  65 536 evenly spaced points of the shard -> a fixed-size message (device pointer, 32-bit words)
 *   gndt_owner_map_device     the messages of ALL ranks, concatenated in rank order -> the block table kept on the handle and
 *                             used by gndt_owner_split_device calls for the same `world` until replaced.  Every rank must
 *                             build it from the same messages (it then holds the same table); columns of blocks the
 *                             samples missed, or of a block holding more than 1/(4 world) of the cloud, go by the hash.
 * Without these two calls (or with world > 16) ownership is gndt_owner_of_columns' hash. */
int gndt_owner_sample_device(gndt_handle* h, const void* shard_xyz_dev, size_t n, size_t stride_bytes, const uint32_t** msg_dev,
                             uint64_t* msg_words, void* hip_stream);
int gndt_owner_map_device(gndt_handle* h, const uint32_t* all_msgs_dev, uint32_t world, void* hip_stream);
int gndt_owner_split_device(gndt_handle* h, const void* shard_xyz_dev, size_t n, size_t stride_bytes, uint64_t first_idx_base,
                            uint64_t total_points, uint32_t world, const void** records_dev, uint64_t* counts_host,
                            uint64_t* offsets_host, void* hip_stream);
int gndt_build_records_device(gndt_handle* h, const void* records_dev, size_t n_records, uint64_t total_points, void* hip_stream);
/* The same from TWO segments, so that the run a rank keeps for itself need not be copied next to what it receives: the
 * second segment must be preceded, in its own allocation, by room for the n_first records of the first (only small builds,
 * which partition one array, fill that room). */
int gndt_build_records2_device(gndt_handle* h, const void* first_dev, size_t n_first, void* second_dev, size_t n_second,
                               uint64_t total_points, void* hip_stream);
int gndt_owned_columns_device(gndt_handle* h, const uint64_t** pairs_dev, uint64_t* n_pairs, void* hip_stream);
int gndt_owned_global_rows_device(gndt_handle* h, const uint64_t* all_pairs_dev, uint64_t n_all, uint64_t total_points,
                                  const uint32_t** global_row_dev, uint64_t* global_nodes, uint64_t* global_columns, void* hip_stream);
int gndt_build_owned_device(gndt_handle* h, gndt_comm* c, const void* shard_xyz_dev, size_t n, size_t stride_bytes,
                            uint64_t first_idx_base, uint64_t total_points, const uint32_t** global_row_dev,
                            gndt_owned_info* info, void* hip_stream);
/* Errors inside the collective sequence of gndt_build_owned_device / gndt_build_global_device: a rank that fails on its own
 * (bad input, a point outside the key range, a build that does not fit) still takes part in the collectives that follow, its
 * error code travelling with the messages that are exchanged anyway; every rank that sees a non-zero code leaves at the same
 * point — the failing rank with its own error, the others with GNDT_ERR_PEER — so no rank is left waiting.  Not reportable
 * this way, hence fatal for the group like the loss of a rank: a failing collective, and memory exhaustion for the exchange
 * buffers themselves.
 *
 * ---- one map for the consumers (src/receiver.cpp:171-175 runs computeCost and the planner on ONE map)
 * After gndt_build_owned_device the map is sharded by column owner.  gndt_gather_owned_map_device moves the finished rows —
 * 84 bytes each: the 76-byte row of gndt_cells, the column index the cost map uses, and the row's place in the map of the
 * whole cloud — to rank `root` (ncclSend / ncclRecv) or, with root < 0, to every rank (one padded all-gather), and scatters
 * them by that place into the handle's result arrays: the handle then HOLDS THE WHOLE MAP, in the reference's order, and
 * gndt_sync / gndt_export* / gndt_compute_cost / gndt_compat::TwoDmap work on it as after a single-GPU build.  Ranks other
 * than `root` keep the columns they own.  All ranks of the communicator call it, after a successful gndt_build_owned_device,
 * ONCE per build (a second call fails with GNDT_ERR_INVALID on every rank, before any collective).
 * For hosts with their own transport the two device steps are exported:
 *   gndt_owned_pack_rows_device   this rank's rows as packed records (GNDT_PACKED_ROW_WORDS 32-bit words each; device pointer
 *                                 valid until the next call on the handle)
 *   gndt_adopt_rows_device        packed records of ANY number of ranks (padding records, last word 0xFFFFFFFF, are skipped)
 *                                 -> the handle's result rows; fails with GNDT_ERR_INVALID unless exactly total_nodes rows
 *                                 arrive. */
#define GNDT_PACKED_ROW_WORDS 21
int gndt_gather_owned_map_device(gndt_handle* h, gndt_comm* c, int32_t root, void* hip_stream);
int gndt_owned_pack_rows_device(gndt_handle* h, const uint32_t** rows_dev, uint64_t* n_rows, void* hip_stream);
int gndt_adopt_rows_device(gndt_handle* h, const uint32_t* rows_dev, uint64_t n_rows, uint64_t total_nodes, uint64_t total_columns,
                           uint64_t total_slopes, void* hip_stream);

/* ---- cost map over the finished grid -------------------------------------
 * gndt_compute_cost replaces TwoDmap::computeCost (include/map2D.h:1285-1397; called at receiver.cpp:171
 * right after create2DMap): the FIFO flood from the goal slope with CollisionCheck (:351-411; the 3D variants
 * :414-474 when the handle's demand is "true") and AccessibleNeighbors (:530-588).  It fills, per result row,
 *   h      Slope::h  (FLT_MAX where the row has no Slope, was never reached, or collided and was not relaxed again)
 *   state  0 = untouched, 1 = traversable (the reference's `traversability` list), 2 = closed (collision)
 * bit-identical to the reference's sequential flood on the same grid (DESIGN.md "Cost map").
 * The call returns when the flood is complete.  Column indices must not exceed 32767 (mortonToXY's range,
 * Stopwatch.h:171-189).  `robot` NULL = RobotSphere(0.25) with robot.h:38-46's thresholds. */
typedef struct gndt_robot {
    float radius;            /* RobotSphere::r            receiver.cpp:33  */
    float reachable_height;  /* getReachableHeight() 0.15  robot.h:38-39   */
    float max_rough;         /* getRough() 100             robot.h:40-43   */
    float max_angle_deg;     /* getAngle() 30              robot.h:44-46   */
} gndt_robot;

typedef struct gndt_cost_stats {
    int32_t goal_status;     /* 0 flood ran; 1 no cell at the goal (reference: nothing happens); 2 no slope at the
                                goal's level ("Goal position wrong", map2D.h:1304-1306) */
    uint32_t ring;           /* n = (ceil(2r/gridLen)-1)/2, the collision ring depth (map2D.h:1310) */
    uint32_t levels;         /* layers of the flood */
    uint32_t ring_store;     /* 0: no collision ring (robot no wider than a cell); 1: the rings' verdicts came from `ring` rounds of
                                neighbour propagation over the whole map before the flood (no ring is listed, no capacity) */
    uint64_t traversable;    /* traversability.size() (map2D.h:1382) */
    uint64_t closed;         /* slopes closed by a collision */
    uint64_t check_pushes;   /* checkList.size() (map2D.h:1383) */
} gndt_cost_stats;

int gndt_compute_cost(gndt_handle* h, const float goal_xyz[3], const gndt_robot* robot, void* hip_stream);
/* Device-resident h (fp32) and state (u32) per result row; valid until the next build/update/compute_cost. */
int gndt_cost_export_device(gndt_handle* h, const float** h_dev, const uint32_t** state_dev, gndt_cost_stats* stats);
/* Copies into caller-allocated host arrays of num_nodes elements (NULL arrays skipped). */
int gndt_cost_export(gndt_handle* h, float* h_out, uint32_t* state_out, gndt_cost_stats* stats);

/* ---- input side ---------------------------------------------------------
 * Where x, y, z sit inside one raw point record: sensor_msgs::PointCloud2 fields / point_step, the records of a
 * binary .pcd, or pcl::PointXYZ itself (step 16, offsets 0, 4, 8).  Offsets are multiples of 4. */
typedef struct gndt_point_layout {
    uint32_t point_step, offset_x, offset_y, offset_z;
} gndt_point_layout;

/* A .pcd file as pcl::io::loadPCDFile reads it (src/publisher.cpp:19): header + payload.  `DATA binary` payloads are
 * returned as they are in the file (binary_compressed ones after LZF decompression and re-interleaving) (`layout` says where x, y, z are), `DATA ascii` as packed xyz.  Host-only. */
typedef struct gndt_pcd {
    uint64_t num_points;
    gndt_point_layout layout;
    int32_t data_kind;          /* 0 ascii (converted to packed xyz), 1 binary, 2 binary_compressed (decompressed, records rebuilt) */
    int32_t reserved;
    void* data;                 /* num_points * layout.point_step bytes, released by gndt_pcd_free */
} gndt_pcd;
int gndt_pcd_read(const char* path, gndt_pcd* out, char err[256]);
void gndt_pcd_free(gndt_pcd* pcd);

/* Raw records on the device -> packed fp32 xyz on the device ([n][3], caller-allocated), rows with a non-finite
 * coordinate dropped, order kept: pcl::fromPCLPointCloud2 (receiver.cpp:140-143) + pcl::removeNaNFromPointCloud
 * (publisher.cpp:24-26).  *n_valid (host) = rows written; the call returns when it is known. */
int gndt_pack_points_device(gndt_handle* h, const void* raw_dev, size_t n, const gndt_point_layout* layout,
                            float* xyz_out_dev, uint64_t* n_valid, void* hip_stream);
/* chatterCallback in one call (receiver.cpp:137-160): raw host records -> device, NaN strip, origin := the first
 * valid point (receiver.cpp:145), build of the rest.  Returns like gndt_build. */
int gndt_build_cloud(gndt_handle* h, const void* raw_host, size_t n, const gndt_point_layout* layout);

/* ---- host key codec (consumers call transMortonXYZ on pos/goal: map2D.h:1071,1293; GlobalPlan.h:56) */
/* `transMortonXYZ` (map2D.h:950-976): quadrant letter, 1-based indices, signed z level and the
 * map key string (letter + decimal Morton, <= 12 chars + NUL). */
int gndt_trans_morton_xyz(const float origin[3], float grid_len, float z_len, const float p[3],
                          char* quadrant, int32_t* nx, int32_t* ny, int32_t* sz, char key_out[16]);
/* `countMorton` (Stopwatch.h:116-147), decimal string of the 32-bit interleave. */
int gndt_count_morton(int32_t a, int32_t b, char out[16]);
/* `mortonToXY` (Stopwatch.h:171-189). */
int gndt_morton_to_xy(int32_t morton, int32_t* a, int32_t* b);
/* Packed 64-bit node key used by gndt_stats: bits 63..43 sx+2^20, 42..22 sy+2^20, 21..0 sz+2^21. */
uint64_t gndt_pack_key(int32_t sx, int32_t sy, int32_t sz);
void gndt_unpack_key(uint64_t key, int32_t* sx, int32_t* sy, int32_t* sz);

/* ---- phase timing (bench.py / profiling) ------------------------------------------------------ */
/* With profiling on, build/accumulate/finalize record HIP events on the launch stream around each
 * phase; gndt_get_phase_times waits for them and returns milliseconds (-1 = phase did not run).
 * Phase names depend on the strategy the last build used (gndt_last_strategy):
 *   ATOMIC:    [0] clear  [1] accumulate  [2] columns  [3] rows  [4] bitmap_scan  [5] rank
 *              [6] column_scan  [7] dest  [8] emit   (bitmap_scan = prefix of the per-word column weights; rank and
 *              column_scan are empty since the ordering needs neither: kept so that phase indices stay put)
 *   PARTITION_EXACT: [0] clear  [1] hist  [2] offsets  [3] scatter  [4] bucket_build  [5] bitmap_scan
 *              [6] rank  [7] column_scan  [8] dest  [9] emit
 *   PARTITION (two-level): as PARTITION_EXACT with [1] level1  [2] (unused)  [3] level2 */
#define GNDT_NUM_PHASES 10
/* enable: 0 off, 1 events around every phase, 2 only around the dominant phase of the strategy in use
 * (bucket_build / accumulate): two events per build instead of eleven. */
int gndt_set_profiling(gndt_handle* h, int enable);
int gndt_get_phase_times(gndt_handle* h, double ms_out[GNDT_NUM_PHASES]);
/* GNDT_STRATEGY_ATOMIC, _PARTITION (two-level, hashed buckets), _PARTITION_BLOCKED (two-level, spatial blocks as buckets), _PARTITION_ONE_LEVEL,
 * _PARTITION_EXACT or _TILE: what the last build actually ran (AUTO resolves, and PARTITION falls back to ATOMIC when a bucket does not
 * fit in LDS). */
/* The measurement AUTO bases that choice on, for logs and tuning: `tiles` tiles of 2048 consecutive points spread over the
 * cloud; *points_per_partial = points looked at / distinct nodes met per tile.  Waits for the result. */
int gndt_locality_sample(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint32_t tiles, double* points_per_partial,
                         void* hip_stream);
int gndt_last_strategy(const gndt_handle* h);
/* Diagnostic (not for timed runs): after gndt_debug_enable_stamps(1) the bucket kernel
 * stamps the shader clock at its phase boundaries; this returns the mean cycles per bucket of
 * [0] clear [1] accumulate [2] columns [3] labels [4] order [5] emit, then the accumulate phase split into
 * [6] load wait [7] classify [8] scan+scatter [9] reduce (first chunk), and the bucket count. */
int gndt_debug_bucket_phases(gndt_handle* h, double cycles_out[10], uint32_t* buckets_out);
/* How many times this handle has re-run a PARTITION build because an LDS table, a partition region or the staging rows were
 * too small (each is a whole extra build).  A build is launched without waiting and its flags are only looked at by the call
 * that needs the result: a benchmark that enqueues builds back to back should check that this does not move. */
int gndt_debug_retry_count(gndt_handle* h, uint64_t* retries);
/* Switch the stamps on or off for the builds that follow (process-wide). */
int gndt_debug_enable_stamps(int on);
/* The library reads NO environment variable: what used to be GNDT_VERBOSE / GNDT_TILE_RATIO / GNDT_COST_WG (rounds 1-5) is set
 * here, process-wide, for the calls that follow.  Returns GNDT_ERR_INVALID for an unknown option or a value out of range.
 *   GNDT_DEBUG_VERBOSE            value != 0: one stderr line per resolved partition build (fullest region, flags, re-runs, second
 *                                 pass, node sketch) and per locality sample                                        default 0
 *   GNDT_DEBUG_TILE_RATIO         points per partial from which strategy AUTO takes TILE (tools/calibrate_tile.py sweeps it);
 *                                 value >= 1                                                                        default 48
 *   GNDT_DEBUG_COST_ONE_WORKGROUP value == 0: gndt_compute_cost launches every layer on its own, the one-workgroup kernel that walks
 *                                 the narrow layers is not used (tests run the flood both ways)                     default 1 */
#define GNDT_DEBUG_VERBOSE 1
#define GNDT_DEBUG_TILE_RATIO 2
#define GNDT_DEBUG_COST_ONE_WORKGROUP 3
int gndt_debug_set_option(int option, double value);
/* The bucket kernel finds a node through a 21-bit fingerprint of its key and confirms it with the key itself; a bucket in
 * which a fingerprint named the wrong node (~1 in 10^4) is accumulated a second time with every probe confirmed.  Tests narrow
 * the fingerprint (0 .. 21 bits, process-wide, for the builds that follow) so that this happens in every bucket;
 * gndt_debug_fp_clashes: buckets of the last resolved PARTITION build that took the second pass. */
int gndt_debug_set_fp_bits(int bits);
int gndt_debug_fp_clashes(gndt_handle* h, uint64_t* buckets);
/* Buckets of the last resolved PARTITION build whose 512-slot LDS table overflowed and that were done again by the bucket kernel's
 * second pass (1024-slot tables, those buckets only) instead of the whole build being re-run (gndt_debug_retry_count). */
int gndt_debug_second_pass_buckets(gndt_handle* h, uint64_t* buckets);
/* Tests of the sharded builds: the next allocation at `site` on this handle fails once, as if the device were out of memory —
 * 1: the receive buffer of gndt_build_owned_device's exchange, 2: its column-pair buffers, 3: the buffers of
 * gndt_gather_owned_map_device, 4: the fixed-size message buffers of a communicator's first owned build (0: none).  Every rank
 * of the build must then return — the failing one GNDT_ERR_NOMEM, the others GNDT_ERR_PEER — instead of waiting in a collective. */
int gndt_debug_fail_next_alloc(gndt_handle* h, int site);

/* Library / device information for logs: returns 0 and fills what it can. */
int gndt_device_info(int32_t device_id, char name_out[128], int32_t* compute_units, uint64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GNDT_H */
