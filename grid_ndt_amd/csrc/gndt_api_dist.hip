// gndt_api_dist.hip — one global map from a sharded cloud: shard -> statistics (PARTITION pipeline, no node table), and merged statistics (sorted by key) -> map.
#include "gndt_handle.hpp"
#include "gndt_table.hpp"
#include "gndt_exchange.hpp"

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

using namespace gndt;
using namespace gndt_host;

// Ranks that are THREADS of one process (gndt_comm_create_threads): the collectives of the owner-partitioned build as device
// copies between the ranks' buffers, with a host barrier on either side.  One GPU can then play a whole node — the tests run
// gndt_build_owned_device itself with 2 .. 8 ranks on the single-GPU box — and a process that drives several handles from
// several threads needs no RCCL.
struct ThreadGroup {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> send;                  // per rank: what it publishes for the collective in flight
    std::vector<const uint64_t*> off, cnt;          // exchange: byte offset / byte count of the run for every destination
    std::vector<int> ok;                            // per rank: it reached the collective in good order (0: its buffers are not to be read)
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

struct gndt_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    uint64_t owned_builds = 0;                      // owned builds all ranks of this communicator entered together (gndt_build_owned_device)
    std::shared_ptr<ThreadGroup> threads;           // set: the ranks are threads of this process, nccl is not used
};

namespace {
thread_local std::string g_comm_error;

#define RCCL_TRY(h, expr)                                                                                \
    do {                                                                                                 \
        ncclResult_t r__ = (expr);                                                                       \
        if (r__ != ncclSuccess) {                                                                        \
            (h)->err = std::string(#expr) + ": " + rccl().GetErrorString(r__);                           \
            return GNDT_ERR_HIP;                                                                         \
        }                                                                                                \
    } while (0)

ncclDataType_t nccl_type_of(size_t elem_bytes) { return elem_bytes == 8 ? ncclUint64 : ncclUint32; }

// every rank's `count` elements of `elem_bytes` (4 or 8) -> all of them, in rank order, on every rank.
// Thread ranks: a rank never returns between the two barriers of a collective (the others would wait for it for ever, and read
// buffers it has freed): what goes wrong in between is recorded, the second barrier is passed, then it is reported — and a
// rank that arrives in trouble says so (ThreadGroup::ok), which every rank sees and reports as GNDT_ERR_PEER.
int comm_all_gather(gndt_handle* h, gndt_comm* c, const void* send, void* recv, size_t count, size_t elem_bytes, hipStream_t s) {
    if (!c->threads) {
        RCCL_TRY(h, rccl().AllGather(send, recv, count, nccl_type_of(elem_bytes), c->nccl, s));
        return GNDT_OK;
    }
    ThreadGroup& G = *c->threads;
    const size_t bytes = count * elem_bytes;
    int rc = GNDT_OK;
    if (hipStreamSynchronize(s) != hipSuccess) { h->err = "hipStreamSynchronize failed before a collective"; rc = GNDT_ERR_HIP; }   // what this rank publishes is complete
    G.send[c->rank] = send; G.ok[c->rank] = rc == GNDT_OK;
    G.barrier();
    bool peers_ok = true;
    for (int q = 0; q < c->world; ++q) peers_ok = peers_ok && G.ok[q];
    if (rc == GNDT_OK && peers_ok) {
        for (int q = 0; q < c->world && rc == GNDT_OK; ++q) {
            void* dst = static_cast<char*>(recv) + (size_t)q * bytes;
            if (bytes && dst != G.send[q] && hipMemcpyAsync(dst, G.send[q], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                h->err = "hipMemcpyAsync failed inside a collective"; rc = GNDT_ERR_HIP;
            }
        }
        if (hipStreamSynchronize(s) != hipSuccess && rc == GNDT_OK) { h->err = "hipStreamSynchronize failed inside a collective"; rc = GNDT_ERR_HIP; }
    }
    G.barrier();                                               // everybody has read: the buffers may change again
    if (rc == GNDT_OK && !peers_ok) { h->err = "another rank failed on its way into a collective"; rc = GNDT_ERR_PEER; }
    return rc;
}

// all-to-all of runs: rank r sends cnt_bytes[q] bytes at send_base + off_bytes[q] to every q != r and receives the run every
// q holds for it at recv_base + recv_off_bytes[q] (recv_cnt_bytes[q] bytes: the same number the sender counted)
int comm_exchange(gndt_handle* h, gndt_comm* c, const char* send_base, const uint64_t* off_bytes, const uint64_t* cnt_bytes,
                  char* recv_base, const uint64_t* recv_off_bytes, const uint64_t* recv_cnt_bytes, hipStream_t s) {
    const int W = c->world, me = c->rank;
    if (!c->threads) {
        // A failing Send / Recv must not leave the group open (round 3 returned from inside it: the communicator was then stuck
        // in a group for good): the first error is kept, nothing more is added, the group is CLOSED, then the error is reported.
        RCCL_TRY(h, rccl().GroupStart());
        ncclResult_t first = ncclSuccess;
        const char* what = "";
        for (int r = 0; r < W && first == ncclSuccess; ++r) {
            if (r == me) continue;
            if (cnt_bytes[r]) { first = rccl().Send(send_base + off_bytes[r], (size_t)(cnt_bytes[r] / 4), ncclFloat, r, c->nccl, s); what = "ncclSend"; }
            if (first == ncclSuccess && recv_cnt_bytes[r]) {
                first = rccl().Recv(recv_base + recv_off_bytes[r], (size_t)(recv_cnt_bytes[r] / 4), ncclFloat, r, c->nccl, s); what = "ncclRecv";
            }
        }
        const ncclResult_t ended = rccl().GroupEnd();
        if (first != ncclSuccess) { h->err = std::string(what) + " inside the exchange: " + rccl().GetErrorString(first); return GNDT_ERR_HIP; }
        if (ended != ncclSuccess) { h->err = std::string("ncclGroupEnd: ") + rccl().GetErrorString(ended); return GNDT_ERR_HIP; }
        return GNDT_OK;
    }
    ThreadGroup& G = *c->threads;
    int rc = GNDT_OK;
    if (hipStreamSynchronize(s) != hipSuccess) { h->err = "hipStreamSynchronize failed before a collective"; rc = GNDT_ERR_HIP; }
    G.send[me] = send_base; G.off[me] = off_bytes; G.cnt[me] = cnt_bytes; G.ok[me] = rc == GNDT_OK;
    G.barrier();
    bool peers_ok = true;
    for (int q = 0; q < W; ++q) peers_ok = peers_ok && G.ok[q];
    if (rc == GNDT_OK && peers_ok) {
        for (int q = 0; q < W && rc == GNDT_OK; ++q) {
            if (q == me || !recv_cnt_bytes[q]) continue;
            if (G.cnt[q][me] != recv_cnt_bytes[q]) { h->err = "exchange: the sender's count differs from the receiver's"; rc = GNDT_ERR_INVALID; break; }
            if (hipMemcpyAsync(recv_base + recv_off_bytes[q], static_cast<const char*>(G.send[q]) + G.off[q][me], recv_cnt_bytes[q],
                               hipMemcpyDeviceToDevice, s) != hipSuccess) { h->err = "hipMemcpyAsync failed inside a collective"; rc = GNDT_ERR_HIP; }
        }
        if (hipStreamSynchronize(s) != hipSuccess && rc == GNDT_OK) { h->err = "hipStreamSynchronize failed inside a collective"; rc = GNDT_ERR_HIP; }
    }
    G.barrier();                                               // (the senders' offset / count vectors are read until here)
    if (rc == GNDT_OK && !peers_ok) { h->err = "another rank failed on its way into a collective"; rc = GNDT_ERR_PEER; }
    return rc;
}

// sum over the ranks of `send` ([W x count] u32), every rank keeping its own segment of `count` elements in `recv`
int comm_reduce_scatter_u32(gndt_handle* h, gndt_comm* c, const uint32_t* send, uint32_t* recv, size_t count, hipStream_t s) {
    const int W = c->world, me = c->rank;
    if (!c->threads) {
        if (rccl().ReduceScatter) {
            RCCL_TRY(h, rccl().ReduceScatter(send, recv, count, ncclUint32, ncclSum, c->nccl, s));
        } else {                                     // (same result through the collective every RCCL has; the array is small)
            RCCL_TRY(h, rccl().AllReduce(send, const_cast<uint32_t*>(send), count * (size_t)W, ncclUint32, ncclSum, c->nccl, s));
            HIP_TRY(h, hipMemcpyAsync(recv, send + (size_t)me * count, count * 4, hipMemcpyDeviceToDevice, s));
        }
        return GNDT_OK;
    }
    ThreadGroup& G = *c->threads;
    int rc = GNDT_OK;
    if (hipStreamSynchronize(s) != hipSuccess) { h->err = "hipStreamSynchronize failed before a collective"; rc = GNDT_ERR_HIP; }
    G.send[me] = send; G.ok[me] = rc == GNDT_OK;
    G.barrier();
    bool peers_ok = true;
    for (int q = 0; q < W; ++q) peers_ok = peers_ok && G.ok[q];
    if (rc == GNDT_OK && peers_ok && count) {
        for (int q = 0; q < W && rc == GNDT_OK; ++q) {
            const uint32_t* src = static_cast<const uint32_t*>(G.send[q]) + (size_t)me * count;
            if (q == 0) { if (hipMemcpyAsync(recv, src, count * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = GNDT_ERR_HIP; }
            else hipLaunchKernelGGL(k_add_u32, dim3(grid_for(count)), dim3(256), 0, s, recv, src, (uint64_t)count);
        }
        if (hipStreamSynchronize(s) != hipSuccess && rc == GNDT_OK) rc = GNDT_ERR_HIP;
        if (rc) h->err = "a device copy failed inside a collective";
    }
    G.barrier();
    if (rc == GNDT_OK && !peers_ok) { h->err = "another rank failed on its way into a collective"; rc = GNDT_ERR_PEER; }
    return rc;
}
// in-place all-reduce of `count` doubles (sum; the ranks add in rank order: every rank gets the same bits) or u32 (min)
int comm_all_reduce(gndt_handle* h, gndt_comm* c, void* buf, size_t count, bool f64_sum, hipStream_t s) {
    if (!c->threads) {
        if (f64_sum) RCCL_TRY(h, rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, c->nccl, s));
        else RCCL_TRY(h, rccl().AllReduce(buf, buf, count, ncclUint32, ncclMin, c->nccl, s));
        return GNDT_OK;
    }
    ThreadGroup& G = *c->threads;
    auto& X = h->exch;
    const int W = c->world, me = c->rank;
    const size_t bytes = count * (f64_sum ? 8 : 4);
    int rc = grow_buf(h, X.red_tmp, X.red_tmp_cap, (uint64_t)(bytes + 7) / 8 + 1);      // (before the first barrier: nobody waits yet)
    if (rc == GNDT_OK && hipStreamSynchronize(s) != hipSuccess) { h->err = "hipStreamSynchronize failed before a collective"; rc = GNDT_ERR_HIP; }
    G.send[me] = buf; G.ok[me] = rc == GNDT_OK;
    G.barrier();
    bool peers_ok = true;
    for (int q = 0; q < W; ++q) peers_ok = peers_ok && G.ok[q];
    const bool go = rc == GNDT_OK && peers_ok && count;
    if (go) {
        for (int q = 0; q < W && rc == GNDT_OK; ++q) {
            if (q == 0) { if (hipMemcpyAsync(X.red_tmp, G.send[0], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = GNDT_ERR_HIP; }
            else if (f64_sum) hipLaunchKernelGGL(k_add_f64, dim3(grid_for(count)), dim3(256), 0, s, X.red_tmp, static_cast<const double*>(G.send[q]), (uint64_t)count);
            else hipLaunchKernelGGL(k_min_u32, dim3(grid_for(count)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(X.red_tmp), static_cast<const uint32_t*>(G.send[q]), (uint64_t)count);
        }
        if (hipStreamSynchronize(s) != hipSuccess && rc == GNDT_OK) rc = GNDT_ERR_HIP;
        if (rc) h->err = "a device copy failed inside a collective";
    }
    G.barrier();                                               // everybody has read everybody's contribution: the result may go in place
    if (go && rc == GNDT_OK && hipMemcpyAsync(buf, X.red_tmp, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { h->err = "hipMemcpyAsync failed after a collective"; rc = GNDT_ERR_HIP; }
    if (rc == GNDT_OK && !peers_ok) { h->err = "another rank failed on its way into a collective"; rc = GNDT_ERR_PEER; }
    return rc;
}
}  // namespace

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


// ---------------------------------------------------------------------------------------------
// the exchange behind the C ABI (gndt_exchange.hpp): communicator helpers and the whole sharded build
// ---------------------------------------------------------------------------------------------
const char* gndt_comm_last_error(void) { return g_comm_error.c_str(); }

int gndt_comm_unique_id(char id_out[GNDT_COMM_ID_BYTES]) {
    if (!id_out) return GNDT_ERR_INVALID;
    if (!rccl().ok()) { g_comm_error = "librccl.so not found (dlopen)"; return GNDT_ERR_NO_DEVICE; }
    static_assert(sizeof(ncclUniqueId) <= GNDT_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_error = rccl().GetErrorString(r); return GNDT_ERR_HIP; }
    memset(id_out, 0, GNDT_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return GNDT_OK;
}

int gndt_comm_create(const char id[GNDT_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device_id, gndt_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { g_comm_error = "bad argument"; return GNDT_ERR_INVALID; }
    *out = nullptr;
    if (!rccl().ok()) { g_comm_error = "librccl.so not found (dlopen)"; return GNDT_ERR_NO_DEVICE; }
    if (hipSetDevice(device_id) != hipSuccess) { g_comm_error = "hipSetDevice failed"; return GNDT_ERR_HIP; }
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof nid);
    gndt_comm* c = new (std::nothrow) gndt_comm;
    if (!c) return GNDT_ERR_NOMEM;
    const ncclResult_t r = rccl().CommInitRank(&c->nccl, world, nid, rank);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r); delete c; return GNDT_ERR_HIP; }
    c->rank = rank; c->world = world; c->device = device_id;
    *out = c;
    return GNDT_OK;
}

int gndt_comm_create_threads(int32_t world, int32_t device_id, gndt_comm** out) {
    if (!out || world < 1 || world > 1024) { g_comm_error = "bad argument"; return GNDT_ERR_INVALID; }
    auto G = std::make_shared<ThreadGroup>();
    G->world = world;
    G->send.assign((size_t)world, nullptr); G->off.assign((size_t)world, nullptr); G->cnt.assign((size_t)world, nullptr);
    G->ok.assign((size_t)world, 1);
    for (int r = 0; r < world; ++r) {
        out[r] = new (std::nothrow) gndt_comm;
        if (!out[r]) { for (int q = 0; q < r; ++q) delete out[q]; return GNDT_ERR_NOMEM; }
        out[r]->rank = r; out[r]->world = world; out[r]->device = device_id; out[r]->threads = G;
    }
    return GNDT_OK;
}

void gndt_comm_destroy(gndt_comm* c) {
    if (!c) return;
    if (c->nccl && rccl().ok()) (void)rccl().CommDestroy(c->nccl);
    delete c;
}

int gndt_build_global_device(gndt_handle* h, gndt_comm* c, const void* shard_xyz_dev, size_t n, size_t stride_bytes,
                             uint64_t first_idx_base, uint64_t total_points, gndt_exchange_times* times, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!c || (!c->nccl && !c->threads)) { h->err = "no communicator"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    auto& X = h->exch;
    hipEvent_t* ev = X.ev;                     // (kept on the handle: an early error return leaks nothing)
    if (times) for (int i = 0; i < 4; ++i) if (!ev[i]) HIP_TRY(h, hipEventCreate(&ev[i]));
    auto stamp = [&](int i) { if (times) (void)hipEventRecord(ev[i], s); };
    stamp(0);
    // 1. this rank's shard -> statistics of its occupied nodes (the partition pipeline's statistics epilogue).  A failure here (a
    //    point outside the key range, a capacity error) does not end the call: the other ranks are on their way into the
    //    collectives, so this rank's code travels with its node count and ALL ranks leave together after the first all-gather.
    gndt_stats st{};
    const int shard_rc = gndt_shard_stats_device(h, shard_xyz_dev, n, stride_bytes, first_idx_base, &st, hip_stream);
    const std::string shard_err = shard_rc ? h->err : std::string();
    stamp(1);
    const uint32_t m = shard_rc ? 0u : (uint32_t)st.num_nodes;
    // 2. node counts of every rank (one tiny all-gather; the host needs them to size the key exchange), status in the top byte
    const int W = c->world;
    if (W > 1024) { h->err = "more than 1024 ranks"; return GNDT_ERR_INVALID; }
    if ((rc = grow_buf(h, X.d_counts, X.counts_cap, (uint64_t)W))) return rc;
    if (!X.h_counts) HIP_TRY(h, hipHostMalloc(&X.h_counts, 1024 * sizeof(unsigned long long)));
    X.h_counts[c->rank] = (unsigned long long)m | ((unsigned long long)(shard_rc & 0xFF) << 56);
    HIP_TRY(h, hipMemcpyAsync(X.d_counts + c->rank, X.h_counts + c->rank, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    if ((rc = comm_all_gather(h, c, X.d_counts + c->rank, X.d_counts, 1, 8, s))) return rc;
    HIP_TRY(h, hipMemcpyAsync(X.h_counts, X.d_counts, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int r = 0; r < W; ++r) {
        const int code = (int)(X.h_counts[r] >> 56);
        if (!code) continue;
        if (shard_rc) { h->err = shard_err; return shard_rc; }
        h->err = "rank " + std::to_string(r) + " could not build its shard (error " + std::to_string(code) + "): the global build was abandoned by all ranks";
        return GNDT_ERR_PEER;
    }
    uint64_t m_max = 1;
    for (int r = 0; r < W; ++r) m_max = std::max<uint64_t>(m_max, X.h_counts[r]);
    // 3. every rank's keys, padded to the longest list -> sorted unique union = canonical node order (identical on all ranks)
    const uint64_t all = m_max * (uint64_t)W;
    if (all >= 0xFFFFFFFFull) { h->err = "too many nodes for the exchange"; return GNDT_ERR_CAPACITY; }
    if ((rc = grow_buf(h, X.keys_in, X.keys_in_cap, m_max))) return rc;
    if ((rc = grow_buf(h, X.keys_all, X.keys_all_cap, all))) return rc;
    if ((rc = grow_buf(h, X.keys_sorted, X.keys_sorted_cap, all))) return rc;
    if ((rc = grow_buf(h, X.canon, X.canon_cap, all))) return rc;
    if (!X.d_unique) HIP_TRY(h, hipMalloc(&X.d_unique, sizeof(unsigned int)));
    if (m) HIP_TRY(h, hipMemcpyAsync(X.keys_in, st.key, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_exchange_pad, dim3(grid_for(m_max)), dim3(256), 0, s, X.keys_in, m, (uint32_t)m_max);
    if ((rc = comm_all_gather(h, c, X.keys_in, X.keys_all, (size_t)m_max, 8, s))) return rc;
    size_t tmp = 0, tmp2 = 0;
    HIP_TRY(h, rocprim::radix_sort_keys(nullptr, tmp, X.keys_all, X.keys_sorted, (size_t)all, 0, 64, s));
    HIP_TRY(h, rocprim::unique(nullptr, tmp2, X.keys_sorted, X.canon, X.d_unique, (size_t)all, rocprim::equal_to<uint64_t>(), s));
    if ((rc = grow_buf(h, X.scratch, X.scratch_cap, (uint64_t)std::max(tmp, tmp2) + 256))) return rc;
    size_t tb = (size_t)X.scratch_cap;
    HIP_TRY(h, rocprim::radix_sort_keys(X.scratch, tb, X.keys_all, X.keys_sorted, (size_t)all, 0, 64, s));
    tb = (size_t)X.scratch_cap;
    HIP_TRY(h, rocprim::unique(X.scratch, tb, X.keys_sorted, X.canon, X.d_unique, (size_t)all, rocprim::equal_to<uint64_t>(), s));
    unsigned int n_unique = 0;
    HIP_TRY(h, hipMemcpyAsync(&n_unique, X.d_unique, sizeof n_unique, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    // (the pad value sorts last: drop it if some rank had fewer nodes than the longest list)
    bool padded = false;
    for (int r = 0; r < W; ++r) padded = padded || X.h_counts[r] < m_max;
    if (m_max == 1) { bool any = false; for (int r = 0; r < W; ++r) any = any || X.h_counts[r] > 0; if (!any) padded = true; }
    const uint32_t C = n_unique - (padded && n_unique ? 1u : 0u);
    // 4. local statistics scattered into the canonical order, ONE packed sum all-reduce and one min all-reduce
    if ((rc = grow_buf(h, X.packed, X.packed_cap, (uint64_t)std::max<uint32_t>(C, 1) * kExWidth))) return rc;
    if ((rc = grow_buf(h, X.pfirst, X.pfirst_cap, (uint64_t)std::max<uint32_t>(C, 1)))) return rc;
    if ((rc = grow_buf(h, X.r_sums, X.r_sums_cap, (uint64_t)std::max<uint32_t>(C, 1) * 9))) return rc;
    if ((rc = grow_buf(h, X.r_count, X.r_count_cap, (uint64_t)std::max<uint32_t>(C, 1)))) return rc;
    if (!X.d_missing) { HIP_TRY(h, hipMalloc(&X.d_missing, sizeof(uint32_t))); HIP_TRY(h, hipMemsetAsync(X.d_missing, 0, sizeof(uint32_t), s)); }
    if (C) {
        hipLaunchKernelGGL(k_exchange_init, dim3(grid_for((uint64_t)C * kExWidth)), dim3(256), 0, s, X.packed, X.pfirst, C);
        if (m) hipLaunchKernelGGL(k_exchange_scatter, dim3(grid_for(m)), dim3(256), 0, s, (const uint64_t*)st.key, (const double*)st.sums,
                                  (const uint32_t*)st.count, (const uint32_t*)st.first_idx, m, (const uint64_t*)X.canon, C, X.packed,
                                  X.pfirst, X.d_missing);
        HIP_TRY(h, hipGetLastError());
        if ((rc = comm_all_reduce(h, c, X.packed, (size_t)C * kExWidth, true, s))) return rc;
        if ((rc = comm_all_reduce(h, c, X.pfirst, (size_t)C, false, s))) return rc;
        hipLaunchKernelGGL(k_exchange_unpack, dim3(grid_for(C)), dim3(256), 0, s, (const double*)X.packed, C, X.r_sums, X.r_count);
        HIP_TRY(h, hipGetLastError());
    }
    stamp(2);
    // 5. every rank: the reduced statistics (sorted by key) -> labels, order, rows: the map of the WHOLE cloud
    gndt_stats red;
    red.num_nodes = C; red.key = X.canon; red.sums = X.r_sums; red.count = X.r_count; red.first_idx = X.pfirst;
    rc = gndt_finalize_stats_device(h, &red, total_points, hip_stream);
    if (rc) return rc;
    stamp(3);
    // a key of this rank that the canonical union does not hold would have lost its statistics silently (k_exchange_scatter counts them)
    if (!X.h_bad) HIP_TRY(h, hipHostMalloc(&X.h_bad, sizeof(uint32_t)));
    HIP_TRY(h, hipMemcpyAsync(X.h_bad, X.d_missing, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (*X.h_bad) {
        const uint32_t lost = *X.h_bad;
        HIP_TRY(h, hipMemsetAsync(X.d_missing, 0, sizeof(uint32_t), s));
        h->err = std::to_string(lost) + " node(s) of this rank are missing from the union of the ranks' keys";
        return GNDT_ERR_INVALID;
    }
    if (times) {
        float a = 0, b = 0, d = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]); (void)hipEventElapsedTime(&d, ev[2], ev[3]);
        times->shard_ms = a; times->exchange_ms = b; times->finalize_ms = d;
        times->local_nodes = m; times->global_nodes = C; times->ranks = (uint32_t)W;
        times->bytes_reduced = (uint64_t)C * (kExWidth * 8 + 4);
    }
    return GNDT_OK;
}


// ---------------------------------------------------------------------------------------------
// owner-partitioned build (gndt_exchange.hpp, second half): the points travel
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

constexpr uint32_t kMaxRanks = 1024;
constexpr size_t kMatrixHostWords = (size_t)kMaxRanks * (kMaxRanks + 2);     // a rank's row: W send counts, its status word, its receive capacity

// kOwnerSamples evenly spaced points of the shard -> their blocks, with the shard size in front (X.owner_msg)
int owner_sample_launch(gndt_handle* h, const void* xyz, size_t n, size_t stride_bytes, hipStream_t s) {
    auto& X = h->exch;
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (!X.owner_msg) HIP_TRY(h, hipMalloc(&X.owner_msg, (size_t)kOwnerMsgWords * 4));
    const GridParams gp = grid_params(h);
    const float* p = static_cast<const float*>(xyz);
    if (stride_bytes == 12) hipLaunchKernelGGL(k_owner_sample<3>, dim3(kOwnerSamples / 256), dim3(256), 0, s, p, (uint64_t)n, gp, X.owner_msg);
    else hipLaunchKernelGGL(k_owner_sample<4>, dim3(kOwnerSamples / 256), dim3(256), 0, s, p, (uint64_t)n, gp, X.owner_msg);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}

// everybody's sample messages -> the block table of this build (X.bkey / X.bown); the same on every rank
int owner_map_launch(gndt_handle* h, const uint32_t* msgs, uint32_t W, hipStream_t s) {
    auto& X = h->exch;
    int rc;
    X.owner_map_world = 0;
    if (W < 2 || W > kOwnerMapMaxRanks) return GNDT_OK;               // (hash ownership)
    if (!X.bkey) HIP_TRY(h, hipMalloc(&X.bkey, (size_t)kOwnerSlots * 4));
    if (!X.bown) HIP_TRY(h, hipMalloc(&X.bown, (size_t)kOwnerSlots));
    if (!X.d_owner_full) HIP_TRY(h, hipMalloc(&X.d_owner_full, sizeof(uint32_t)));
    if ((rc = grow_buf(h, X.bcnt, X.bcnt_cap, (uint64_t)kOwnerSlots * W))) return rc;
    HIP_TRY(h, hipMemsetAsync(X.bkey, 0, (size_t)kOwnerSlots * 4, s));
    HIP_TRY(h, hipMemsetAsync(X.bcnt, 0, (size_t)kOwnerSlots * W * 4, s));
    HIP_TRY(h, hipMemsetAsync(X.d_owner_full, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_owner_vote, dim3(grid_for((uint64_t)W * kOwnerSamples)), dim3(256), 0, s, msgs, W, X.bkey, X.bcnt, kOwnerSlots - 1u, X.d_owner_full);
    hipLaunchKernelGGL(k_owner_pick, dim3(grid_for(kOwnerSlots)), dim3(256), 0, s, msgs, W, (const uint32_t*)X.bkey, (const uint32_t*)X.bcnt, kOwnerSlots, X.bown);
    HIP_TRY(h, hipGetLastError());
    X.owner_map_world = W;
    return GNDT_OK;
}

// shard -> records grouped by owner (X.send_recs); the run starts and the counters come back with the stream
// (owner_split_finish after the next wait)
int owner_split_launch(gndt_handle* h, const void* xyz, size_t n, size_t stride_bytes, uint64_t first_base, uint64_t total_points, uint32_t W,
                       hipStream_t s) {
    auto& q = h->part;
    auto& X = h->exch;
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (W < 1 || W > kMaxRanks) { h->err = "world must be 1..1024"; return GNDT_ERR_INVALID; }
    if (first_base + n > total_points || total_points >= 0x7FFFFFFFull) { h->err = "shard outside the cloud, or more than 2^31 points"; return GNDT_ERR_INVALID; }
    int rc;
    if ((rc = ensure_part_counters(h))) return rc;
    if (!X.d_split_cnt) HIP_TRY(h, hipMalloc(&X.d_split_cnt, sizeof(Counters)));
    if (!X.h_split_cnt) HIP_TRY(h, hipHostMalloc(&X.h_split_cnt, sizeof(Counters)));
    if (!X.h_matrix) HIP_TRY(h, hipHostMalloc(&X.h_matrix, (kMatrixHostWords + kMaxRanks + 1) * sizeof(uint32_t)));
    HIP_TRY(h, hipMemsetAsync(X.d_split_cnt, 0, sizeof(Counters), s));
    const uint32_t nwg = (uint32_t)std::min<uint64_t>((uint64_t)kPartWgs, std::max<uint64_t>(1, n / 8192));
    if ((rc = grow_buf(h, q.hist, q.hist_cap, (uint64_t)nwg * W))) return rc;
    if (W > q.bucket_cap) {
        release_device(h, q.totals);          // (shared with the exact partition, whose recorded builds write through them)
        release_device(h, q.bucket_base);
        q.totals = q.bucket_base = nullptr; q.bucket_cap = 0;
        HIP_TRY(h, hipMalloc(&q.totals, (size_t)W * 4));
        HIP_TRY(h, hipMalloc(&q.bucket_base, ((size_t)W + 1) * 4));
        q.bucket_cap = W;
    }
    const GridParams gp = grid_params(h);
    const uint32_t compress = total_points < (uint64_t)kWeightIndexLimit ? 1u : 0u;
    const float* p = static_cast<const float*>(xyz);
    // the block table all ranks made from the same samples (owner_map_launch), if it is for this many ranks; else hash ownership
    const OwnerMap M = (X.owner_map_world == W && X.bkey) ? OwnerMap{X.bkey, X.bown, kOwnerSlots - 1u, X.d_owner_full} : OwnerMap{nullptr, nullptr, 0u, nullptr};
    // ONE pass: level 1 of the partition pipeline with the owner as digit (k_part2_level1<.., OWNER>): owner r's run starts at
    // r * cap, every run could take the whole shard.  (W x shard records of address space: beyond 2^32, or with more ranks than
    // the tile sort's fan-out, the two-pass counting partition packs the runs instead.)
    constexpr uint64_t kTile1 = (uint64_t)kTileThreads * kTilePer1;
    const uint64_t cap = (uint64_t)n + kTile1;
    X.split_one_pass = W <= 256 && cap * W < 0xFFFFFFFFull && n >= (1u << 16);
    if (X.split_one_pass) {
        if ((rc = grow_buf(h, X.send_recs, X.send_cap, cap * W))) return rc;
        HIP_TRY(h, hipMemsetAsync(q.totals, 0, (size_t)W * 4, s));          // the runs' cursors = what every owner gets
        const uint32_t tiles = (uint32_t)((n + kTile1 - 1) / kTile1);
        const dim3 g1(std::max<uint32_t>(1, std::min<uint32_t>(tiles, tuning().l1_wgs)));
        if (stride_bytes == 12)
            hipLaunchKernelGGL((k_part2_level1<3, 256, false, true>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, (uint32_t)first_base, gp, W, W, 0u, 1u,
                               q.totals, (uint32_t)cap, (uint32_t*)nullptr, X.send_recs, X.d_split_cnt, q.d_pc, compress, M, (uint32_t*)nullptr,
                               (uint32_t*)nullptr, 0ull, FoldClear{}, 0);
        else
            hipLaunchKernelGGL((k_part2_level1<4, 256, false, true>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, (uint32_t)first_base, gp, W, W, 0u, 1u,
                               q.totals, (uint32_t)cap, (uint32_t*)nullptr, X.send_recs, X.d_split_cnt, q.d_pc, compress, M, (uint32_t*)nullptr,
                               (uint32_t*)nullptr, 0ull, FoldClear{}, 0);
        HIP_TRY(h, hipGetLastError());
        X.split_cap = cap;
        uint32_t* h_base = X.h_matrix + kMatrixHostWords;
        HIP_TRY(h, hipMemcpyAsync(h_base, q.totals, (size_t)W * 4, hipMemcpyDeviceToHost, s));
    } else {
        if ((rc = grow_buf(h, X.send_recs, X.send_cap, std::max<uint64_t>(n, 1)))) return rc;
        const size_t lds = (size_t)W * 4;
        if (stride_bytes == 12)
            hipLaunchKernelGGL(k_part_hist<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, W, q.hist, X.d_split_cnt, compress, kPartModeOwner, M);
        else
            hipLaunchKernelGGL(k_part_hist<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, W, q.hist, X.d_split_cnt, compress, kPartModeOwner, M);
        hipLaunchKernelGGL(k_part_offsets, dim3((W + 31) / 32), dim3(256), 0, s, q.hist, q.totals, W, nwg);
        if (stride_bytes == 12)
            hipLaunchKernelGGL(k_part_scatter<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, (uint32_t)first_base, gp, W, q.hist, q.totals,
                               q.bucket_base, X.send_recs, compress, kPartModeOwner, M);
        else
            hipLaunchKernelGGL(k_part_scatter<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, (uint32_t)first_base, gp, W, q.hist, q.totals,
                               q.bucket_base, X.send_recs, compress, kPartModeOwner, M);
        HIP_TRY(h, hipGetLastError());
        uint32_t* h_base = X.h_matrix + kMatrixHostWords;
        HIP_TRY(h, hipMemcpyAsync(h_base, q.bucket_base, ((size_t)W + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(h, hipMemcpyAsync(X.h_split_cnt, X.d_split_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    return GNDT_OK;
}
int owner_split_finish(gndt_handle* h, uint32_t W) {        // (after the stream has been waited for)
    auto& X = h->exch;
    const uint32_t* h_base = X.h_matrix + kMatrixHostWords;
    for (uint32_t r = 0; r < W; ++r) {
        if (X.split_one_pass) { X.send_off[r] = (uint64_t)r * X.split_cap; X.send_cnt[r] = h_base[r]; }
        else { X.send_off[r] = h_base[r]; X.send_cnt[r] = h_base[r + 1] - h_base[r]; }
    }
    if (X.h_split_cnt->err_key_range) {
        h->err = std::to_string(X.h_split_cnt->err_key_range) + " point(s) outside the key range";
        return GNDT_ERR_KEY_RANGE;
    }
    return GNDT_OK;
}

// the records this rank owns -> its map (launched, not awaited)
// (`recs2`: a second segment, preceded in its allocation by room for the first: gndt_handle::Pending)
int build_records(gndt_handle* h, const void* recs, size_t n, uint64_t total_points, hipStream_t s, const void* recs2 = nullptr, size_t n2 = 0) {
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    next_event_set(h);
    const int rc = partition_begin(h, recs, n, 16, s, true, std::max<uint64_t>(total_points, 1), recs2, n2);
    if (rc == -1) { h->err = "the records do not fit the partition pipeline"; return GNDT_ERR_CAPACITY; }
    return rc;
}

// columns of the finished local map as (first-seen index << 32 | node count) pairs in X.pairs; count in X.d_npairs[0]
// (the build may still be running: sizes come from the staging capacity, the kernel reads the node count on the device and
// does nothing if the build raised an overflow flag)
int owned_columns_enqueue(gndt_handle* h, hipStream_t s) {
    auto& q = h->part;
    auto& X = h->exch;
    int rc;
    if ((rc = grow_buf(h, X.pairs, X.pairs_cap, std::max<uint64_t>(q.stage_cap, 1)))) return rc;
    if ((rc = grow_buf(h, X.row_of_pair, X.row_of_pair_cap, X.pairs_cap))) return rc;
    if (!X.d_npairs) HIP_TRY(h, hipMalloc(&X.d_npairs, 2 * sizeof(uint32_t)));
    HIP_TRY(h, hipMemsetAsync(X.d_npairs, 0, 2 * sizeof(uint32_t), s));
    const uint32_t col_wgs = (uint32_t)std::min<uint64_t>(2048, (std::max<uint64_t>(q.stage_cap, 1) + kColChunk - 1) / kColChunk);
    hipLaunchKernelGGL(k_owned_columns, dim3(col_wgs), dim3(256), 0, s, h->out.first_idx, q.row_ncol,
                       h->d_cnt, q.d_pc, X.pairs, (uint32_t)X.pairs_cap, X.d_npairs, X.row_of_pair);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}
int owned_columns_launch(gndt_handle* h, hipStream_t s, uint32_t& ncols_host) {
    int rc = partition_resolve(h);
    if (rc) return rc;
    if (!h->results_valid || h->map_in_table) { h->err = "no finished PARTITION build on this handle"; return GNDT_ERR_INVALID; }
    if (h->h_cnt->err_key_range) { h->err = std::to_string(h->h_cnt->err_key_range) + " point(s) outside the key range"; return GNDT_ERR_KEY_RANGE; }
    ncols_host = h->h_cnt->num_columns;
    return owned_columns_enqueue(h, s);
}

// everybody's column pairs -> global row of every local row (X.global_row), totals of the whole map in X.d_totals[0..1]
int global_rows_launch(gndt_handle* h, const unsigned long long* all_pairs, uint64_t n_all, uint64_t total_points, hipStream_t s) {
    auto& q = h->part;
    auto& X = h->exch;
    int rc;
    const uint64_t words = (std::max<uint64_t>(total_points, 1) + 31) / 32 + 1;
    if (words > q.word_cap) { h->err = "total_points differs from the build's"; return GNDT_ERR_INVALID; }
    if (!X.d_totals) HIP_TRY(h, hipMalloc(&X.d_totals, 4 * sizeof(unsigned long long)));
    if (!X.h_totals) HIP_TRY(h, hipHostMalloc(&X.h_totals, 4 * sizeof(unsigned long long)));
    if (!X.d_npairs) HIP_TRY(h, hipMalloc(&X.d_npairs, 2 * sizeof(uint32_t)));
    if ((rc = grow_buf(h, X.global_row, X.global_row_cap, std::max<uint64_t>(h->h_cnt->num_nodes, 1)))) return rc;
    // the local order is finished (rows emitted): its arrays now take the column order of the WHOLE map
    if ((rc = grow_buf(h, X.gw, X.gw_cap, words))) return rc;
    hipLaunchKernelGGL(k_order_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, X.gw, (uint64_t)words, X.d_totals, X.d_npairs + 1);
    if (n_all)
        hipLaunchKernelGGL(k_pairs_note, dim3((uint32_t)std::min<uint64_t>(1024, (n_all + 255) / 256)), dim3(256), 0, s, all_pairs, n_all,
                           X.gw, q.ncol_at, words, X.d_totals, X.d_npairs + 1);
    hipLaunchKernelGGL(k_order_split, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, (const unsigned long long*)X.gw, (uint64_t)words, q.bitmap,
                       q.word_weight);
    const uint32_t nbw = (uint32_t)((words + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr, (uint32_t)words, q.bsum_words);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr, (uint32_t)words, q.bsum_words,
                       q.word_base);
    hipLaunchKernelGGL(k_global_rows, dim3(grid_for(std::max<uint64_t>(h->h_cnt->num_nodes, 1))), dim3(256), 0, s, h->out.first_idx, q.row_ncol, h->d_cnt,
                       q.bitmap, q.word_base, q.ncol_at, X.global_row);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}


// The same for gndt_build_owned_device, sliced over the ranks (gndt_exchange.hpp "the same, sliced"): rank t orders the columns
// first seen in its slice of the index range; two more small collectives instead of W times the atomics.  X.pairs holds this
// rank's pairs padded to m_max, all_pairs everybody's (rank r at r * m_max).
// `own_rows`: false for a rank whose build failed after its peers went on: it only plays its part in the collectives.
int global_rows_sliced(gndt_handle* h, gndt_comm* c, const unsigned long long* all_pairs, uint64_t m_max, uint64_t total_points, hipStream_t s,
                       bool own_rows = true) {
    auto& q = h->part;
    auto& X = h->exch;
    const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank;
    int rc;
    const uint64_t words = (std::max<uint64_t>(total_points, 1) + 31) / 32 + 1;
    if (words > q.word_cap) { h->err = "total_points differs from the build's"; return GNDT_ERR_INVALID; }
    const uint64_t slice_words = (words + W - 1) / W;
    const uint64_t lo = std::min<uint64_t>(words, (uint64_t)me * slice_words), hi = std::min<uint64_t>(words, lo + slice_words);
    const uint64_t n_all = m_max * W;
    if (!X.d_totals) HIP_TRY(h, hipMalloc(&X.d_totals, 4 * sizeof(unsigned long long)));
    if (!X.h_totals) HIP_TRY(h, hipHostMalloc(&X.h_totals, 4 * sizeof(unsigned long long)));
    if (!X.d_slice) HIP_TRY(h, hipMalloc(&X.d_slice, (2 + 3 * (size_t)kMaxRanks) * sizeof(unsigned long long)));
    unsigned long long *mine = X.d_slice, *all = X.d_slice + 2, *rows = X.d_slice + 2 + 2 * (size_t)kMaxRanks;
    if ((rc = grow_buf(h, X.global_row, X.global_row_cap, own_rows ? std::max<uint64_t>(h->h_cnt->num_nodes, 1) : 1))) return rc;
    if ((rc = grow_buf(h, X.gw, X.gw_cap, std::max<uint64_t>(slice_words, 1)))) return rc;
    if ((rc = grow_buf(h, X.place_all, X.place_all_cap, std::max<uint64_t>(n_all, 1)))) return rc;
    if ((rc = grow_buf(h, X.place_mine, X.place_mine_cap, std::max<uint64_t>(m_max, 1)))) return rc;
    // (the local order is finished — rows emitted —: its arrays now take this rank's slice of the column order of the WHOLE map)
    hipLaunchKernelGGL(k_order_clear, dim3(grid_for(slice_words, 256, 512)), dim3(256), 0, s, X.gw, (uint64_t)(hi - lo), mine, X.d_npairs + 1);
    hipLaunchKernelGGL(k_pairs_note_slice, dim3((uint32_t)std::min<uint64_t>(1024, (n_all + 255) / 256)), dim3(256), 0, s, all_pairs, n_all, words, lo, hi,
                       X.gw, q.ncol_at, mine, X.d_npairs + 1);
    hipLaunchKernelGGL(k_order_split, dim3(grid_for(slice_words, 256, 512)), dim3(256), 0, s, (const unsigned long long*)X.gw, (uint64_t)(hi - lo), q.bitmap,
                       q.word_weight);
    const uint32_t nbw = (uint32_t)std::max<uint64_t>(1, ((hi - lo) + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr, (uint32_t)(hi - lo), q.bsum_words);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr, (uint32_t)(hi - lo), q.bsum_words,
                       q.word_base);
    hipLaunchKernelGGL(k_pair_places, dim3((uint32_t)std::min<uint64_t>(2048, (n_all + 255) / 256)), dim3(256), 0, s, all_pairs, n_all, lo, hi,
                       (const uint32_t*)q.bitmap, (const uint32_t*)q.word_base, (const uint32_t*)q.ncol_at, X.place_all);
    HIP_TRY(h, hipGetLastError());
    // everybody's slice totals (rank t's pair at all + 2 t), and every rank's own places
    HIP_TRY(h, hipMemcpyAsync(all + 2 * (size_t)me, mine, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    if ((rc = comm_all_gather(h, c, all + 2 * (size_t)me, all, 2, 8, s))) return rc;
    if ((rc = comm_reduce_scatter_u32(h, c, X.place_all, X.place_mine, (size_t)m_max, s))) return rc;
    hipLaunchKernelGGL(k_slice_rows, dim3(1), dim3(1), 0, s, (const unsigned long long*)all, W, rows, X.d_totals);
    if (own_rows) hipLaunchKernelGGL(k_global_rows_sliced, dim3(grid_for(m_max)), dim3(256), 0, s, (const unsigned long long*)X.pairs, (const uint32_t*)X.row_of_pair,
                       (const uint32_t*)X.place_mine, (uint32_t)m_max, slice_words, (const unsigned long long*)rows, X.global_row);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}

}  // namespace

extern "C" {

int gndt_owner_of_columns(const int32_t* sx, const int32_t* sy, size_t n, uint32_t world, uint32_t* owner_out) {
    if ((!sx || !sy || !owner_out) && n) return GNDT_ERR_INVALID;
    if (world < 1 || world > kMaxRanks) return GNDT_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) owner_out[i] = owner_of(sx[i], sy[i], world);
    return GNDT_OK;
}

int gndt_owner_sample_device(gndt_handle* h, const void* shard_xyz_dev, size_t n, size_t stride_bytes, const uint32_t** msg_dev,
                             uint64_t* msg_words, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!msg_dev || !msg_words || (!shard_xyz_dev && n)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if ((rc = owner_sample_launch(h, shard_xyz_dev, n, stride_bytes, s))) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));
    *msg_dev = h->exch.owner_msg;
    *msg_words = kOwnerMsgWords;
    return GNDT_OK;
}

int gndt_owner_map_device(gndt_handle* h, const uint32_t* all_msgs_dev, uint32_t world, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!all_msgs_dev || world < 1 || world > kMaxRanks) { h->err = "bad argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if ((rc = owner_map_launch(h, all_msgs_dev, world, s))) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));             // (the caller's message buffer is free again)
    return GNDT_OK;
}

int gndt_owner_split_device(gndt_handle* h, const void* shard_xyz_dev, size_t n, size_t stride_bytes, uint64_t first_idx_base,
                            uint64_t total_points, uint32_t world, const void** records_dev, uint64_t* counts_host, uint64_t* offsets_host,
                            void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!records_dev || !counts_host || !offsets_host || (!shard_xyz_dev && n)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if ((rc = owner_split_launch(h, shard_xyz_dev, n, stride_bytes, first_idx_base, total_points, world, s))) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));
    if ((rc = owner_split_finish(h, world))) return rc;
    for (uint32_t r = 0; r < world; ++r) { counts_host[r] = h->exch.send_cnt[r]; offsets_host[r] = h->exch.send_off[r]; }
    *records_dev = h->exch.send_recs;
    return GNDT_OK;
}

int gndt_build_records_device(gndt_handle* h, const void* records_dev, size_t n_records, uint64_t total_points, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!records_dev && n_records) { h->err = "null input"; return GNDT_ERR_INVALID; }
    return build_records(h, records_dev, n_records, total_points, stream_of(h, hip_stream));
}

int gndt_build_records2_device(gndt_handle* h, const void* first_dev, size_t n_first, void* second_dev, size_t n_second,
                               uint64_t total_points, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if ((!first_dev && n_first) || (!second_dev && n_second)) { h->err = "null input"; return GNDT_ERR_INVALID; }
    return build_records(h, first_dev, n_first, total_points, stream_of(h, hip_stream), second_dev, n_second);
}

int gndt_owned_columns_device(gndt_handle* h, const uint64_t** pairs_dev, uint64_t* n_pairs, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!pairs_dev || !n_pairs) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    uint32_t nc = 0;
    if ((rc = owned_columns_launch(h, s, nc))) return rc;
    HIP_TRY(h, hipStreamSynchronize(s));
    *pairs_dev = reinterpret_cast<const uint64_t*>(h->exch.pairs);
    *n_pairs = nc;
    return GNDT_OK;
}

int gndt_owned_global_rows_device(gndt_handle* h, const uint64_t* all_pairs_dev, uint64_t n_all, uint64_t total_points,
                                  const uint32_t** global_row_dev, uint64_t* global_nodes, uint64_t* global_columns, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!global_row_dev || (!all_pairs_dev && n_all)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    if ((rc = partition_resolve(h))) return rc;
    if (!h->results_valid || h->map_in_table) { h->err = "no finished PARTITION build on this handle"; return GNDT_ERR_INVALID; }
    auto& X = h->exch;
    if ((rc = global_rows_launch(h, reinterpret_cast<const unsigned long long*>(all_pairs_dev), n_all, total_points, s))) return rc;
    if (!X.h_bad) HIP_TRY(h, hipHostMalloc(&X.h_bad, sizeof(uint32_t)));
    HIP_TRY(h, hipMemcpyAsync(X.h_totals, X.d_totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(X.h_bad, X.d_npairs + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    const uint32_t bad = *X.h_bad;
    if (bad) { h->err = std::to_string(bad) + " column pair(s) with an index beyond total_points"; return GNDT_ERR_INVALID; }
    *global_row_dev = X.global_row;
    if (global_nodes) *global_nodes = X.h_totals[0];
    if (global_columns) *global_columns = X.h_totals[1];
    X.owned_serial = h->result_serial; X.owned_world = 0;      // (gndt_owned_pack_rows_device may follow; the gather needs gndt_build_owned_device)
    return GNDT_OK;
}

namespace {
// One status word of every rank to every rank (an all-gather of 8 bytes per rank through buffers that exist since gndt_create) and
// the common verdict: GNDT_OK if all words are zero; a rank whose own word is not leaves with it (and its message), every other
// rank with GNDT_ERR_PEER — all of them at this point, so that nobody is left waiting in the collective that would follow.
int agree_on(gndt_handle* h, gndt_comm* c, hipStream_t s, int my_code, const std::string& my_msg) {
    auto& X = h->exch;
    const int W = c->world, me = c->rank;
    X.h_agree[kMaxRanks] = (unsigned long long)my_code;
    HIP_TRY(h, hipMemcpyAsync(X.d_agree + me, X.h_agree + kMaxRanks, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    { const int arc = comm_all_gather(h, c, X.d_agree + me, X.d_agree, 1, 8, s); if (arc) return arc; }
    HIP_TRY(h, hipMemcpyAsync(X.h_agree, X.d_agree, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (my_code) { h->err = my_msg; return my_code; }
    for (int r = 0; r < W; ++r)
        if (X.h_agree[r]) {
            h->err = "rank " + std::to_string(r) + " reported error " + std::to_string(X.h_agree[r]) + ": abandoned by all ranks";
            return GNDT_ERR_PEER;
        }
    return GNDT_OK;
}
bool injected_failure(gndt_handle* h, int site) {      // tests: gndt_debug_fail_next_alloc
    if (h->exch.inject_site != site) return false;
    h->exch.inject_site = 0;
    h->err = "allocation failure injected at site " + std::to_string(site) + " (gndt_debug_fail_next_alloc)";
    return true;
}
}  // namespace

int gndt_debug_fail_next_alloc(gndt_handle* h, int site) {
    if (!h || site < 0 || site > 4) return GNDT_ERR_INVALID;
    h->exch.inject_site = site;
    return GNDT_OK;
}

int gndt_build_owned_device(gndt_handle* h, gndt_comm* c, const void* shard_xyz_dev, size_t n, size_t stride_bytes,
                            uint64_t first_idx_base, uint64_t total_points, const uint32_t** global_row_dev,
                            gndt_owned_info* info, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!c || (!c->nccl && !c->threads)) { h->err = "no communicator"; return GNDT_ERR_INVALID; }
    const int W = c->world, me = c->rank;
    if (W > (int)kMaxRanks) { h->err = "world must be 1..1024"; return GNDT_ERR_INVALID; }
    if (W > 1 && !c->threads && !rccl().p2p()) { h->err = "this RCCL has no ncclSend / ncclRecv"; return GNDT_ERR_NO_DEVICE; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    auto& q = h->part;
    auto& X = h->exch;
    X.owned_serial = 0;
    hipEvent_t* ev = X.ev;                     // (kept on the handle: an early error return leaks nothing)
    if (info) for (int i = 0; i < 5; ++i) if (!ev[i]) HIP_TRY(h, hipEventCreate(&ev[i]));
    auto stamp = [&](int i) { if (info) (void)hipEventRecord(ev[i], s); };
    // ---- errors and collectives ----
    // The ranks meet in five collectives.  A rank that fails on its own (bad input, a point outside the key range, a build
    // that does not fit) must not simply return: the others would wait for it in the next collective.  Its error code
    // travels instead — with its row of the send-count matrix, then with its column message — and every rank that sees a
    // non-zero code leaves at that point: the failing rank with its own error, the others with GNDT_ERR_PEER.  What
    // cannot be reported this way is fatal for the group, as the loss of a rank is: a failing collective itself, and running
    // out of memory for the buffers allocated here, before the first collective, or between two collectives.
    int err = GNDT_OK;                         // this rank's first error; the protocol goes on with empty contributions
    std::string err_msg;
    auto note = [&](int code) { if (code && !err) { err = code; err_msg = h->err; } return code == GNDT_OK; };
    auto leave = [&]() { h->err = err_msg; return err; };
    auto peer_failed = [&](int r, unsigned long long code) {
        h->err = "rank " + std::to_string(r) + " reported error " + std::to_string(code) + ": the owner-partitioned build was abandoned by all ranks";
        return GNDT_ERR_PEER;
    };
    // ---- agreement rounds (round 5) ----
    // Buffers whose size the ranks learn from each other (what arrives in the exchange, everybody's column pairs) are grown
    // BETWEEN two collectives, and a rank that cannot grow one cannot take part in the collective that follows: returning there
    // left its peers waiting in that collective for ever (missing 5).  So every rank publishes the CAPACITY it has
    // with the message that tells the sizes (its row of the matrix, its column message); all ranks then know whether anybody has
    // to grow anything.  If nobody has — the steady state — nothing is added.  If somebody has, he tries, and ALL ranks meet in
    // one more tiny all-gather of status words (`agree`) before the collective that needs the buffer: a rank that failed
    // leaves with its own error, the others with GNDT_ERR_PEER, together.  The words travel through buffers allocated with the
    // handle (Exchange::d_agree), so a rank can say "I cannot go on" whatever else it failed to allocate.
    auto agree = [&]() -> int { return agree_on(h, c, s, err, err_msg); };
    auto injected = [&](int site) { return injected_failure(h, site); };
    const uint32_t RW = (uint32_t)W + 2u;      // a rank's row of the matrix: W send counts, its status word, its receive capacity
    // The fixed-size message buffers of this handle.  On a communicator's FIRST owned build every rank allocates them, and the
    // ranks agree on the outcome before the first collective that uses them; later builds find them in place (a fresh handle on
    // a used communicator is not covered: its failure here is fatal for the group, as the loss of a rank is).
    {
        auto small = [&]() -> int {
            if (injected(4)) return GNDT_ERR_NOMEM;
            int r2;
            if ((r2 = grow_buf(h, X.d_matrix, X.matrix_cap, (uint64_t)W * RW))) return r2;
            if (!X.h_matrix) HIP_TRY(h, hipHostMalloc(&X.h_matrix, (kMatrixHostWords + kMaxRanks + 1) * sizeof(uint32_t)));
            if (!X.d_colmsg) HIP_TRY(h, hipMalloc(&X.d_colmsg, kColMsgWords * (size_t)kMaxRanks * sizeof(unsigned long long)));
            if (!X.h_colmsg) HIP_TRY(h, hipHostMalloc(&X.h_colmsg, kColMsgWords * ((size_t)kMaxRanks + 1) * sizeof(unsigned long long)));
            if (!X.d_split_cnt) HIP_TRY(h, hipMalloc(&X.d_split_cnt, sizeof(Counters)));
            return GNDT_OK;
        };
        const int src = small();
        if (c->owned_builds == 0 && W > 1) {
            note(src);
            if ((rc = agree())) return rc;
        } else if (src) return src;
        ++c->owned_builds;                     // (all ranks pass this point together, or none does)
    }
    if (!shard_xyz_dev && n) { h->err = "null input"; note(GNDT_ERR_INVALID); }
    stamp(0);
    // 0. who owns what: everybody's samples (one fixed-size all-gather, no wait) -> the block table, identical on every rank
    X.owner_map_world = 0;
    if (W > 1 && W <= (int)kOwnerMapMaxRanks && tuning().owner_locality) {
        if (!X.owner_msg) HIP_TRY(h, hipMalloc(&X.owner_msg, (size_t)kOwnerMsgWords * 4));
        if ((rc = grow_buf(h, X.owner_msgs_all, X.owner_msgs_cap, (uint64_t)kOwnerMsgWords * W))) return rc;
        if (err || !note(owner_sample_launch(h, shard_xyz_dev, n, stride_bytes, s)))
            HIP_TRY(h, hipMemsetAsync(X.owner_msg, 0xFF, (size_t)kOwnerMsgWords * 4, s));      // (no samples; nobody will use the table)
        if ((rc = comm_all_gather(h, c, X.owner_msg, X.owner_msgs_all, (size_t)kOwnerMsgWords, 4, s))) return rc;
        if (!err) note(owner_map_launch(h, X.owner_msgs_all, (uint32_t)W, s));
    }
    // 1. split by owner; who sends how much to whom (W x W counts, and every rank's status word) follows on the stream
    if (!err) note(owner_split_launch(h, shard_xyz_dev, n, stride_bytes, first_idx_base, total_points, (uint32_t)W, s));
    stamp(1);
    uint32_t* my_row = X.d_matrix + (size_t)me * RW;
    if (!err) HIP_TRY(h, hipMemcpyAsync(my_row, q.totals, (size_t)W * 4, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(h, hipMemsetAsync(my_row, 0, (size_t)W * 4, s));
    hipLaunchKernelGGL(k_split_status, dim3(1), dim3(1), 0, s, (const Counters*)(err ? nullptr : X.d_split_cnt), (uint32_t)err, (uint32_t)GNDT_ERR_KEY_RANGE,
                       my_row + W, (uint32_t)std::min<uint64_t>(X.own_cap, 0xFFFFFFFFull));
    if ((rc = comm_all_gather(h, c, my_row, X.d_matrix, (size_t)RW, 4, s))) return rc;
    HIP_TRY(h, hipMemcpyAsync(X.h_matrix, X.d_matrix, (size_t)W * RW * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (!err) note(owner_split_finish(h, (uint32_t)W));        // (points outside the key range: the same verdict the status word carries)
    for (int r = 0; r < W; ++r) {
        const uint32_t code = X.h_matrix[(size_t)r * RW + W];
        if (code) return err ? leave() : peer_failed(r, code);
    }
    if (err) return leave();                                   // (cannot happen: this rank's own status word was zero)
    // 2. the runs themselves
    // The run this rank keeps stays where the split left it: the build reads two segments (it, and what arrives).  own_recs
    // holds [room for the kept run | the runs of the other ranks in rank order]; the room is only filled when the build
    // takes the counting partition, which wants one array.
    const uint64_t kept = X.send_cnt[me];
    std::vector<uint64_t> recv_off((size_t)W + 1, 0);
    recv_off[0] = kept;
    for (int r = 0; r < W; ++r) recv_off[r + 1] = recv_off[r] + (r == me ? 0 : X.h_matrix[(size_t)r * RW + me]);      // what rank r holds for me
    const uint64_t n_own = recv_off[W];
    {   // does ANY rank have to grow its receive buffer?  (every rank computes every rank's answer from the same matrix)
        bool somebody_grows = false;
        for (int r = 0; r < W && !somebody_grows; ++r) {
            uint64_t own_r = 0;
            for (int q2 = 0; q2 < W; ++q2) own_r += X.h_matrix[(size_t)q2 * RW + r];
            somebody_grows = std::max<uint64_t>(own_r, 1) > X.h_matrix[(size_t)r * RW + W + 1];
        }
        if (somebody_grows) {
            if (std::max<uint64_t>(n_own, 1) > X.own_cap) {
                // (room to spare: the next builds of a similar cloud find the buffer large enough and skip the agreement round)
                if (injected(1)) note(GNDT_ERR_NOMEM);
                else note(grow_buf(h, X.own_recs, X.own_cap, n_own + n_own / 8 + 1024));
            }
            if (W > 1 && (rc = agree())) return rc;
            if (err) return leave();
        }
    }
    uint64_t sent = 0, received = 0;
    if (W > 1) {
        std::vector<uint64_t> so((size_t)W), sc((size_t)W), ro((size_t)W), rcnt((size_t)W);
        for (int r = 0; r < W; ++r) {
            so[r] = X.send_off[r] * sizeof(float4); sc[r] = r == me ? 0 : X.send_cnt[r] * sizeof(float4);
            ro[r] = recv_off[r] * sizeof(float4); rcnt[r] = r == me ? 0 : (recv_off[r + 1] - recv_off[r]) * sizeof(float4);
            sent += sc[r]; received += rcnt[r];
        }
        if ((rc = comm_exchange(h, c, reinterpret_cast<const char*>(X.send_recs), so.data(), sc.data(), reinterpret_cast<char*>(X.own_recs), ro.data(),
                                rcnt.data(), s))) return rc;
    }
    stamp(2);
    // 3. the columns this rank owns, finished: the ordinary pipeline on the records.  The column pairs and every rank's
    //    message (column count, "my build has to be re-run", status) follow on the stream: ONE wait for the build and the
    //    messages.  If some rank's build overflowed, that rank re-runs it and ALL ranks repeat the round (they all saw the same
    //    messages); a rank that cannot go on says so in its message and all ranks leave.
    note(build_records(h, X.send_recs + X.send_off[me], (size_t)kept, total_points, s, X.own_recs + kept, (size_t)(n_own - kept)));
    unsigned long long* my_msg_host = X.h_colmsg + kColMsgWords * (size_t)kMaxRanks;       // (pinned: what a failed rank sends)
    uint32_t ncols = 0;
    uint64_t m_max = 1;
    bool peers_know = false;                   // this rank's error has been seen by everybody
    for (int round = 0;; ++round) {
        if (!err) note(owned_columns_enqueue(h, s));
        if (!err) {
            hipLaunchKernelGGL(k_owned_status, dim3(1), dim3(1), 0, s, (const uint32_t*)X.d_npairs, (const PartCounters*)q.d_pc, (const Counters*)h->d_cnt,
                               (unsigned long long)n_own, X.d_colmsg + kColMsgWords * me, (unsigned long long)X.pairs_cap, (unsigned long long)X.pairs_all_cap);
        } else {
            for (int k = 0; k < kColMsgWords; ++k) my_msg_host[k] = 0ull;
            my_msg_host[4] = (unsigned long long)err;
            HIP_TRY(h, hipMemcpyAsync(X.d_colmsg + kColMsgWords * me, my_msg_host, kColMsgWords * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
            peers_know = true;
        }
        if ((rc = comm_all_gather(h, c, X.d_colmsg + kColMsgWords * me, X.d_colmsg, kColMsgWords, 8, s))) return rc;
        HIP_TRY(h, hipMemcpyAsync(X.h_colmsg, X.d_colmsg, kColMsgWords * (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        if (!err) note(partition_resolve(h));                          // the wait; this rank's own overflow re-runs happen here
        HIP_TRY(h, hipStreamSynchronize(s));                           // (a repeated round finds the build resolved already: the messages still have to arrive)
        bool someone_failed = false;
        m_max = 1;
        for (int r = 0; r < W; ++r) {
            const unsigned long long code = X.h_colmsg[kColMsgWords * r + 4];
            if (code) return err ? leave() : peer_failed(r, code);
            someone_failed = someone_failed || X.h_colmsg[kColMsgWords * r + 1] != 0;
            m_max = std::max<uint64_t>(m_max, X.h_colmsg[kColMsgWords * r]);
        }
        if (err && !peers_know && !someone_failed) break;      // the build failed AFTER this round's message said it was fine: the
                                                               //   others go on to the pairs; this rank joins that collective empty-handed
        if (!someone_failed && !err) break;
        if (round >= 8 && !err) { h->err = "a rank's build keeps overflowing"; note(GNDT_ERR_CAPACITY); }
    }
    if (!err) ncols = h->h_cnt->num_columns;
    stamp(3);
    // 4. everybody's columns -> the global row of every local row
    {   // the pair buffers: does ANY rank have to grow one?  (capacities as every rank's last column message told them)
        bool somebody_grows = false;
        for (int r = 0; r < W && !somebody_grows; ++r)
            somebody_grows = m_max > X.h_colmsg[kColMsgWords * r + 6] || m_max * (uint64_t)W > X.h_colmsg[kColMsgWords * r + 7];
        if (somebody_grows) {
            // (err is 0 here, or the error of a build that failed AFTER this rank's last message: that one is told by the poison
            //  pair below — its rank still needs the buffers to play its part — and only a failure to grow is agreed on here)
            int grow_err = GNDT_OK;
            const uint64_t want = m_max + m_max / 8 + 64;      // (room to spare: similar clouds skip the round next time)
            if (injected(2)) grow_err = GNDT_ERR_NOMEM;
            if (!grow_err && m_max > X.pairs_cap) {             // (another rank owns more columns: a longer send buffer, contents kept)
                unsigned long long* bigger = nullptr;
                if (hipMalloc(&bigger, want * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); h->err = "hipMalloc of the column-pair buffer failed"; grow_err = GNDT_ERR_NOMEM; }
                else {
                    if (!err && ncols && X.pairs) HIP_TRY(h, hipMemcpyAsync(bigger, X.pairs, (size_t)ncols * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
                    HIP_TRY(h, hipStreamSynchronize(s));
                    release_device(h, X.pairs);
                    X.pairs = bigger; X.pairs_cap = want;
                }
            }
            if (!grow_err && m_max * (uint64_t)W > X.pairs_all_cap) grow_err = grow_buf(h, X.pairs_all, X.pairs_all_cap, want * (uint64_t)W);
            if (grow_err) { err = grow_err; err_msg = h->err; }
            if (W > 1) {
                const int late = grow_err ? GNDT_OK : err;
                if (!grow_err) err = GNDT_OK;
                rc = agree();
                if (!grow_err) err = late;
                if (rc) return rc;
            } else if (grow_err) return leave();
        }
    }
    if (m_max > X.pairs_cap || m_max * (uint64_t)W > X.pairs_all_cap) {     // (cannot happen: the capacities the ranks agreed on)
        h->err = "column-pair buffers smaller than the ranks agreed on";
        return GNDT_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_pairs_pad, dim3(grid_for(m_max)), dim3(256), 0, s, X.pairs, ncols, (uint32_t)m_max);
    if (err) {      // this rank's columns will be missing from everybody's order: a pair no map can hold makes the other ranks fail too
        my_msg_host[0] = kPoisonPair;
        HIP_TRY(h, hipMemcpyAsync(X.pairs, my_msg_host, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    }
    if ((rc = comm_all_gather(h, c, X.pairs, X.pairs_all, (size_t)m_max, 8, s))) return rc;
    const bool sliced = W > 1;
    if (err) {      // the others finish without this rank's columns — and notice, by its poison pair — once it has played its part
        if (sliced && (rc = global_rows_sliced(h, c, X.pairs_all, m_max, total_points, s, false))) return rc;
        return leave();
    }
    if (sliced) { if ((rc = global_rows_sliced(h, c, X.pairs_all, m_max, total_points, s))) return rc; }
    else if ((rc = global_rows_launch(h, X.pairs_all, m_max * (uint64_t)W, total_points, s))) return rc;
    if (!X.h_bad) HIP_TRY(h, hipHostMalloc(&X.h_bad, sizeof(uint32_t)));
    HIP_TRY(h, hipMemcpyAsync(X.h_totals, X.d_totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(X.h_bad, X.d_npairs + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    stamp(4);
    HIP_TRY(h, hipStreamSynchronize(s));
    const uint32_t bad = *X.h_bad;
    if (bad) { h->err = std::to_string(bad) + " column pair(s) with an index beyond total_points (a rank whose build failed late sends one on purpose)"; return GNDT_ERR_PEER; }
    if (h->h_cnt->err_key_range) {          // (a z level beyond the key range: found by the build; no collective follows)
        h->err = std::to_string(h->h_cnt->err_key_range) + " point(s) outside the key range";
        return GNDT_ERR_KEY_RANGE;
    }
    if (global_row_dev) *global_row_dev = X.global_row;
    X.h_totals[2] = 0;                                    // slopes of the whole map: every rank's share came with its column message
    for (int r = 0; r < W; ++r) X.h_totals[2] += X.h_colmsg[kColMsgWords * r + 2];
    X.owned_serial = h->result_serial; X.owned_world = (uint32_t)W; X.gathered = false;
    if (info) {
        float t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
        info->split_ms = t[0]; info->exchange_ms = t[1]; info->build_ms = t[2]; info->order_ms = t[3];
        info->owned_points = n_own; info->local_nodes = h->h_cnt->num_nodes; info->local_columns = ncols;
        info->global_nodes = X.h_totals[0]; info->global_columns = X.h_totals[1]; info->global_slopes = X.h_totals[2];
        info->bytes_sent = sent; info->bytes_received = received; info->ranks = (uint32_t)W;
    }
    return GNDT_OK;
}

// ---------------------------------------------------------------------------------------------
// the assembled map: the ranks' rows gathered and scattered by their global row
// ---------------------------------------------------------------------------------------------
int gndt_owned_pack_rows_device(gndt_handle* h, const uint32_t** rows_dev, uint64_t* n_rows, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!rows_dev || !n_rows) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    auto& X = h->exch;
    if ((rc = partition_resolve(h))) return rc;
    if (!h->results_valid || !X.owned_serial || X.owned_serial != h->result_serial) {
        h->err = "no owner-partitioned build on this handle (gndt_build_owned_device / gndt_owned_global_rows_device come first)";
        return GNDT_ERR_INVALID;
    }
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    const uint64_t nl = h->h_cnt->num_nodes;
    if ((rc = grow_buf(h, X.grec, X.grec_cap, std::max<uint64_t>(nl, 1) * kPackedRowWords))) return rc;
    if (nl) hipLaunchKernelGGL(k_rows_pack, dim3(grid_for(nl)), dim3(256), 0, s, h->out, (const uint32_t*)h->part.row_ncol, (const uint32_t*)X.global_row,
                               (uint32_t)nl, (uint32_t)nl, X.grec);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    *rows_dev = X.grec; *n_rows = nl;
    return GNDT_OK;
}

int gndt_adopt_rows_device(gndt_handle* h, const uint32_t* rows_dev, uint64_t n_rows, uint64_t total_nodes, uint64_t total_columns,
                           uint64_t total_slopes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if ((!rows_dev && n_rows) || total_nodes >= 0xFFFFFFFFull) { h->err = "bad argument"; return GNDT_ERR_INVALID; }
    auto& X = h->exch;
    if (rows_dev == X.grec) { h->err = "the rows to adopt must not be this handle's own pack buffer alone: pass everybody's rows"; return GNDT_ERR_INVALID; }
    hipStream_t s = stream_of(h, hip_stream);
    h->pending.active = false;
    { const int urc = use_stream(h, s); if (urc) return urc; }
    HIP_TRY(h, hipStreamSynchronize(s));                      // (the result arrays may be re-allocated: nothing may still read them)
    if ((rc = ensure_out(h, std::max<uint64_t>(total_nodes, 1)))) return rc;
    if ((rc = grow_buf(h, h->part.row_ncol, h->part.row_ncol_cap, std::max<uint64_t>(total_nodes, 1)))) return rc;
    if (!X.d_tally) HIP_TRY(h, hipMalloc(&X.d_tally, 2 * sizeof(uint32_t)));
    if (!X.h_tally) HIP_TRY(h, hipHostMalloc(&X.h_tally, 2 * sizeof(uint32_t)));
    HIP_TRY(h, hipMemsetAsync(X.d_tally, 0, 2 * sizeof(uint32_t), s));
    h->results_valid = false;
    if (n_rows) hipLaunchKernelGGL(k_rows_adopt, dim3(grid_for(n_rows)), dim3(256), 0, s, rows_dev, (uint64_t)n_rows, (uint32_t)total_nodes, h->out,
                                   h->part.row_ncol, X.d_tally);
    HIP_TRY(h, hipGetLastError());
    // the handle's counters now describe the adopted map (gndt_sync / gndt_export* / gndt_compute_cost read them)
    h->h_cnt->num_nodes = (uint32_t)total_nodes; h->h_cnt->num_columns = (uint32_t)total_columns; h->h_cnt->num_slopes = (uint32_t)total_slopes;
    h->h_cnt->err_key_range = 0; h->h_cnt->err_table_full = 0;
    HIP_TRY(h, hipMemcpyAsync(h->d_cnt, h->h_cnt, sizeof(Counters), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(X.h_tally, X.d_tally, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (X.h_tally[1] || X.h_tally[0] != total_nodes) {
        h->err = "adopted " + std::to_string(X.h_tally[0]) + " row(s) for a map of " + std::to_string(total_nodes) + " (" + std::to_string(X.h_tally[1]) +
                 " beyond it): the rows of some rank are missing or belong to another build";
        return GNDT_ERR_INVALID;
    }
    if (h->part.h_pc) { h->part.h_pc->stage_overflow = 0; h->part.h_pc->index_overflow = 0; }
    h->results_valid = true;
    ++h->result_serial;
    h->map_in_table = false; h->incr_ok = false;      // (the node table, if any, no longer describes the rows)
    h->last_stream = s;
    X.owned_serial = 0;
    return GNDT_OK;
}

int gndt_gather_owned_map_device(gndt_handle* h, gndt_comm* c, int32_t root, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!c || (!c->nccl && !c->threads)) { h->err = "no communicator"; return GNDT_ERR_INVALID; }
    const int W = c->world, me = c->rank;
    if (root >= W) { h->err = "root outside the communicator"; return GNDT_ERR_INVALID; }
    auto& X = h->exch;
    // Everything checked here is the same on all ranks after a successful gndt_build_owned_device with this communicator, so
    // either all ranks return or none does — ONCE per owned build: the root's handle holds the whole map afterwards (its owned
    // rows are gone), so every rank, root or not, refuses a second gather of the same build here, before any collective.
    // (Round 3 let the non-root ranks into the exchange of a second call while the old root had already returned: a hang.)
    if (!h->results_valid || !X.owned_serial || X.owned_serial != h->result_serial || X.owned_world != (uint32_t)W || X.gathered) {
        h->err = "gndt_gather_owned_map_device follows a successful gndt_build_owned_device on the same communicator, once per build";
        return GNDT_ERR_INVALID;
    }
    X.gathered = true;
    hipStream_t s = stream_of(h, hip_stream);
    { const int urc = use_stream(h, s); if (urc) return urc; }
    const uint64_t N = X.h_totals[0], K = X.h_totals[1], S = X.h_totals[2];
    std::vector<uint64_t> rows_of((size_t)W);
    uint64_t m_max = 1, sum = 0;
    for (int r = 0; r < W; ++r) { rows_of[r] = X.h_colmsg[kColMsgWords * r + 5]; m_max = std::max(m_max, rows_of[r]); sum += rows_of[r]; }
    const uint64_t nl = rows_of[me];
    if (sum != N || nl != h->h_cnt->num_nodes) { h->err = "the ranks' row counts do not add up to the map"; return GNDT_ERR_INVALID; }
    const bool all = root < 0;
    // this rank's rows, packed (padded to the longest list for the fixed-size all-gather)
    const uint64_t padded = all ? m_max : std::max<uint64_t>(nl, 1);
    // The buffers the collective below needs, then ONE agreement round (round 5): a rank that cannot allocate them used to return
    // here and leave the others waiting in the exchange.  (The gather runs once per assembled map, not per build: the round's
    // tiny all-gather is paid every time instead of publishing capacities first.)
    {
        int gerr = GNDT_OK;
        if (injected_failure(h, 3)) gerr = GNDT_ERR_NOMEM;
        if (!gerr) gerr = grow_buf(h, X.grec, X.grec_cap, padded * kPackedRowWords);
        if (!gerr && all) gerr = grow_buf(h, X.grec_all, X.grec_all_cap, m_max * (uint64_t)W * kPackedRowWords);
        if (!gerr && !all && me == root) gerr = grow_buf(h, X.grec_all, X.grec_all_cap, std::max<uint64_t>(N, 1) * kPackedRowWords);
        const std::string gmsg = h->err;
        if (W > 1) rc = agree_on(h, c, s, gerr, gmsg); else rc = gerr;
        if (rc) { X.gathered = false; return rc; }     // (all ranks leave here together: the same build may be gathered again)
    }
    hipLaunchKernelGGL(k_rows_pack, dim3(grid_for(padded)), dim3(256), 0, s, h->out, (const uint32_t*)h->part.row_ncol, (const uint32_t*)X.global_row,
                       (uint32_t)nl, (uint32_t)(all ? padded : nl), X.grec);
    HIP_TRY(h, hipGetLastError());
    uint64_t n_recv = 0;
    if (all) {
        n_recv = m_max * (uint64_t)W;
        if ((rc = grow_buf(h, X.grec_all, X.grec_all_cap, n_recv * kPackedRowWords))) return rc;
        if ((rc = comm_all_gather(h, c, X.grec, X.grec_all, (size_t)(m_max * kPackedRowWords), 4, s))) return rc;
    } else {
        // to ONE rank: the all-to-all primitive with a single receiver (ncclSend / ncclRecv in one group)
        std::vector<uint64_t> so((size_t)W, 0), sc((size_t)W, 0), ro((size_t)W, 0), rcnt((size_t)W, 0);
        if (me == root) {
            uint64_t off = 0;
            for (int r = 0; r < W; ++r) { ro[r] = off * kPackedRowWords * 4; rcnt[r] = r == me ? 0 : rows_of[r] * kPackedRowWords * 4; off += rows_of[r]; }
            n_recv = N;
            if ((rc = grow_buf(h, X.grec_all, X.grec_all_cap, std::max<uint64_t>(N, 1) * kPackedRowWords))) return rc;
            if (nl) HIP_TRY(h, hipMemcpyAsync(X.grec_all + ro[me] / 4, X.grec, nl * kPackedRowWords * 4, hipMemcpyDeviceToDevice, s));
        } else {
            sc[root] = nl * kPackedRowWords * 4;
        }
        if (W > 1 && (rc = comm_exchange(h, c, reinterpret_cast<const char*>(X.grec), so.data(), sc.data(), reinterpret_cast<char*>(X.grec_all), ro.data(),
                                          rcnt.data(), s))) return rc;
        if (me != root) { HIP_TRY(h, hipStreamSynchronize(s)); return GNDT_OK; }     // (this rank keeps the columns it owns)
    }
    return gndt_adopt_rows_device(h, X.grec_all, n_recv, N, K, S, hip_stream);
}

// One tiny round of every collective the sharded builds use, each verified against what it must produce: the first multi-rank run
// of a node is otherwise also the first test of libgndt's RCCL path (this pool has one-GPU boxes: rounds 1-3 never saw two RCCL
// ranks).  All ranks call it; every rank checks its own results.  bench.py --gpus N runs it before the first build.
int gndt_comm_selftest(gndt_handle* h, gndt_comm* c, gndt_comm_selftest_report* out, void* hip_stream) {
    if (!h || !c || !out) return GNDT_ERR_INVALID;
    if (!c->nccl && !c->threads) { h->err = "no communicator"; return GNDT_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    h->capturing = false;
    hipStream_t s = stream_of(h, hip_stream);
    const int W = c->world, me = c->rank;
    *out = gndt_comm_selftest_report{};
    out->ranks = (uint32_t)W;
    constexpr size_t kN = 256;                         // elements per rank and primitive
    const size_t words = (size_t)W * kN * 2 + 64;
    uint32_t *d_send = nullptr, *d_recv = nullptr;
    HIP_TRY(h, hipMalloc(&d_send, words * 4));
    HIP_TRY(h, hipMalloc(&d_recv, words * 4));
    std::vector<uint32_t> hs(words), hr(words);
    int rc = GNDT_OK;
    auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto up = [&](size_t nwords) { return hipMemcpyAsync(d_send, hs.data(), nwords * 4, hipMemcpyHostToDevice, s) == hipSuccess; };
    auto down = [&](size_t nwords) { return hipMemcpyAsync(hr.data(), d_recv, nwords * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess; };
    std::string bad;
    // 1. all-gather (u32)
    if (rc == GNDT_OK) {
        for (size_t i = 0; i < kN; ++i) hs[i] = (uint32_t)(me * 1000 + i);
        const double t0 = wall();
        if (!up(kN)) rc = GNDT_ERR_HIP;
        if (rc == GNDT_OK) rc = comm_all_gather(h, c, d_send, d_recv, kN, 4, s);
        if (rc == GNDT_OK && !down((size_t)W * kN)) rc = GNDT_ERR_HIP;
        out->all_gather_ms = (float)(wall() - t0);
        bool ok = rc == GNDT_OK;
        for (int q = 0; ok && q < W; ++q) for (size_t i = 0; i < kN; ++i) if (hr[(size_t)q * kN + i] != (uint32_t)(q * 1000 + i)) { ok = false; break; }
        if (ok) out->ok_mask |= 1u; else if (rc == GNDT_OK) bad += " all-gather";
    }
    // 2. all-to-all of runs (ncclSend / ncclRecv in one group): rank r sends 64 words r * 100 + q + i to every q
    if (rc == GNDT_OK) {
        constexpr size_t kRun = 64;
        std::vector<uint64_t> so((size_t)W), sc((size_t)W), ro((size_t)W), rcv((size_t)W);
        for (int q = 0; q < W; ++q) {
            so[q] = (uint64_t)q * kRun * 4; sc[q] = q == me ? 0 : kRun * 4; ro[q] = (uint64_t)q * kRun * 4; rcv[q] = q == me ? 0 : kRun * 4;
            for (size_t i = 0; i < kRun; ++i) hs[(size_t)q * kRun + i] = (uint32_t)(me * 100 + q + i * 7);
        }
        const double t0 = wall();
        if (!up((size_t)W * kRun)) rc = GNDT_ERR_HIP;
        if (rc == GNDT_OK && hipMemsetAsync(d_recv, 0, (size_t)W * kRun * 4, s) != hipSuccess) rc = GNDT_ERR_HIP;
        if (rc == GNDT_OK && W > 1) rc = comm_exchange(h, c, reinterpret_cast<const char*>(d_send), so.data(), sc.data(), reinterpret_cast<char*>(d_recv), ro.data(), rcv.data(), s);
        if (rc == GNDT_OK && !down((size_t)W * kRun)) rc = GNDT_ERR_HIP;
        out->exchange_ms = (float)(wall() - t0);
        bool ok = rc == GNDT_OK;
        for (int q = 0; ok && q < W; ++q) { if (q == me) continue; for (size_t i = 0; i < kRun; ++i) if (hr[(size_t)q * kRun + i] != (uint32_t)(q * 100 + me + i * 7)) { ok = false; break; } }
        if (ok) out->ok_mask |= 2u; else if (rc == GNDT_OK) bad += " send/recv";
    }
    // 3. reduce-scatter (u32 sum)
    if (rc == GNDT_OK) {
        for (int q = 0; q < W; ++q) for (size_t i = 0; i < kN; ++i) hs[(size_t)q * kN + i] = (uint32_t)(me + 1 + q * 3 + i);
        const double t0 = wall();
        if (!up((size_t)W * kN)) rc = GNDT_ERR_HIP;
        if (rc == GNDT_OK) rc = comm_reduce_scatter_u32(h, c, d_send, d_recv, kN, s);
        if (rc == GNDT_OK && !down(kN)) rc = GNDT_ERR_HIP;
        out->reduce_scatter_ms = (float)(wall() - t0);
        bool ok = rc == GNDT_OK;
        const uint32_t ranks_sum = (uint32_t)(W * (W + 1) / 2);
        for (size_t i = 0; ok && i < kN; ++i) if (hr[i] != ranks_sum + (uint32_t)W * (uint32_t)(me * 3 + i)) ok = false;
        if (ok) out->ok_mask |= 4u; else if (rc == GNDT_OK) bad += " reduce-scatter";
    }
    // 4. all-reduce: f64 sum and u32 min, in place
    if (rc == GNDT_OK) {
        double* hd = reinterpret_cast<double*>(hs.data());
        for (size_t i = 0; i < kN; ++i) hd[i] = 0.5 * (me + 1) + (double)i;
        const double t0 = wall();
        if (!up(2 * kN)) rc = GNDT_ERR_HIP;
        if (rc == GNDT_OK) rc = comm_all_reduce(h, c, d_send, kN, true, s);
        if (rc == GNDT_OK && (hipMemcpyAsync(hr.data(), d_send, 2 * kN * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = GNDT_ERR_HIP;
        bool ok = rc == GNDT_OK;
        const double* rd = reinterpret_cast<const double*>(hr.data());
        for (size_t i = 0; ok && i < kN; ++i) if (rd[i] != 0.5 * (W * (W + 1) / 2) + (double)W * (double)i) ok = false;
        if (rc == GNDT_OK) {
            for (size_t i = 0; i < kN; ++i) hs[i] = (uint32_t)(1000 - me + i);
            if (!up(kN)) rc = GNDT_ERR_HIP;
            if (rc == GNDT_OK) rc = comm_all_reduce(h, c, d_send, kN, false, s);
            if (rc == GNDT_OK && (hipMemcpyAsync(hr.data(), d_send, kN * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = GNDT_ERR_HIP;
            for (size_t i = 0; ok && rc == GNDT_OK && i < kN; ++i) if (hr[i] != (uint32_t)(1000 - (W - 1) + i)) ok = false;
        }
        out->all_reduce_ms = (float)(wall() - t0);
        if (ok && rc == GNDT_OK) out->ok_mask |= 8u; else if (rc == GNDT_OK) bad += " all-reduce";
    }
    (void)hipFree(d_send);
    (void)hipFree(d_recv);
    if (rc) return rc;
    if (out->ok_mask != 15u) { h->err = "communicator self-test: wrong result from" + bad + " on rank " + std::to_string(me); return GNDT_ERR_PEER; }
    return GNDT_OK;
}

}  // extern "C"
