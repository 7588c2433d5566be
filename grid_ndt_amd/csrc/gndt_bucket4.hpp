// gndt_bucket4.hpp — k_bucket_owner: one workgroup per bucket; every node is OWNED by one thread, which keeps the node's
// statistics in registers from the first record to the staging row.
//
// Measured on MI355X (tools/lds_atomic_bench.hip, profiles/r02_lds_atomic_rates.txt): an LDS instruction whose 64 lanes go
// to random addresses runs at ~6 lanes per clock and CU (bank conflicts), a random ds_add_f64 at 2.5.  That, not
// instruction issue, bounds a bucket kernel that adds every point into an LDS table (k_bucket_direct: 11 atomics per
// point, 182 us for the 10 M-point bench scene, 83 % of the LDS bound) and is what the run flushes of k_bucket_build2
// (5 atomics per point) pay as well.  Here a point costs FOUR random LDS instructions and no floating-point atomic:
//
//   per chunk of up to CH records (one chunk for nearly every bucket)
//     classify  record -> key (divide-free exact index) -> slot of the LDS key table (probe, CAS on a miss; a new node
//               gets the next node number = the thread that owns it) -> arrival rank in the slot (one returning u32 atomic)
//     scan      exclusive prefix of the slot counts
//     scatter   record -> slot-sorted image of the chunk in LDS (one 16-byte store)
//     reduce    the TWO lanes that own a node walk its run (even / odd records, 16-byte loads) and add it to fp64 sums in
//               their REGISTERS; the pair's halves are added once, after the last chunk
//   then
//     the owners publish count / first-seen / mean-z, join their column's list, walk it for the slope label and the
//     index in column, and write the 96-B staging row straight from their registers: the sums never live in LDS.
//
// Runs longer than kLongRun records (dense nodes, the reference's zero padding) are summed by a whole wave each and
// handed to the owner through a small LDS scratch.  Semantics are those of the other bucket kernels (same gndt_math.hpp
// arithmetic, same order-free label rule); tests run every one of them against the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_bucket3.hpp"

namespace gndt {

constexpr int kLongRun = 96;        // records of one node in one chunk beyond which a wave sums the run
constexpr int kLongMax = 32;        // such runs per chunk the scratch holds (CH / kLongRun at most)

template <int H, int CH>
struct BucketLds4 {
    unsigned long long key[H];
    uint32_t start[H + 2];      // per chunk: arrival counters, then exclusive start offsets; start[H] = records in the chunk
    uint16_t list[H];           // the bucket's nodes in discovery order: node i is owned by thread i
    union {
        float4 img[CH];         // the chunk's records grouped by slot
        struct {                // after the last chunk
            uint32_t first[H], cnt[H];
            float mean_z[H];
            uint32_t chead[H], next[H];
            uint16_t cslot[H];
        } fin;
    } u;
    double long_sum[kLongMax][9];
    uint32_t long_cnt[kLongMax], long_first[kLongMax];
    uint16_t long_slot[kLongMax];
    uint32_t wave_tot[16];
    uint32_t n_nodes, n_cols, n_slopes, n_long, stage_base, overflow, err_range, pad;
};

// find-or-insert that also numbers a new node (discovery order) and records its slot in `list`
template <int H, typename Lds>
__device__ __forceinline__ uint32_t owner_find_or_insert(Lds& L, uint32_t start, uint64_t key) {
    uint32_t slot = start & (H - 1);
    for (int probe = 0; probe < H; ++probe) {
        const unsigned long long k = L.key[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) {
            const unsigned long long old = atomicCAS(&L.key[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) {
                const uint32_t id = atomicAdd(&L.n_nodes, 1u);
                if (id < (uint32_t)H) L.list[id] = (uint16_t)slot;
                return slot;
            }
            if (old == key) return slot;
        }
        slot = (slot + 1) & (H - 1);
    }
    return H;
}

struct OwnerAcc {               // one node's additive statistics, in the owning thread's registers
    double s0, s1, s2, s3, s4, s5, s6, s7, s8;
    uint32_t cnt, first;
};

__device__ __forceinline__ void owner_add(OwnerAcc& a, const float4& r, double c0, double c1, double c2) {
    const uint32_t iw = __float_as_uint(r.w);
    const double v0 = (double)r.x - c0, v1 = (double)r.y - c1, v2 = (double)r.z - c2;
    const uint32_t wn = record_weight(iw);                // 64 or 512 identical points in one record (exact: powers of two)
    const double wf = (double)wn;
    const double w0 = wf * v0, w1 = wf * v1, w2 = wf * v2;
    a.s0 += w0; a.s1 += w1; a.s2 += w2;
    // fused multiply-add on purpose: one rounding per term (the sums are order-free anyway)
    a.s3 = fma(w0, v0, a.s3); a.s4 = fma(w0, v1, a.s4); a.s5 = fma(w0, v2, a.s5);
    a.s6 = fma(w1, v1, a.s6); a.s7 = fma(w1, v2, a.s7); a.s8 = fma(w2, v2, a.s8);
    a.cnt += wn;
    a.first = min(a.first, record_index(iw));
}

__device__ __forceinline__ double wave_sum_all(double v) {   // every lane gets the sum
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int T, int H, int CH, bool STATS>
__device__ __forceinline__ void bucket_owner_one(BucketLds4<H, CH>& L, const uint32_t bucket, const float4* __restrict__ recs,
                                                 const uint32_t lo, const uint32_t hi, const GridParams& P,
                                                 StageRow* __restrict__ stage, uint32_t stage_cap,
                                                 uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                 const ColumnOrder& O, Counters* __restrict__ cnt,
                                                 PartCounters* __restrict__ pc, unsigned long long* __restrict__ dbg,
                                                 const StatsOut& so) {
    static_assert(T == H, "one slot per thread in the scan; a node per thread in the per-node phases");
    static_assert(CH % T == 0 && CH <= 65536, "chunk");
    static_assert(CH / (kLongRun + 1) < kLongMax, "the long-run scratch holds every run a chunk can contain");
    constexpr int PER = CH / T;
    constexpr int kFill = T / 2;               // two lanes own a node
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#define GNDT_STAMP4(k) do { if (dbg && tid == 0) dbg[(size_t)bucket * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    GNDT_STAMP4(0);
    // the first chunk's records are requested before the tables are cleared: their HBM round trip (~2 us) overlaps the set-up
    float4 pre[PER];
    if (lo < hi) {
        const uint32_t n0 = min((uint32_t)CH, hi - lo);
#pragma unroll
        for (int j = 0; j < PER; ++j) pre[j] = recs[lo + min((uint32_t)(j * T + tid), n0 - 1u)];
    }
    L.key[tid] = kEmptyKey;
    L.start[tid] = 0;
    if (tid == 0) { L.start[H] = 0; L.n_nodes = 0; L.n_cols = 0; L.n_slopes = 0; L.n_long = 0; L.stage_base = 0; L.overflow = 0; L.err_range = 0; }
    OwnerAcc acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0u, 0xFFFFFFFFu};
    const double hx = 0.5 * (double)P.grid_len, hz = 0.5 * (double)P.z_len;
    const double ox = (double)P.ox, oy = (double)P.oy, oz = (double)P.oz;
    __syncthreads();
    GNDT_STAMP4(1);

    unsigned long long acc_t[4] = {0, 0, 0, 0}, t_prev = 0;     // diagnostic: classify / scan / scatter / reduce, all chunks
#define GNDT_LAP4(k) do { if (dbg && tid == 0) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); acc_t[k] += t_now - t_prev; t_prev = t_now; } } while (0)
    if (dbg && tid == 0) t_prev = __builtin_amdgcn_s_memtime();
    for (uint32_t cbeg = lo; cbeg < hi; cbeg += CH) {
        const uint32_t nchunk = min((uint32_t)CH, hi - cbeg);
        // ---- classify: the records are NOT kept in registers across the barriers (the scatter re-reads them, L2-hot), so that
        //      three workgroups fit a CU's register file ----
        uint32_t tag[PER];                      // slot << 16 | arrival rank
        {
            // Staged over the PER records of a thread so that their latencies overlap: all loads, all keys and first
            // probes, then the (rarer) misses, then all arrival-rank atomics — one wait per stage, not per record.
            float4 rec[PER];
            if (cbeg == lo) {                   // (uniform)
#pragma unroll
                for (int j = 0; j < PER; ++j) rec[j] = pre[j];
            } else {
#pragma unroll
                for (int j = 0; j < PER; ++j) rec[j] = recs[cbeg + min((uint32_t)(j * T + tid), nchunk - 1u)];   // (out-of-range lanes re-read the last record)
            }
            unsigned long long pkey[PER], k0[PER];
            uint32_t slot[PER];
            bool use[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const PointKey k = point_key_fast(rec[j].x, rec[j].y, rec[j].z, P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
                pkey[j] = pack_key(k.sx, k.sy, k.sz);
                slot[j] = node_slot3(column_hash(k.sx, k.sy), k.sz) & (uint32_t)(H - 1);
                const bool live = (uint32_t)(j * T + tid) < nchunk;
                if (live && !k.ok) atomicAdd(&L.err_range, 1u);          // |nz| beyond the key range (x, y were checked by the partition)
                use[j] = live && k.ok;
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) k0[j] = L.key[slot[j]];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (use[j] && k0[j] != pkey[j]) {                        // first probe missed: new node or a collision
                    slot[j] = owner_find_or_insert<H>(L, slot[j], pkey[j]);
                    if (slot[j] >= (uint32_t)H) { L.overflow = 1; use[j] = false; }
                }
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                uint32_t r = 0xFFFFu;
                if (use[j]) r = atomicAdd(&L.start[slot[j]], 1u);
                tag[j] = use[j] ? ((slot[j] << 16) | r) : 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        GNDT_LAP4(0);
        if (L.overflow || L.n_nodes > (uint32_t)kFill) {     // uniform
            if (tid == 0) atomicAdd(&pc->lds_overflow, 1u);
            return;
        }
        // ---- scan: exclusive prefix of the slot counts, in slot order (so slot s owns [start[s], start[s + 1])) ----
        {
            const uint32_t c = L.start[tid];
            uint32_t incl = c;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) L.wave_tot[wave] = incl;
            __syncthreads();
            uint32_t run = incl - c;
            for (int w = 0; w < wave; ++w) run += L.wave_tot[w];
            L.start[tid] = run;
            if (tid == T - 1) L.start[H] = run + c;
            if (c > (uint32_t)kLongRun) {                    // a long run: summed by a wave (below), not by its owner alone
                const uint32_t q = atomicAdd(&L.n_long, 1u);
                if (q < (uint32_t)kLongMax) L.long_slot[q] = (uint16_t)tid;
            }
        }
        __syncthreads();
        GNDT_LAP4(1);
        // ---- scatter: the chunk grouped by slot ----
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (tag[j] != 0xFFFFFFFFu) L.u.img[L.start[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = recs[cbeg + (uint32_t)(j * T + tid)];
        __syncthreads();
        GNDT_LAP4(2);
        // ---- reduce: every owner walks the run of its node ----
        const uint32_t M = L.n_nodes;
        const uint32_t n_long = min(L.n_long, (uint32_t)kLongMax);
        uint32_t my_slot = 0xFFFFFFFFu;
        if ((uint32_t)(tid >> 1) < M) {                    // lanes 2i, 2i+1 own node i: even and odd records of its run
            my_slot = L.list[tid >> 1];
            const uint32_t b = L.start[my_slot], e = L.start[my_slot + 1];
            if (e > b && e - b <= (uint32_t)kLongRun) {
                int sx, sy, sz;
                unpack_key(L.key[my_slot], sx, sy, sz);
                const double c0 = fma((double)(2 * sx - (sx > 0 ? 1 : -1)), hx, ox), c1 = fma((double)(2 * sy - (sy > 0 ? 1 : -1)), hx, oy),
                             c2 = fma((double)(2 * sz - (sz > 0 ? 1 : -1)), hz, oz);
                uint32_t p = b + ((uint32_t)tid & 1u);
                if (p < e) {
                    float4 r = L.u.img[p];
                    for (; p < e; p += 2) {
                        const float4 nx = L.u.img[min(p + 2u, e - 1u)];   // next record in flight while this one is added
                        owner_add(acc, r, c0, c1, c2);
                        r = nx;
                    }
                }
            }
        }
        if (n_long) {                                      // (uniform) long runs: one wave each, lanes stride over the run
            for (uint32_t q = wave; q < n_long; q += T / 64) {
                const uint32_t s = L.long_slot[q];
                const uint32_t b = L.start[s], e = L.start[s + 1];
                int sx, sy, sz;
                unpack_key(L.key[s], sx, sy, sz);
                const double c0 = fma((double)(2 * sx - (sx > 0 ? 1 : -1)), hx, ox), c1 = fma((double)(2 * sy - (sy > 0 ? 1 : -1)), hx, oy),
                             c2 = fma((double)(2 * sz - (sz > 0 ? 1 : -1)), hz, oz);
                OwnerAcc a{0, 0, 0, 0, 0, 0, 0, 0, 0, 0u, 0xFFFFFFFFu};
                for (uint32_t p = b + lane; p < e; p += 64) owner_add(a, L.u.img[p], c0, c1, c2);
                a.s0 = wave_sum_all(a.s0); a.s1 = wave_sum_all(a.s1); a.s2 = wave_sum_all(a.s2);
                a.s3 = wave_sum_all(a.s3); a.s4 = wave_sum_all(a.s4); a.s5 = wave_sum_all(a.s5);
                a.s6 = wave_sum_all(a.s6); a.s7 = wave_sum_all(a.s7); a.s8 = wave_sum_all(a.s8);
                uint32_t c = a.cnt, f = a.first;
                for (int off = 32; off > 0; off >>= 1) { c += (uint32_t)__shfl_xor((int)c, off, 64); f = min(f, (uint32_t)__shfl_xor((int)f, off, 64)); }
                if (lane == 0) {
                    L.long_sum[q][0] = a.s0; L.long_sum[q][1] = a.s1; L.long_sum[q][2] = a.s2; L.long_sum[q][3] = a.s3; L.long_sum[q][4] = a.s4;
                    L.long_sum[q][5] = a.s5; L.long_sum[q][6] = a.s6; L.long_sum[q][7] = a.s7; L.long_sum[q][8] = a.s8;
                    L.long_cnt[q] = c; L.long_first[q] = f;
                }
            }
            __syncthreads();
            if (my_slot != 0xFFFFFFFFu && (tid & 1) == 0) {
                for (uint32_t q = 0; q < n_long; ++q)
                    if ((uint32_t)L.long_slot[q] == my_slot) {
                        acc.s0 += L.long_sum[q][0]; acc.s1 += L.long_sum[q][1]; acc.s2 += L.long_sum[q][2]; acc.s3 += L.long_sum[q][3];
                        acc.s4 += L.long_sum[q][4]; acc.s5 += L.long_sum[q][5]; acc.s6 += L.long_sum[q][6]; acc.s7 += L.long_sum[q][7];
                        acc.s8 += L.long_sum[q][8];
                        acc.cnt += L.long_cnt[q]; acc.first = min(acc.first, L.long_first[q]);
                    }
            }
        }
        __syncthreads();                                   // the image and the counters are reused by the next chunk
        GNDT_LAP4(3);
        L.start[tid] = 0;
        if (tid == 0) { L.start[H] = 0; L.n_long = 0; }
        if (cbeg + CH < hi) __syncthreads();
    }
#undef GNDT_LAP4
    if (dbg && tid == 0) for (int q = 0; q < 4; ++q) dbg[(size_t)bucket * 16 + 8 + q] = acc_t[q];
    GNDT_STAMP4(2);

    // ---- the pair's halves -> the node's statistics, in both lanes (neighbouring lanes: a DPP move per word) ----
    acc.s0 += __shfl_xor(acc.s0, 1, 64); acc.s1 += __shfl_xor(acc.s1, 1, 64); acc.s2 += __shfl_xor(acc.s2, 1, 64);
    acc.s3 += __shfl_xor(acc.s3, 1, 64); acc.s4 += __shfl_xor(acc.s4, 1, 64); acc.s5 += __shfl_xor(acc.s5, 1, 64);
    acc.s6 += __shfl_xor(acc.s6, 1, 64); acc.s7 += __shfl_xor(acc.s7, 1, 64); acc.s8 += __shfl_xor(acc.s8, 1, 64);
    acc.cnt += (uint32_t)__shfl_xor((int)acc.cnt, 1, 64);
    acc.first = min(acc.first, (uint32_t)__shfl_xor((int)acc.first, 1, 64));

    // ---- the owners publish what the other nodes of their column need (the image is dead: its LDS holds these arrays).
    //      From here on the EVEN lane of a pair does the column work and the row's head, the ODD lane the moments. ----
    const uint32_t M = L.n_nodes;
    uint32_t stage_base_reg = 0;
    if (tid == T - 1) stage_base_reg = atomicAdd(&cnt->num_nodes, M);    // (its round trip hides behind the per-node phases)
    if (tid == 0 && L.err_range) atomicAdd(&cnt->err_key_range, L.err_range);
    const uint32_t node = (uint32_t)tid >> 1;
    const bool mine = node < M, even = (tid & 1) == 0;
    const uint32_t s = mine ? (uint32_t)L.list[node] : 0u;
    uint64_t key = 0;
    int sx = 0, sy = 0, sz = 0;
    float mz = 0.f;
    L.u.fin.chead[tid] = kNoNode;
    const bool has = acc.cnt >= (uint32_t)P.min_points;
    if (mine) {
        key = L.key[s];
        unpack_key(key, sx, sy, sz);
        if (even) {
            if (has) mz = node_mean_z(acc.cnt, acc.s2, axis_centre(sz, P.oz, P.z_len));
            L.u.fin.first[s] = acc.first; L.u.fin.cnt[s] = acc.cnt; L.u.fin.mean_z[s] = mz;
        }
    }
    if (tid == T - 1) L.stage_base = stage_base_reg;
    __syncthreads();
    const uint32_t sbase = L.stage_base;
    if (sbase + M > stage_cap) {                   // uniform
        if (tid == 0) atomicAdd(&pc->stage_overflow, M);
        return;
    }
    if constexpr (STATS) {
        if (mine && even) {
            const uint32_t dst = sbase + node;
            so.key[dst] = key;
            double* o = so.sums + 9 * (size_t)dst;
            o[0] = acc.s0; o[1] = acc.s1; o[2] = acc.s2; o[3] = acc.s3; o[4] = acc.s4; o[5] = acc.s5; o[6] = acc.s6; o[7] = acc.s7; o[8] = acc.s8;
            so.count[dst] = acc.cnt;
            so.first[dst] = acc.first;
        }
        return;
    }
    StageRow* __restrict__ const row = stage + sbase + node;
    // the ODD lane: mean + fp64 scatter from its registers, straight into the row (runs beside the column work of the even lane)
    if (mine && !even) {
        float mean[3] = {0.f, 0.f, 0.f};
        double S[6] = {0, 0, 0, 0, 0, 0};
        if (has) {
            const double sums[9] = {acc.s0, acc.s1, acc.s2, acc.s3, acc.s4, acc.s5, acc.s6, acc.s7, acc.s8};
            const double c[3] = {axis_centre(sx, P.ox, P.grid_len), axis_centre(sy, P.oy, P.grid_len), axis_centre(sz, P.oz, P.z_len)};
            node_moments(acc.cnt, sums, c, mean, S);
        }
        for (int q = 0; q < 3; ++q) row->mean[q] = mean[q];
        for (int q = 0; q < 6; ++q) row->scatter[q] = S[q];
    }
    // ---- columns: every node joins the list of its column (the column's key is the key of any node on its list) ----
    if (mine && even) {
        const uint64_t ck = column_key(key);
        uint32_t c = ((column_hash(sx, sy) * 0x85EBCA77u) >> 12) & (uint32_t)(H - 1);
        for (int probe = 0; probe < H; ++probe) {            // (terminates: the table has more slots than nodes)
            uint32_t head = L.u.fin.chead[c];
            if (head == kNoNode) {
                head = atomicCAS(&L.u.fin.chead[c], kNoNode, s);
                if (head == kNoNode) { L.u.fin.next[s] = kNoNode; atomicAdd(&L.n_cols, 1u); break; }          // first node of a new column
            }
            if (column_key(L.key[head]) == ck) { L.u.fin.next[s] = atomicExch(&L.u.fin.chead[c], s); break; }   // push in front
            c = (c + 1) & (uint32_t)(H - 1);
        }
        L.u.fin.cslot[s] = (uint16_t)c;
    }
    __syncthreads();
    GNDT_STAMP4(3);
    if (tid == 0) atomicAdd(&cnt->num_columns, L.n_cols);
    // ---- rows: slope label (OcNode::isSlope, map2D.h:66-108), index in column, column size and first-seen index by walking
    //      the column's short list -> the head of the staging row ----
    uint32_t got_slope = 0;
    if (mine && even) {
        const int za = level_above(sz), zb = level_below(sz);
        uint32_t icol = 0, ncol = 0, cf = 0xFFFFFFFFu;
        bool up = false, down = false;
        for (uint32_t t = L.u.fin.chead[L.u.fin.cslot[s]]; t != kNoNode; t = L.u.fin.next[t]) {
            const uint32_t tf = L.u.fin.first[t];
            ++ncol;
            cf = min(cf, tf);
            if (t == s) continue;
            icol += (tf < acc.first) ? 1u : 0u;
            const int tz = (int)(L.key[t] & 0x3FFFFFu) - (1 << 21);
            if (tz == za || tz == zb) {
                const bool visited = tf < acc.first && L.u.fin.cnt[t] >= (uint32_t)P.min_points;
                const float oz2 = visited ? L.u.fin.mean_z[t] : 0.f;
                const bool far = fabsf(oz2 - mz) > P.slope_interval;
                if (tz == za) up = up || far; else down = down || far;
            }
        }
        uint32_t fl = has ? 1u : 0u;
        if (fl) {
            bool slope = true;
            if (P.demand == 0) slope = !up; else down = false;
            if (slope) { fl |= 2u; if (down) fl |= 4u; got_slope = 1u; }
        }
        row->sx = sx; row->sy = sy; row->sz = sz;
        row->count = acc.cnt; row->first = acc.first; row->flags = fl;
        row->col_first = cf; row->idx_in_col = icol; row->ncol = ncol;
        const uint32_t dst = sbase + node;
        ord_cf[dst] = cf;
        ord_idx[dst] = icol;
        if (icol == 0) note_column(O, cf, ncol);
    }
    // counters: aggregated per wave, then in LDS: ONE memory-side atomic per bucket and counter
    const uint32_t wave_slopes = (uint32_t)__popcll(__ballot(got_slope != 0u));
    if (lane == 0 && wave_slopes) atomicAdd(&L.n_slopes, wave_slopes);
    __syncthreads();
    if (tid == 0 && L.n_slopes) atomicAdd(&cnt->num_slopes, L.n_slopes);
    GNDT_STAMP4(4);
#undef GNDT_STAMP4
}

template <int T, int H, int CH, bool STATS = false>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bucket_owner(
    const float4* __restrict__ recs, const uint32_t* __restrict__ range_lo, const uint32_t* __restrict__ range_hi,
    uint32_t num_buckets, GridParams P, StageRow* __restrict__ stage, uint32_t stage_cap, uint32_t* __restrict__ ord_cf,
    uint32_t* __restrict__ ord_idx, ColumnOrder O, Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
    unsigned long long* __restrict__ dbg, StatsOut so) {
    __shared__ BucketLds4<H, CH> L;
    for (uint32_t bucket = blockIdx.x; bucket < num_buckets; bucket += gridDim.x) {
        bucket_owner_one<T, H, CH, STATS>(L, bucket, recs, range_lo[bucket], range_hi[bucket], P, stage, stage_cap, ord_cf, ord_idx, O,
                                          cnt, pc, dbg, so);
        __syncthreads();        // the LDS tables are re-initialised by the next bucket
    }
}

}  // namespace gndt
