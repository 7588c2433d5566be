// gndt_exchange.hpp — the statistics exchange that builds ONE global map from a cloud sharded over the GPUs of a node
// (BASELINE.json configs[2]), with RCCL called from C++ behind the C ABI: a C++ / ROS host needs no Python
// and no torch.distributed to shard a cloud.  RCCL is resolved at run time (dlopen): single-GPU users of libgndt do not
// need it installed.
//
// Per-node statistics in cell-local coordinates are additive over any partition of the points and the first-seen index
// combines with min, so one round builds the global map:
//   all-gather   every rank's occupied keys (8 B each)  -> sort + unique on the device = the canonical node order
//   all-reduce   SUM over a PACKED [C x 10] fp64 buffer (9 sums + the count, exact in fp64) scattered into that order
//   all-reduce   MIN over [C] u32 first-seen indices
//   every rank   gndt_finalize_stats_device on the reduced statistics (already sorted by key: a column's nodes adjacent)
// Only occupied nodes travel (84 B each), never points.  xGMI is point to point (7 links per GPU): the ring all-reduce
// moves 2 (N-1)/N x 84 B x C per GPU, the dominant cost at scale.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rccl/rccl.h>          // types and prototypes only: the functions come from dlopen

#include "gndt_math.hpp"

namespace gndt {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllGather && AllReduce && GetErrorString; }
};

// the RCCL the process already has (PyTorch bundles its own librccl.so) or the system one
inline const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            a.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (a.lib) break;
        }
        if (!a.lib)
            for (const char* nm : names) {
                a.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (a.lib) break;
            }
        if (a.lib) {
            a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
            a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
            a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
            a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
            a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        }
        return a;
    }();
    return api;
}

// ---- device side of the exchange ----
constexpr int kExWidth = 10;       // packed fp64 words per node: 9 sums + count

// position of every local node in the canonical (sorted, unique) key list, and its statistics scattered there
static __global__ void __launch_bounds__(256) k_exchange_scatter(const uint64_t* __restrict__ key, const double* __restrict__ sums,
                                                                 const uint32_t* __restrict__ count, const uint32_t* __restrict__ first,
                                                                 uint32_t m, const uint64_t* __restrict__ canon, uint32_t C,
                                                                 double* __restrict__ packed, uint32_t* __restrict__ pfirst,
                                                                 uint32_t* __restrict__ missing) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint64_t k = key[i];
        uint32_t lo = 0, hi = C;                          // lower_bound
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (canon[mid] < k) lo = mid + 1; else hi = mid; }
        if (lo >= C || canon[lo] != k) { atomicAdd(missing, 1u); continue; }      // (cannot happen: the list is the union)
        double* o = packed + (size_t)lo * kExWidth;
#pragma unroll
        for (int j = 0; j < 9; ++j) o[j] = sums[9 * (size_t)i + j];
        o[9] = (double)count[i];
        pfirst[lo] = first[i];
    }
}

static __global__ void __launch_bounds__(256) k_exchange_init(double* __restrict__ packed, uint32_t* __restrict__ pfirst, uint32_t C) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)C * kExWidth; i += (uint64_t)gridDim.x * blockDim.x) packed[i] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) pfirst[i] = 0xFFFFFFFFu;
}

static __global__ void __launch_bounds__(256) k_exchange_unpack(const double* __restrict__ packed, uint32_t C, double* __restrict__ sums,
                                                                uint32_t* __restrict__ count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) {
        const double* p = packed + (size_t)i * kExWidth;
#pragma unroll
        for (int j = 0; j < 9; ++j) sums[9 * (size_t)i + j] = p[j];
        count[i] = (uint32_t)p[9];
    }
}

// keys padded to a common length with kEmptyKey (sorts last); after sort + unique the pad is the last entry, if present
static __global__ void k_exchange_pad(uint64_t* __restrict__ buf, uint32_t have, uint32_t padded) {
    for (uint32_t i = have + blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += gridDim.x * blockDim.x) buf[i] = kEmptyKey;
}

}  // namespace gndt
