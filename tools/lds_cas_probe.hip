// Diagnostic: cost of the LDS read-modify-write flavours a hash-table insert can use, gfx950.  One 1024-thread workgroup per CU;
// every lane hits a pseudo-random slot of 1024 (or ONE shared slot).  Prints clocks per wave-instruction of one wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_cas_probe.hip -o build/lds_cas_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int T = 1024, SLOTS = 1024, ITER = 256;
// OP: 0 cmpst_rtn_b64, 1 cmpst_rtn_b32, 2 min_rtn_u64, 3 wrxchg_rtn_b64, 4 add_rtn_u32, 5 add_u32 (no return), 6 read_b64, 7 or_rtn_b32, 8 add_rtn_u64
template <int OP, bool SAME>
__global__ void __launch_bounds__(T) k(unsigned long long* cycles, unsigned long long* sink) {
    __shared__ unsigned long long sl[SLOTS];
    __shared__ uint32_t su[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += T) { sl[i] = ~0ull; su[i] = 0; }
    __syncthreads();
    uint32_t r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long acc = 0;
    unsigned long long c0 = clock64();
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        r = r * 1664525u + 1013904223u;
        const uint32_t s = SAME ? 7u : (r >> 12) & (SLOTS - 1);
        if (OP == 0) acc += atomicCAS(&sl[s], ~0ull, (unsigned long long)r);
        if (OP == 1) acc += atomicCAS(&su[s], 0u, r);
        if (OP == 2) acc += atomicMin(&sl[s], (unsigned long long)r << 20);
        if (OP == 3) acc += atomicExch(&sl[s], (unsigned long long)r);
        if (OP == 4) acc += atomicAdd(&su[s], 1u);
        if (OP == 5) atomicAdd(&su[s], 1u);
        if (OP == 6) acc += sl[s];
        if (OP == 7) acc += atomicOr(&su[s], r);
        if (OP == 8) acc += atomicAdd(&sl[s], 1ull);
    }
    __syncthreads();
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = c1 - c0;
    if (acc == 0x12345678u) sink[0] = acc;
}
static unsigned long long *d_c, *d_s;
template <int OP, bool SAME> void run(const char* name) {
    const int blocks = 256;
    hipLaunchKernelGGL((k<OP, SAME>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipLaunchKernelGGL((k<OP, SAME>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= blocks;
    printf("%-18s %-12s %8.2f clk per wave-instruction\n", name, SAME ? "ONE slot" : "random slot", m / ITER / (T / 64));
}
#define BOTH(OP, N) run<OP, false>(N); run<OP, true>(N);
int main() {
    hipMalloc(&d_c, 256 * 8); hipMalloc(&d_s, 64);
    BOTH(0, "cmpst_rtn_b64") BOTH(1, "cmpst_rtn_b32") BOTH(2, "min_rtn_u64") BOTH(3, "wrxchg_rtn_b64") BOTH(4, "add_rtn_u32") BOTH(5, "add_u32")
    BOTH(6, "read_b64") BOTH(7, "or_rtn_b32") BOTH(8, "add_rtn_u64")
    return 0;
}
