// Diagnostic: which lanes of a wave64 LDS atomic conflict on gfx950?  One 1024-thread workgroup per CU; every lane adds to
// slot = column(lane) + row * ROWSTRIDE with a pseudo-random row per iteration (an LCG: the loop stays LDS-bound).
// Prints clocks per wave-instruction of one wave (16 waves per CU issue concurrently).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_bank_probe.hip -o build/lds_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int T = 1024, SLOTS = 2048, ITER = 512;
// COLS: 0 = lane (64 distinct), 1 = lane & 31, 2 = lane & 15, 3 = (lane * 5) & 31, 4 = fully random slot, 5 = lane & 7
template <int MODE, int COLS, int ROWS, int ROWSTRIDE>
__global__ void __launch_bounds__(T) k(unsigned long long* cycles) {
    __shared__ double sd[SLOTS];
    __shared__ uint32_t su[SLOTS];
    __shared__ unsigned long long sl[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += T) { sd[i] = 0; su[i] = 0; sl[i] = 0; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t col = COLS == 0 ? lane : COLS == 1 ? (lane & 31u) : COLS == 2 ? (lane & 15u) : COLS == 3 ? ((lane * 5u) & 31u) : COLS == 5 ? (lane & 7u) : 0u;
    uint32_t r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long c0 = clock64();
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        r = r * 1664525u + 1013904223u;
        const uint32_t s = COLS == 4 ? (r >> 12) & (SLOTS - 1) : (col + ((r >> 12) & (uint32_t)(ROWS - 1)) * (uint32_t)ROWSTRIDE) & (SLOTS - 1);
        if (MODE == 0) atomicAdd(&sd[s], 1.0);
        if (MODE == 1) atomicAdd(&su[s], 1u);
        if (MODE == 2) atomicAdd(&sl[s], 1ull);
    }
    __syncthreads();
    unsigned long long c1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = c1 - c0;
}
static unsigned long long* d_c;
template <int MODE, int COLS, int ROWS, int ROWSTRIDE> void run(const char* name) {
    const int blocks = 256;
    hipLaunchKernelGGL((k<MODE, COLS, ROWS, ROWSTRIDE>), dim3(blocks), dim3(T), 0, 0, d_c);
    hipLaunchKernelGGL((k<MODE, COLS, ROWS, ROWSTRIDE>), dim3(blocks), dim3(T), 0, 0, d_c);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= blocks;
    printf("%-10s %-58s %7.2f clk per wave-instruction\n", MODE == 0 ? "ds_add_f64" : (MODE == 1 ? "ds_add_u32" : "ds_add_u64"), name, m / ITER / (T / 64));
}
#define ALL(M)                                                                        \
    run<M, 0, 1, 64>("64 distinct cols, 1 row (= tid)");                               \
    run<M, 0, 32, 64>("64 distinct cols, random row of 32 (stride 64)");               \
    run<M, 1, 1, 32>("cols lane&31, 1 row: lanes l, l+32 SAME slot");                  \
    run<M, 1, 64, 32>("cols lane&31, random row of 64 (stride 32)");                   \
    run<M, 1, 32, 64>("cols lane&31, random row of 32 (stride 64)");                   \
    run<M, 3, 64, 32>("cols (5*lane)&31, random row of 64 (stride 32)");               \
    run<M, 2, 128, 16>("cols lane&15, random row of 128 (stride 16)");                 \
    run<M, 2, 32, 64>("cols lane&15, random row of 32 (stride 64)");                   \
    run<M, 5, 32, 64>("cols lane&7, random row of 32 (stride 64)");                    \
    run<M, 4, 1, 1>("random slot of 2048");
int main() {
    hipMalloc(&d_c, 256 * 8);
    ALL(0) ALL(1) ALL(2)
    return 0;
}
