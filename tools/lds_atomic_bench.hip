// Microbenchmark (diagnostic, not part of the product): LDS atomic throughput on gfx950 per CU.
// One 1024-thread workgroup per CU; each thread issues ITER atomics to an LDS array of SLOTS entries,
// either at random slots (hash of thread/iter) or conflict-free (slot = tid).  Reports lane-ops/cycle/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int T = 1024, SLOTS = 1024, ITER = 256;
__device__ inline uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE, bool RANDOM>
__global__ void __launch_bounds__(T) k(unsigned long long* cycles, double* sink) {
    __shared__ double sd[SLOTS];
    __shared__ float sf[SLOTS];
    __shared__ uint32_t su[SLOTS];
    __shared__ unsigned long long sl[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += T) { sd[i] = 0; sf[i] = 0; su[i] = 0; sl[i] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int it = 0; it < ITER; ++it) {
        uint32_t s = RANDOM ? (hsh(threadIdx.x * 977u + it * 131071u + blockIdx.x) & (SLOTS - 1)) : threadIdx.x;
        if (MODE == 0) atomicAdd(&su[s], 1u);                       // ds_add_u32
        if (MODE == 1) acc += atomicAdd(&su[s], 1u);                // ds_add_rtn_u32
        if (MODE == 2) atomicAdd(&sf[s], 1.0f);                     // ds_add_f32
        if (MODE == 3) atomicAdd(&sd[s], 1.0);                      // ds_add_f64
        if (MODE == 4) atomicAdd(&sl[s], 1ull);                     // ds_add_u64
        if (MODE == 5) atomicMin(&su[s], (uint32_t)it);             // ds_min_u32
        if (MODE == 6) { sd[s] += 1.0; }                            // plain RMW (racy, for rate only)
        if (MODE == 7) acc += su[s];                                // plain read
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = sd[3] + sf[4] + su[5] + sl[6];
}
template <int MODE, bool RANDOM> void run(const char* name, unsigned long long* d_c, double* d_s, int blocks) {
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += v; m /= blocks;
    printf("%-28s %s  %9.0f cycles  %.2f lane-ops/cycle/CU\n", name, RANDOM ? "random  " : "conflict-free", m, (double)T * ITER / m);
}
int main() {
    unsigned long long* d_c; double* d_s;
    int blocks = 256;
    hipMalloc(&d_c, blocks * 8); hipMalloc(&d_s, 64);
    run<0, true>("ds_add_u32", d_c, d_s, blocks);      run<0, false>("ds_add_u32", d_c, d_s, blocks);
    run<1, true>("ds_add_rtn_u32", d_c, d_s, blocks);  run<1, false>("ds_add_rtn_u32", d_c, d_s, blocks);
    run<2, true>("ds_add_f32", d_c, d_s, blocks);      run<2, false>("ds_add_f32", d_c, d_s, blocks);
    run<3, true>("ds_add_f64", d_c, d_s, blocks);      run<3, false>("ds_add_f64", d_c, d_s, blocks);
    run<4, true>("ds_add_u64", d_c, d_s, blocks);      run<4, false>("ds_add_u64", d_c, d_s, blocks);
    run<5, true>("ds_min_u32", d_c, d_s, blocks);      run<5, false>("ds_min_u32", d_c, d_s, blocks);
    run<6, true>("plain f64 read+write", d_c, d_s, blocks); run<6, false>("plain f64 read+write", d_c, d_s, blocks);
    run<7, true>("plain u32 read", d_c, d_s, blocks);  run<7, false>("plain u32 read", d_c, d_s, blocks);
    return 0;
}
