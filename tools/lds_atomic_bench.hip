// Microbenchmark (diagnostic, not part of the product): LDS atomic throughput on gfx950 per CU.
// One 1024-thread workgroup per CU; each thread issues ITER atomics to an LDS array of SLOTS entries.
// Slot patterns:
//   random      hash of thread / iteration (what a hash table sees)
//   tid         slot = thread index (conflict-free, every lane its own row)
//   banked      a random ROW per lane, but the 32 lanes of each half-wave sit in 32 distinct (slot mod 32) classes:
//               what the bank-binned accumulate of k_bucket_direct produces
//   active K    `tid` pattern with only one lane in K active (EXEC-masked instruction: does the LDS price an atomic by
//               its active lanes or per instruction?)
// Build:  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_atomic_bench
//         (-munsafe-fp-atomics: atomicAdd(float*) on LDS becomes the native ds_add_f32; without it hipcc expands a
//          compare-and-swap loop, which is what profiles/r02_lds_atomic_rates.txt measured as "0.33 lanes per clock")
// Reports lane-ops per cycle and CU (memtime ticks at 100 MHz are converted with the measured shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int T = 1024, SLOTS = 1024, ITER = 256;
__device__ inline uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
enum { PAT_RANDOM = 0, PAT_TID = 1, PAT_BANKED = 2 };
template <int MODE, int PAT, int ACTIVE_EVERY>
__global__ void __launch_bounds__(T) k(unsigned long long* cycles, double* sink) {
    __shared__ double sd[SLOTS];
    __shared__ float sf[SLOTS];
    __shared__ uint32_t su[SLOTS];
    __shared__ unsigned long long sl[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += T) { sd[i] = 0; sf[i] = 0; su[i] = 0; sl[i] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c0 = clock64();
    uint32_t acc = 0;
    const bool active = (threadIdx.x % ACTIVE_EVERY) == 0;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r = hsh(threadIdx.x * 977u + it * 131071u + blockIdx.x);
        uint32_t s = threadIdx.x;
        if (PAT == PAT_RANDOM) s = r & (SLOTS - 1);
        if (PAT == PAT_BANKED) s = ((r & (SLOTS / 32 - 1)) << 5) | (threadIdx.x & 31u);
        if (active) {
            if (MODE == 0) atomicAdd(&su[s], 1u);                       // ds_add_u32
            if (MODE == 1) acc += atomicAdd(&su[s], 1u);                // ds_add_rtn_u32
            if (MODE == 2) atomicAdd(&sf[s], 1.0f);                     // ds_add_f32 (native with -munsafe-fp-atomics)
            if (MODE == 3) atomicAdd(&sd[s], 1.0);                      // ds_add_f64
            if (MODE == 4) atomicAdd(&sl[s], 1ull);                     // ds_add_u64
            if (MODE == 5) atomicMin(&su[s], (uint32_t)it);             // ds_min_u32
        }
    }
    __syncthreads();
    unsigned long long c1 = clock64();
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cycles[2 * blockIdx.x] = c1 - c0; cycles[2 * blockIdx.x + 1] = t1 - t0; }
    if (acc == 0x12345678u) sink[0] = sd[3] + sf[4] + su[5] + sl[6];
}
template <int MODE, int PAT, int AE = 1> void run(const char* name, unsigned long long* d_c, double* d_s, int blocks) {
    hipLaunchKernelGGL((k<MODE, PAT, AE>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipLaunchKernelGGL((k<MODE, PAT, AE>), dim3(blocks), dim3(T), 0, 0, d_c, d_s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(2 * blocks);
    hipMemcpy(c.data(), d_c, 2 * blocks * 8, hipMemcpyDeviceToHost);
    double m = 0, w = 0; for (int i = 0; i < blocks; ++i) { m += c[2 * i]; w += c[2 * i + 1]; } m /= blocks; w /= blocks;
    const char* pat = PAT == PAT_RANDOM ? "random" : (PAT == PAT_TID ? "tid   " : "banked");
    const double ops = (double)(T / AE) * ITER;
    // clock64 = s_memtime (shader clock ticks on gfx950 builds that map it so); s_memrealtime = 100 MHz wall ticks
    printf("%-16s %s active 1/%-2d  %9.0f clk  %7.2f lanes/clk/CU  %6.2f clk per wave-instruction of one wave  (%.0f ns)\n", name, pat, AE, m, ops / m,
           m / ITER / (T / 64) , w * 10.0);
}
int main() {
    unsigned long long* d_c; double* d_s;
    int blocks = 256;
    hipMalloc(&d_c, 2 * blocks * 8); hipMalloc(&d_s, 64);
#define ROW(M, N) run<M, PAT_RANDOM>(N, d_c, d_s, blocks); run<M, PAT_TID>(N, d_c, d_s, blocks); run<M, PAT_BANKED>(N, d_c, d_s, blocks);
    ROW(0, "ds_add_u32") ROW(1, "ds_add_rtn_u32") ROW(2, "ds_add_f32") ROW(3, "ds_add_f64") ROW(4, "ds_add_u64") ROW(5, "ds_min_u32")
    // priced per instruction or per active lane?
    run<3, PAT_TID, 2>("ds_add_f64", d_c, d_s, blocks); run<3, PAT_TID, 4>("ds_add_f64", d_c, d_s, blocks); run<3, PAT_TID, 8>("ds_add_f64", d_c, d_s, blocks);
    run<3, PAT_RANDOM, 2>("ds_add_f64", d_c, d_s, blocks); run<3, PAT_RANDOM, 4>("ds_add_f64", d_c, d_s, blocks); run<3, PAT_RANDOM, 8>("ds_add_f64", d_c, d_s, blocks);
    run<2, PAT_TID, 4>("ds_add_f32", d_c, d_s, blocks); run<0, PAT_TID, 4>("ds_add_u32", d_c, d_s, blocks);
    return 0;
}
