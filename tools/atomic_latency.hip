// Latency of what a layer of the cost flood waits for, one wave on an idle chip (tools/atomic_latency.hip; hipcc --offload-arch=gfx950):
// a returning global atomic min, a non-returning one followed by s_waitcnt vmcnt(0), a plain load and an sc1 load — each on lines the
// wave has not touched before (stride 4 KB over a 256 MB buffer) and on lines it touched just before.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* buf, unsigned long long* out, int mode, int warm, size_t stride_words, int reps) {
    const int lane = threadIdx.x;
    unsigned long long sum = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned* p = buf + (size_t)(r * 64 + lane) * stride_words;
        if (warm) { unsigned v = __builtin_nontemporal_load(p); asm volatile("s_waitcnt vmcnt(0)" ::"v"(v)); asm volatile("" ::"v"(v)); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = clock64();
        unsigned got = 0;
        if (mode == 0) { got = atomicMin(p, 0x12345u + lane); asm volatile("s_waitcnt vmcnt(0)" : "+v"(got)); }
        else if (mode == 1) { __hip_atomic_fetch_min(p, 0x12345u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (mode == 2) { got = *(volatile unsigned*)p; asm volatile("s_waitcnt vmcnt(0)" : "+v"(got)); }
        else if (mode == 3) { got = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" : "+v"(got)); }
        else if (mode == 4) { *(volatile unsigned*)p = lane; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned long long t1 = clock64();
        sum += t1 - t0;
        if (got == 0xDEADBEEFu) out[1] = got;
    }
    if (lane == 0) out[0] = sum / reps;
}

int main() {
    const size_t bytes = 256ull << 20;
    unsigned* buf; unsigned long long* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 16);
    const char* names[] = {"atomic umin, returning", "atomic umin, no return + vmcnt(0)", "load", "load sc1 (agent scope)", "store + vmcnt(0)"};
    for (int warm = 0; warm < 2; ++warm)
        for (int mode = 0; mode < 5; ++mode) {
            hipMemset(buf, 0xFF, bytes);
            hipDeviceSynchronize();
            // empty timing pair: the cost of the two clock reads themselves
            unsigned long long h[2] = {0, 0};
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out, mode, warm, (size_t)1024, 32);
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("%-36s %s lines: %llu cycles\n", names[mode], warm ? "just-touched" : "untouched  ", h[0]);
        }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out, 9, 0, (size_t)1024, 32);
    unsigned long long h0 = 0; hipMemcpy(&h0, out, 8, hipMemcpyDeviceToHost);
    printf("two clock reads with nothing between: %llu cycles\n", h0);
    return 0;
}
