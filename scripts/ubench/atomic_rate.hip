// Global fp32 atomic-add throughput probe for gfx950 (one launch per variant, HIP-event timed).
//   0: agent-scope atomics, all blocks into ONE buffer, coalesced (lane i -> word i of a 256-B run)
//   1: same, scattered (lane i -> word 64 i)
//   2: workgroup-scope atomics (no sc1: executed in the issuing XCD's L2) into a per-XCD copy selected by HW_REG_XCC_ID
//   3: plain (non-atomic) read-modify-write of a block-private region, for the store-rate ceiling
// Build: hipcc -O3 --offload-arch=gfx950 atomic_rate.hip -o atomic_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* buf, long words, int iters) {
    const long base = ((long)blockIdx.x * 2654435761u) % (words - 64 * 256 - 256);
    float* p = buf;
    if (KIND == 2) p = buf + (long)xcc_id() * words;
    for (int i = 0; i < iters; ++i) {
        const long o = (base + (long)i * 256 * (KIND == 1 ? 64 : 1)) % (words - 64 * 256 - 256);
        if (KIND == 0) unsafeAtomicAdd(p + o + threadIdx.x, 1.0f);
        else if (KIND == 1) unsafeAtomicAdd(p + o + (long)threadIdx.x * 64 % (64 * 256), 1.0f);
        else if (KIND == 2) __hip_atomic_fetch_add(p + o + threadIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else { float* q = buf + ((long)blockIdx.x * 256 + threadIdx.x) % words; *q += 1.0f; }
    }
}

int main() {
    const long words = 128L * 160 * 32;   // one source-view gradient map at the headline size (2.6 MB)
    float* buf;
    hipMalloc(&buf, words * 4 * 8);
    hipMemset(buf, 0, words * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 8192, iters = 256;
    const char* names[4] = {"agent coalesced", "agent scattered", "workgroup-scope per-XCD copy", "plain rmw"};
    for (int k = 0; k < 4; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, buf, words, iters);
            if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, buf, words, iters);
            if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, buf, words, iters);
            if (k == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, buf, words, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-32s %8.3f ms  %8.1f G lane-atomics/s\n", names[k], ms, (double)blocks * 256 * iters / ms * 1e-6);
        }
    }
    // check variant 2 sums: total over the 8 copies must equal blocks*256*iters*2 (two reps)
    float* h = (float*)malloc(words * 4 * 8);
    hipMemcpy(h, buf, words * 4 * 8, hipMemcpyDeviceToHost);
    double tot = 0;
    int used = 0;
    for (int c = 0; c < 8; ++c) { double s = 0; for (long i = 0; i < words; ++i) s += h[c * words + i]; tot += s; used += s > 0; }
    printf("copies used %d, total %.0f (expect >= %.0f from the atomic variants)\n", used, tot, (double)blocks * 256 * iters * 2 * 3);
    return 0;
}
