// LDS fp32 atomic-add (ds_add_f32) issue-rate probe for gfx950: cycles per wave64 instruction for a few address patterns.
// Build: hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o lds_atomic_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx;
    if (KIND == 0 || KIND >= 5) idx = wave * 64 + lane;                   // conflict-free, one bank per lane
    else if (KIND == 1) idx = wave * 64 + (lane >> 1);                    // 2 lanes share an address
    else if (KIND == 2) idx = wave * 4096 + (lane >> 1) * 33 + (lane & 1) * 16;   // the warp-backward pattern (texel stride 33)
    else if (KIND == 3) idx = wave * 64 + (lane & 15);                    // 4 lanes share an address
    else idx = wave * 4096 + lane * 64;                                   // all lanes in one bank (64-way conflict)
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 5) { volatile float* p = &lds[idx + j]; *p = *p + 1.0f; }   // plain read-modify-write for reference
            else if (KIND == 6) atomicAdd(reinterpret_cast<unsigned*>(&lds[idx + j]), 3u);                  // ds_add_u32
            else if (KIND == 7) atomicAdd(reinterpret_cast<unsigned long long*>(&lds[2 * (idx + j)]), 3ull);   // ds_add_u64
            else if (KIND == 8) atomicMax(reinterpret_cast<int*>(&lds[idx + j]), i);                        // ds_max_i32
            else atomicAdd(&lds[idx + j], 1.0f);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const char* names[9] = {"conflict-free", "2 lanes/address", "texel-stride-33 pairs", "4 lanes/address", "64-way bank conflict", "plain rmw (no atomic)",
                            "ds_add_u32 conflict-free", "ds_add_u64 conflict-free", "ds_max_i32 conflict-free"};
    const int iters = 1000;
    for (int k = 0; k < 9; ++k) {
        for (int blocks : {1}) {
            if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 6) hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 7) hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            if (k == 8) hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            (void)hipDeviceSynchronize();
            unsigned long long c;
            (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-24s blocks %4d: %7.1f clk per wave-instr (one wave's view; 4 waves per block share the LDS)\n", names[k], blocks, (double)c / (iters * 16.0));
        }
    }
    return 0;
}
