// Vector-ALU issue-rate probe for gfx950: how many cycles does a wave64 instruction of each kind occupy a SIMD?
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float w = 1.0001f;
    uint32_t h = 0x3c003c00u;   // (1.0h, 1.0h)
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pw = {w, w};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // v_fma_f32, 8 independent chains x 16
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        } else if (KIND == 1) { // v_fma_mix_f32 (f16 src0)
            REP16(asm volatile("v_fma_mix_f32 %0, %9, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %9, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %9, %8, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %9, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %9, %8, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %9, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %9, %8, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %9, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(h));)
        } else if (KIND == 2) { // v_pk_fma_f32 (4 chains x 2 = same 8 per block)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                               "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pw));)
        } else if (KIND == 3) { // v_cvt_f32_f16
            REP16(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
                               "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)
        } else if (KIND == 4) { // v_pk_fma_f16
            REP16(asm volatile("v_pk_fma_f16 %0, %0, %8, %0\n v_pk_fma_f16 %1, %1, %8, %1\n v_pk_fma_f16 %2, %2, %8, %2\n v_pk_fma_f16 %3, %3, %8, %3\n"
                               "v_pk_fma_f16 %4, %4, %8, %4\n v_pk_fma_f16 %5, %5, %8, %5\n v_pk_fma_f16 %6, %6, %8, %6\n v_pk_fma_f16 %7, %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)
        } else if (KIND == 5) { // v_dot2_f32_f16
            REP16(asm volatile("v_dot2_f32_f16 %0, %8, %8, %0\n v_dot2_f32_f16 %1, %8, %8, %1\n v_dot2_f32_f16 %2, %8, %8, %2\n v_dot2_f32_f16 %3, %8, %8, %3\n"
                               "v_dot2_f32_f16 %4, %8, %8, %4\n v_dot2_f32_f16 %5, %8, %8, %5\n v_dot2_f32_f16 %6, %8, %8, %6\n v_dot2_f32_f16 %7, %8, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)
        } else if (KIND == 6) { // v_mov_b32 dpp quad_perm
            REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %6, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) { // v_add_u32 (integer)
            REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                               "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)
        } else if (KIND == 8) { // v_rcp_f32
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                               "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 9) { // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pw));)
        } else if (KIND == 10) { // v_mul_f32 e32
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[0] + p2[1] + p3[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, int wgs, int wpb) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(wgs), dim3(64 * wpb), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(wgs), dim3(64 * wpb), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ninstr = (double)iters * 128;                 // per wave
    // waves per SIMD = wgs * wpb / 1024 (256 CUs x 4 SIMDs)
    const double wps = (double)wgs * wpb / 1024.0;
    printf("%-16s wgs=%4d waves/SIMD=%.0f  %.3f ms  -> %.2f ns per wave-instr per SIMD; s_memtime cycles/instr (one wave's view) %.2f\n",
           name, wgs, wps, ms, ms * 1e6 / (ninstr * wps), (double)c / ninstr);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int wpb = 1; wpb <= 4; wpb *= 4) {
        const int wgs = 256 * 4 / (wpb == 1 ? 1 : 1);   // wpb=1: 1024 WGs x 1 wave = 1 wave/SIMD; wpb=4: 1024 x 4 = 4 waves/SIMD
        run<0>("v_fma_f32", wgs, wpb);
        run<10>("v_mul_f32", wgs, wpb);
        run<1>("v_fma_mix_f32", wgs, wpb);
        run<2>("v_pk_fma_f32", wgs, wpb);
        run<9>("v_pk_add_f32", wgs, wpb);
        run<3>("v_cvt_f32_f16", wgs, wpb);
        run<4>("v_pk_fma_f16", wgs, wpb);
        run<5>("v_dot2_f32_f16", wgs, wpb);
        run<6>("v_mov_dpp", wgs, wpb);
        run<7>("v_add_u32", wgs, wpb);
        run<8>("v_rcp_f32", wgs, wpb);
    }
    return 0;
}
