// Second vector-ALU issue-rate probe for gfx950 (round 2): the 16-bit packed / dot / convert / permute forms the
// LDS-staged warp kernel could use for its blend, and ds_read_b128 issued beside them.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x

#define OP8_ACC(ins)                                                                                                     \
    REP16(asm volatile(ins " %0, %8, %9, %0\n " ins " %1, %8, %9, %1\n " ins " %2, %8, %9, %2\n " ins " %3, %8, %9, %3\n" \
                       ins " %4, %8, %9, %4\n " ins " %5, %8, %9, %5\n " ins " %6, %8, %9, %6\n " ins " %7, %8, %9, %7\n" \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h), "v"(g));)
#define OP8_2(ins)                                                                                                   \
    REP16(asm volatile(ins " %0, %8, %0\n " ins " %1, %8, %1\n " ins " %2, %8, %2\n " ins " %3, %8, %3\n"             \
                       ins " %4, %8, %4\n " ins " %5, %8, %5\n " ins " %6, %8, %6\n " ins " %7, %8, %7\n"             \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)
#define OP8_1(ins, suffix)                                                                                           \
    REP16(asm volatile(ins " %0, %8 " suffix "\n " ins " %1, %8 " suffix "\n " ins " %2, %8 " suffix "\n " ins " %3, %8 " suffix "\n" \
                       ins " %4, %8 " suffix "\n " ins " %5, %8 " suffix "\n " ins " %6, %8 " suffix "\n " ins " %7, %8 " suffix "\n" \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h));)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
    __shared__ uint4 lds[1024];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t h = 0x3c003c00u, g = 0x38003800u;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_uint4(h, g, h, g);
    __syncthreads();
    uint32_t la = (threadIdx.x * 16) & 16383;
    uint4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { OP8_ACC("v_pk_fma_f16") }
        else if (KIND == 1) { OP8_2("v_pk_mul_f16") }
        else if (KIND == 2) { OP8_2("v_pk_add_f16") }
        else if (KIND == 3) { OP8_ACC("v_dot2_f32_f16") }
        else if (KIND == 4) { OP8_ACC("v_dot2_f32_bf16") }
        else if (KIND == 5) { OP8_2("v_dot2c_f32_f16") }
        else if (KIND == 6) { OP8_2("v_lshlrev_b32") }
        else if (KIND == 7) { OP8_2("v_and_b32") }
        else if (KIND == 8) { OP8_ACC("v_perm_b32") }
        else if (KIND == 9) { OP8_1("v_cvt_f32_f16", "") }
        else if (KIND == 10) { OP8_1("v_cvt_f32_f16_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") }
        else if (KIND == 11) { OP8_2("v_fmac_f32") }
        else if (KIND == 13) { OP8_2("v_add_f32") }
        else if (KIND == 14) { OP8_ACC("v_fma_f32") }
        else if (KIND == 15) {   // v_fma_mix_f32 lo/hi
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h), "v"(a0));)
        } else if (KIND == 16) {  // v_fma_mixlo_f16 (f32 x f32 + f32 -> f16 lo)
            REP16(asm volatile("v_fma_mixlo_f16 %0, %8, %9, %0\n v_fma_mixlo_f16 %1, %8, %9, %1\n v_fma_mixlo_f16 %2, %8, %9, %2\n v_fma_mixlo_f16 %3, %8, %9, %3\n"
                               "v_fma_mixlo_f16 %4, %8, %9, %4\n v_fma_mixlo_f16 %5, %8, %9, %5\n v_fma_mixlo_f16 %6, %8, %9, %6\n v_fma_mixlo_f16 %7, %8, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h), "v"(g));)
        } else if (KIND == 17) {  // ds_read_b128 alone: 8 per block x16 = 128 per iter
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                               "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                               "s_waitcnt lgkmcnt(0)\n"
                               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(la));)
        } else if (KIND == 18) {  // 4 ds_read_b128 + 32 v_pk_fma_f16 (the pk blend's mix: 8 per tap-load)
            REP16(asm volatile("ds_read_b128 %8, %12\n ds_read_b128 %9, %12 offset:1024\n ds_read_b128 %10, %12 offset:2048\n ds_read_b128 %11, %12 offset:3072\n"
                               "v_pk_fma_f16 %0, %13, %14, %0\n v_pk_fma_f16 %1, %13, %14, %1\n v_pk_fma_f16 %2, %13, %14, %2\n v_pk_fma_f16 %3, %13, %14, %3\n"
                               "v_pk_fma_f16 %4, %13, %14, %4\n v_pk_fma_f16 %5, %13, %14, %5\n v_pk_fma_f16 %6, %13, %14, %6\n v_pk_fma_f16 %7, %13, %14, %7\n"
                               "v_pk_fma_f16 %0, %13, %14, %0\n v_pk_fma_f16 %1, %13, %14, %1\n v_pk_fma_f16 %2, %13, %14, %2\n v_pk_fma_f16 %3, %13, %14, %3\n"
                               "v_pk_fma_f16 %4, %13, %14, %4\n v_pk_fma_f16 %5, %13, %14, %5\n v_pk_fma_f16 %6, %13, %14, %6\n v_pk_fma_f16 %7, %13, %14, %7\n"
                               "s_waitcnt lgkmcnt(0)\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                 "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(la), "v"(h), "v"(g));)
        } else if (KIND == 19) {  // 4 ds_read_b128 + 32 v_fma_f32 (fp32 blend issue mix)
            REP16(asm volatile("ds_read_b128 %8, %12\n ds_read_b128 %9, %12 offset:1024\n ds_read_b128 %10, %12 offset:2048\n ds_read_b128 %11, %12 offset:3072\n"
                               "v_fma_f32 %0, %13, %14, %0\n v_fma_f32 %1, %13, %14, %1\n v_fma_f32 %2, %13, %14, %2\n v_fma_f32 %3, %13, %14, %3\n"
                               "v_fma_f32 %4, %13, %14, %4\n v_fma_f32 %5, %13, %14, %5\n v_fma_f32 %6, %13, %14, %6\n v_fma_f32 %7, %13, %14, %7\n"
                               "v_fma_f32 %0, %13, %14, %0\n v_fma_f32 %1, %13, %14, %1\n v_fma_f32 %2, %13, %14, %2\n v_fma_f32 %3, %13, %14, %3\n"
                               "v_fma_f32 %4, %13, %14, %4\n v_fma_f32 %5, %13, %14, %5\n v_fma_f32 %6, %13, %14, %6\n v_fma_f32 %7, %13, %14, %7\n"
                               "v_fma_f32 %0, %13, %14, %0\n v_fma_f32 %1, %13, %14, %1\n v_fma_f32 %2, %13, %14, %2\n v_fma_f32 %3, %13, %14, %3\n"
                               "v_fma_f32 %4, %13, %14, %4\n v_fma_f32 %5, %13, %14, %5\n v_fma_f32 %6, %13, %14, %6\n v_fma_f32 %7, %13, %14, %7\n"
                               "v_fma_f32 %0, %13, %14, %0\n v_fma_f32 %1, %13, %14, %1\n v_fma_f32 %2, %13, %14, %2\n v_fma_f32 %3, %13, %14, %3\n"
                               "v_fma_f32 %4, %13, %14, %4\n v_fma_f32 %5, %13, %14, %5\n v_fma_f32 %6, %13, %14, %6\n v_fma_f32 %7, %13, %14, %7\n"
                               "s_waitcnt lgkmcnt(0)\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                 "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(la), "v"(a0), "v"(a1));)
        } else if (KIND == 20) {  // v_pk_fma_f16 with op_sel broadcast of a scalar weight half (weights packed two per register)
            REP16(asm volatile("v_pk_fma_f16 %0, %8, %9, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f16 %1, %8, %9, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                               "v_pk_fma_f16 %2, %8, %9, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f16 %3, %8, %9, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                               "v_pk_fma_f16 %4, %8, %9, %4 op_sel_hi:[1,0,1]\n v_pk_fma_f16 %5, %8, %9, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                               "v_pk_fma_f16 %6, %8, %9, %6 op_sel_hi:[1,0,1]\n v_pk_fma_f16 %7, %8, %9, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h), "v"(g));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + __uint_as_float(r0.x ^ r1.y ^ r2.z ^ r3.w);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, int wgs, int wpb, double instr_per_block) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(wgs), dim3(64 * wpb), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(wgs), dim3(64 * wpb), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ninstr = (double)iters * 16 * instr_per_block;   // per wave
    const double wps = (double)wgs * wpb / 1024.0;
    const double ns = ms * 1e6 / (ninstr * wps);
    printf("%-28s waves/SIMD=%.0f  %.3f ms  -> %.3f ns = %.2f cyc@2.4GHz per wave-instr per SIMD; one wave's view %.2f cyc\n",
           name, wps, ms, ns, ns * 2.4, (double)c / ninstr);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int wpb = 1; wpb <= 4; wpb *= 4) {
        const int wgs = 1024;
        run<14>("v_fma_f32", wgs, wpb, 8);
        run<11>("v_fmac_f32 (vop2)", wgs, wpb, 8);
        run<13>("v_add_f32 (vop2)", wgs, wpb, 8);
        run<15>("v_fma_mix_f32", wgs, wpb, 8);
        run<16>("v_fma_mixlo_f16", wgs, wpb, 8);
        run<0>("v_pk_fma_f16", wgs, wpb, 8);
        run<20>("v_pk_fma_f16 op_sel bcast", wgs, wpb, 8);
        run<1>("v_pk_mul_f16", wgs, wpb, 8);
        run<2>("v_pk_add_f16", wgs, wpb, 8);
        run<3>("v_dot2_f32_f16", wgs, wpb, 8);
        run<4>("v_dot2_f32_bf16", wgs, wpb, 8);
        run<5>("v_dot2c_f32_f16", wgs, wpb, 8);
        run<6>("v_lshlrev_b32", wgs, wpb, 8);
        run<7>("v_and_b32", wgs, wpb, 8);
        run<8>("v_perm_b32", wgs, wpb, 8);
        run<9>("v_cvt_f32_f16", wgs, wpb, 8);
        run<10>("v_cvt_f32_f16 sdwa hi", wgs, wpb, 8);
        run<17>("ds_read_b128 x8", wgs, wpb, 8);
        run<18>("4 ds_read_b128 + 16 pk_fma_f16 (per instr, 20)", wgs, wpb, 20);
        run<19>("4 ds_read_b128 + 32 fma_f32 (per instr, 36)", wgs, wpb, 36);
    }
    return 0;
}
