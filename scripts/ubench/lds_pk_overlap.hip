// Stand-alone reproducer attempt for the co-scheduling defect of DESIGN.md section 6 (round 3: the packed-fp32 build of
// warp_cost_lds_kernel returned wrong voxels in lanes 48-63 while LDS + MFMA conv waves of another stream shared its CUs).
//   victim  : 256 threads, 40 KB dynamic LDS (four workgroups per CU, like the warp kernel); per round 8 x ds_read_b128 ->
//             s_waitcnt lgkmcnt(0) -> the warp kernel's blend (4 v_pk_mul_f32 + 12 v_pk_fma_f32) and its two sums (4 v_pk_add_f32
//             + 4 v_pk_fma_f32); PK=0 builds the same IEEE operation chain from v_mul_f32 / v_fma_f32 / v_add_f32.
//   partner : 256 threads, dynamic LDS, ds_write / ds_read_b128 feeding v_mfma_f32_16x16x32_f16, on a second stream.
// Self-checking: every launch's sums are compared bit for bit with (a) the CPU's fmaf() evaluation of the same chain, once, and
// (b) the launch made alone; mismatches are counted per 16-lane group.
//   hipcc --offload-arch=gfx950 -O3 -o lds_pk_overlap lds_pk_overlap.hip && ./lds_pk_overlap [launches=200] [rounds=96]
//   variants: -DFIX_NOP (s_nop 7 x2 behind the wait)  -DFIX_B64 (ds_read_b64 x2 per tap piece)  -DFIX_MOV (v_mov_b32 of each pair's low half)
//             -DOPSEL_HI (one weight broadcast from the HIGH half of a register pair, `op_sel:[0,1,0]`: the operand form round 4's
//             bisect of the real kernel isolated -- scripts/dev/pk_variants.sh eblend / eblendhi)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int LDS_FLOATS = 10240, WG = 256, NWG = 4096;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__host__ __device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); }
__host__ __device__ inline float pat(unsigned i) { return (float)(hash(i) >> 12) * (1.0f / 1048576.0f) - 0.5f; }
__host__ __device__ inline unsigned tap_base(unsigned gtid, int r) { return (hash(gtid * 131u + r) % (LDS_FLOATS / 4 - 160)) * 4; }   // float index, 16-B aligned
__host__ __device__ inline float wgt(unsigned gtid, int r, int k) { return (float)(hash(gtid * 977u + r * 4 + k) >> 16) * (1.0f / 65536.0f); }

template <int PK> __global__ __launch_bounds__(WG, 4) void victim(float* out, int rounds) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < LDS_FLOATS; i += WG) lds[i] = pat(i);
    __syncthreads();
    const unsigned gtid = blockIdx.x * WG + threadIdx.x;
    f2 s[4] = {}, q[4] = {};
    for (int r = 0; r < rounds; ++r) {
        const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + tap_base(gtid, r) * 4;
        f4 t[8];
#ifdef FIX_B64
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f2 lo, hi;
            asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(a + k * 80));
            asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(hi) : "v"(a + k * 80));
            t[k] = f4{lo.x, lo.y, hi.x, hi.y};
        }
#else
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(t[k]) : "v"(a + k * 80));
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#ifdef FIX_NOP
        asm volatile("s_nop 7\n s_nop 7" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#endif
#ifdef FIX_MOV
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1" : "+v"(t[k].x), "+v"(t[k].z));
#endif
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = wgt(gtid, r, k);
        f2 o[4];
        if (PK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 lo = (j & 1) ? f2{t[j >> 1].z, t[j >> 1].w} : f2{t[j >> 1].x, t[j >> 1].y};
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o[j]) : "v"(lo), "v"(f2{w[0], w[0]}));
            }
            const f2 w21 = f2{w[2], w[1]};       // OPSEL_HI: two weights in ONE register pair, like the SLP vectorizer packs them
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 tt = t[2 * k + (j >> 1)];
                    const f2 lo = (j & 1) ? f2{tt.z, tt.w} : f2{tt.x, tt.y};
#ifdef OPSEL_HI
                    // tap 1 broadcasts the HIGH half of the pair into both result lanes (op_sel:[0,1,0]): THE form that fails
                    if (k == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(o[j]) : "v"(lo), "v"(w21));
                    else if (k == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(o[j]) : "v"(lo), "v"(w21));
                    else
#endif
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(o[j]) : "v"(lo), "v"(f2{w[k], w[k]}));
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s[j]) : "v"(o[j]));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(q[j]) : "v"(o[j]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v) : "v"(t[j >> 1][2 * (j & 1) + e]), "v"(w[0]));
#pragma unroll
                    for (int k = 1; k < 4; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v) : "v"(t[2 * k + (j >> 1)][2 * (j & 1) + e]), "v"(w[k]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[j][e]) : "v"(v));
                    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(q[j][e]) : "v"(v));
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { reinterpret_cast<f2*>(out)[(size_t)gtid * 8 + j] = s[j]; reinterpret_cast<f2*>(out)[(size_t)gtid * 8 + 4 + j] = q[j]; }
}

__global__ __launch_bounds__(WG) void partner(float* out, int rounds) {   // LDS ring + MFMA, like the conv sweeps
    extern __shared__ __attribute__((aligned(16))) _Float16 ring[];
    const int n8 = 69632 / 16;
    f4 acc[4] = {};
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < n8; i += WG) reinterpret_cast<f4*>(ring)[i] = f4{(float)r, 1.0f, 2.0f, (float)i};
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {
            const h8 a = reinterpret_cast<const h8*>(ring)[(threadIdx.x * 7 + k * 64) % n8], b = reinterpret_cast<const h8*>(ring)[(threadIdx.x + k * 129) % n8];
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
        }
        __syncthreads();
    }
    reinterpret_cast<f4*>(out)[blockIdx.x * WG + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

static void cpu_expect(std::vector<float>& e, int rounds, int n_threads) {      // the same chain with fmaf(), for the first n_threads threads
    std::vector<float> lds(LDS_FLOATS);
    for (int i = 0; i < LDS_FLOATS; ++i) lds[i] = pat(i);
    for (int g = 0; g < n_threads; ++g) {
        float s[8] = {}, q[8] = {};
        for (int r = 0; r < rounds; ++r) {
            const unsigned b = tap_base(g, r);
            for (int c = 0; c < 8; ++c) {      // channel c: tap piece (c >> 2), element c & 3
                float v = lds[b + (c >> 2) * 20 + (c & 3)] * wgt(g, r, 0);
                for (int k = 1; k < 4; ++k) v = fmaf(lds[b + (2 * k + (c >> 2)) * 20 + (c & 3)], wgt(g, r, k), v);
                s[c] += v; q[c] = fmaf(v, v, q[c]);
            }
        }
        for (int c = 0; c < 8; ++c) { e[g * 16 + c] = s[c]; e[g * 16 + 8 + c] = q[c]; }
    }
}

template <int PK> static long run(const char* tag, int launches, int rounds, bool with_partner) {
    const size_t n = (size_t)NWG * WG * 16;
    float *out, *pout; CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&pout, (size_t)1024 * WG * 16));
    CK(hipFuncSetAttribute((const void*)partner, hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::vector<float> gold(n), got(n), cpu(4096 * 16);
    victim<PK><<<NWG, WG, LDS_FLOATS * 4, sa>>>(out, rounds); CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(gold.data(), out, n * 4, hipMemcpyDeviceToHost));
    cpu_expect(cpu, rounds, 4096);
    long cpu_bad = 0; for (int i = 0; i < 4096 * 16; ++i) cpu_bad += memcmp(&cpu[i], &gold[i], 4) != 0;
    long bad_launches = 0, bad_vals = 0, hist[4] = {0, 0, 0, 0};
    for (int it = 0; it < launches; ++it) {
        if (with_partner) for (int p = 0; p < 4; ++p) partner<<<1024, WG, 69632, sb>>>(pout, 40);
        CK(hipMemsetAsync(out, 0, n * 4, sa));
        victim<PK><<<NWG, WG, LDS_FLOATS * 4, sa>>>(out, rounds);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost));
        long b = 0; for (size_t i = 0; i < n; ++i) if (memcmp(&got[i], &gold[i], 4)) { ++b; ++hist[((i / 16) & 63) >> 4]; }
        bad_launches += b != 0; bad_vals += b;
    }
    printf("%s PK=%d partner=%d: solo launch vs CPU fmaf chain: %ld of %d values differ; %ld of %d launches differ from the solo launch (%ld values; lanes 0-15 / 16-31 / 32-47 / 48-63: %ld %ld %ld %ld)\n",
           tag, PK, (int)with_partner, cpu_bad, 4096 * 16, bad_launches, launches, bad_vals, hist[0], hist[1], hist[2], hist[3]);
    CK(hipFree(out)); CK(hipFree(pout));
    return bad_launches;
}

// Second victim (round 4), no LDS at all: packed fp32 instructions whose LOW result lane selects the HIGH half of one source pair
// (`op_sel` bit set), each checked IN the kernel against plain v_mul / v_add / v_fma of the same operands; mismatches are counted per
// 16-lane group.  Row order of errs[9][4]: v_pk_mov_b32 [1,0]; v_pk_mul_f32 [1,0], [0,1]; v_pk_add_f32 [1,0], [0,1];
// v_pk_fma_f32 [1,0,0], [0,1,0], [0,0,1]; and v_pk_fma_f32 op_sel_hi:[1,0,1] (the HIGH lane reading a LOW half: the safe broadcast).
#define OPSEL_CASE(row, pk_asm, lo_expr, hi_expr)                                                                          \
    { f2 d; asm volatile(pk_asm : "=v"(d) : "v"(a), "v"(b), "v"(c));                                                     \
      const float lo = lo_expr, hi = hi_expr;                                                                              \
      bad[row] += (__float_as_uint(d.x) != __float_as_uint(lo)) || (__float_as_uint(d.y) != __float_as_uint(hi)); }
__device__ __forceinline__ float s_mul(float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float s_add(float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float s_fma(float x, float y, float z) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z) : "v"(x), "v"(y)); return z; }
__global__ __launch_bounds__(WG, 4) void opsel_victim(unsigned* errs, int rounds) {
    const unsigned gtid = blockIdx.x * WG + threadIdx.x;
    const int grp = (threadIdx.x & 63) >> 4;
    unsigned bad[9] = {};
    for (int r = 0; r < rounds; ++r) {
        f2 a = f2{pat(gtid * 7u + r), pat(gtid * 11u + r + 1)}, b = f2{pat(gtid * 13u + r + 2), pat(gtid * 17u + r + 3)};
        f2 c = f2{pat(gtid * 19u + r + 4), pat(gtid * 23u + r + 5)};
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        OPSEL_CASE(0, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]", a.y, b.x)
        OPSEL_CASE(1, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", s_mul(a.y, b.x), s_mul(a.y, b.y))
        OPSEL_CASE(2, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", s_mul(a.x, b.y), s_mul(a.y, b.y))
        OPSEL_CASE(3, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]", s_add(a.y, b.x), s_add(a.y, b.y))
        OPSEL_CASE(4, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", s_add(a.x, b.y), s_add(a.y, b.y))
        OPSEL_CASE(5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", s_fma(a.y, b.x, c.x), s_fma(a.y, b.y, c.y))
        OPSEL_CASE(6, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", s_fma(a.x, b.y, c.x), s_fma(a.y, b.y, c.y))
        OPSEL_CASE(7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", s_fma(a.x, b.x, c.y), s_fma(a.y, b.y, c.y))
        OPSEL_CASE(8, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", s_fma(a.x, b.x, c.x), s_fma(a.y, b.x, c.y))
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
        if (bad[k]) atomicAdd(errs + 4 * k + grp, bad[k]);
}
extern "C" int lpo_opsel_victim(unsigned* errs36, int rounds, void* stream) {
    opsel_victim<<<NWG, WG, 0, (hipStream_t)stream>>>(errs36, rounds);
    return (int)hipGetLastError();
}

// the victim as a library entry (scripts/dev/pk_probe.py launches it beside the engine's real conv0 through tests/test_gpu_overlap.py's harness):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o liblpo.so lds_pk_overlap.hip
extern "C" int lpo_victim(float* out, int rounds, int pk, void* stream) {
    if (pk) victim<1><<<NWG, WG, LDS_FLOATS * 4, (hipStream_t)stream>>>(out, rounds);
    else victim<0><<<NWG, WG, LDS_FLOATS * 4, (hipStream_t)stream>>>(out, rounds);
    return (int)hipGetLastError();
}
extern "C" long lpo_out_floats(void) { return (long)NWG * WG * 16; }

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200, rounds = argc > 2 ? atoi(argv[2]) : 96;
    run<1>("packed", launches, rounds, false);
    const long bp = run<1>("packed", launches, rounds, true);
    const long bs = run<0>("scalar", launches, rounds, true);
    printf("RESULT packed_bad=%ld scalar_bad=%ld\n", bp, bs);
    return 0;
}
