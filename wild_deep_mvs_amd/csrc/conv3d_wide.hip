// 3x3x3 stride-1 convolution for WIDE layers (32 | 64 input channels -> 32 | 64 output channels) on large volumes: CVP-MVSNet's
// conv2 / conv2a (32 -> 32), conv3 (32 -> 64), conv4 / conv4a (64 -> 64) and conv5 (64 -> 32, stride-1 transposed = a stride-1
// convolution with the flipped kernel).  gfx950.
//
// The brick kernel (conv3d.hip) runs these layers with 256-thread workgroups whose waves each stream ALL A fragments of the layer
// from global memory (216 KB per wave at 64 -> 64, 864 KB per 256 output voxels): with a 93 KB input brick only ONE such workgroup
// fits a CU, i.e. one wave per SIMD whose MFMAs wait on its own weight loads -- 0.25 of the MFMA peak, 0.09 of HBM, bound by the
// L2 -> CU path (4.4 GB of weight re-reads per launch at 4 x 512 x 640).  Here:
//
//   workgroup = 512 threads = 8 waves on the same 4 x 4 x 16 output tile (16 rows of 16 x-adjacent voxels, two per wave, all
//     output channels): two waves per SIMD;
//   the weights pass through an LDS double buffer shared by the eight waves: a stage = 6 (C_in 64) or 9 (C_in 32) k-steps of all
//     output tiles = one contiguous 18..36 KB piece of the packed weights, copied cooperatively (each thread 16 B x <= 5) while the
//     previous stage is contracted, one barrier per stage: the weights enter the CU once per workgroup (216 KB at 64 -> 64), the
//     A fragments are ds_read_b128 like the B fragments;
//   per k-step and wave: NT A reads + 2 B reads feed 2 NT MFMAs; every accumulator sums its k-steps in ascending order like the
//     brick kernel's, and the epilogue is the same operation chain: same stored bits (tests/test_gpu_conv3d.py).
//
// Replaces (fdarmon/wild_deep_mvs): ConvBnReLU3D 32 -> 32, 32 -> 64 (stride 1), 64 -> 64 and the stride-1 ConvTranspose3d block
// 64 -> 32 of models/CVP_MVSNet/models/net.py:50-85 at the refinement levels' sizes.
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 wd_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 wd_f16x8;
typedef __attribute__((ext_vector_type(4))) float wd_f32x4;

template <typename H> struct WdMfma;
template <> struct WdMfma<bf16_t> {
    __device__ static __forceinline__ wd_f32x4 run(const uint4& a, const uint4& b, const wd_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wd_bf16x8, a), __builtin_bit_cast(wd_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WdMfma<f16_t> {
    __device__ static __forceinline__ wd_f32x4 run(const uint4& a, const uint4& b, const wd_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wd_f16x8, a), __builtin_bit_cast(wd_f16x8, b), c, 0, 0, 0);
    }
};

struct WideArgs {
    const uint16_t* in;
    const uint4* wpk;        // kind S1 packing: [k-step][output tile][64 lanes] x 8 halves
    const float* scale;
    const float* bias;
    const float* floor;
    const uint16_t* skip;
    void* out;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, D, Hh, W;
    int epi;
    int ntd, nth, ntw;
    unsigned mg_td, mg_th, mg_tw;
};

constexpr int WD_TD = 4, WD_TH = 4, WD_BD = 6, WD_BH = 6, WD_BW = 18, WD_NVOX = WD_BD * WD_BH * WD_BW;
__host__ __device__ constexpr int wd_vs(int cin) { return cin == 32 ? 96 : cin * 2 + 16; }     // LDS bytes per voxel (conv3d.hip: conflict-free strides)
__host__ __device__ constexpr int wd_stage_steps(int cin) { return cin == 64 ? 6 : 9; }   // k-steps per weight stage (9 / 3 stages per layer)
__host__ __device__ constexpr int wd_lds(int cin, int nt) { return ((WD_NVOX * wd_vs(cin) + 15) & ~15) + 2 * wd_stage_steps(cin) * nt * 1024; }

Knob g_conv_wide = {1, KNOB_CONV_WIDE};       // pscv_set_tuning("conv_wide", 0): these layers back on the brick kernel (A/B runs, bit comparison)

template <typename H, int CIN, int NT>
__global__ __launch_bounds__(512, 2) void conv3d_wide_kernel(const WideArgs a) {
    constexpr int VS = wd_vs(CIN), CCH = CIN / 8, S = wd_stage_steps(CIN), NSTEPS = 27 * CIN / 32, NSTAGE = NSTEPS / S;
    constexpr int BRICK = (WD_NVOX * VS + 15) & ~15, SB = S * NT * 1024;          // bytes: input brick, one weight stage
    constexpr int NCH = WD_NVOX * CCH, NLD = (NCH + 511) / 512;                    // brick chunks (16 B), loads per thread
    constexpr int WLD = (SB / 16 + 511) / 512;                                     // weight-stage loads per thread
    static_assert(NSTAGE * S == NSTEPS, "stages tile the k-steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem + BRICK;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot;
    const int tw_i = fast_divmod(wg, a.ntw, a.mg_tw);
    const int th_i = fast_divmod(wg, a.nth, a.mg_th);
    const int td_i = fast_divmod(wg, a.ntd, a.mg_td);
    const int b = wg;
    const int t0d = td_i * WD_TD, t0h = th_i * WD_TH, t0w = tw_i * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    // ---- stage the input brick (zero padding outside the volume) and the first weight stage: every load in flight before the first LDS write ----
    {
        uint4 val[NLD];
        const long plane = (long)a.Hh * a.W;
        const uint16_t* inb = a.in + (long)b * a.D * plane * a.in_cs + a.in_co;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            const int bd = v / (WD_BH * WD_BW), rem = v - bd * (WD_BH * WD_BW);
            const int bh = rem / WD_BW, bw = rem - bh * WD_BW;
            const int gd = t0d - 1 + bd, gh = t0h - 1 + bh, gw = t0w - 1 + bw;
            val[i] = make_uint4(0u, 0u, 0u, 0u);
            if (c < NCH && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W)
                val[i] = *reinterpret_cast<const uint4*>(inb + ((long)gd * plane + (long)gh * a.W + gw) * a.in_cs + cc * 8);
        }
        uint4 w0[WLD];
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            w0[r] = idx < SB / 16 ? a.wpk[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            if (c < NCH) *reinterpret_cast<uint4*>(smem + v * VS + cc * 16) = val[i];
        }
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            if (idx < SB / 16) *reinterpret_cast<uint4*>(wbuf + idx * 16) = w0[r];
        }
    }
    // epilogue constants and the skip values of this wave's two rows, requested before the contraction
    const int mt0 = 2 * wave;                                   // rows (M-tiles) mt0, mt0 + 1: (d, h) = (mt / 4, mt % 4)
    float sc[NT][4], bi[NT][4], fl[NT][4];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = m * 16 + g * 4 + k;
            sc[m][k] = a.scale ? a.scale[c] : 1.0f;
            bi[m][k] = a.bias ? a.bias[c] : 0.0f;
            fl[m][k] = (a.epi & PSCV_EPI_RELU_PRE) ? (a.floor ? a.floor[c] : 0.0f) : -__builtin_inff();
        }
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    const int ow = t0w + n;
    const bool col_ok = ow < a.W;
    uint2 skv[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int od = t0d + (mt0 + i) / WD_TH, oh = t0h + (mt0 + i) % WD_TH;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            skv[i][m] = make_uint2(0u, 0u);
            if (a.skip && col_ok && od < a.D && oh < a.Hh)
                skv[i][m] = *reinterpret_cast<const uint2*>(a.skip + ((((long)b * a.D + od) * a.Hh + oh) * a.W + ow) * a.skip_cs + a.skip_co + m * 16 + g * 4);
        }
    }
    __syncthreads();

    // ---- contraction: k-step st = stage * S + ks covers k = 32 st .. 32 st + 31 of (tap, c_in) -> tap = st / (CIN / 32), channel block st % (CIN / 32) ----
    int anchor[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = mt0 + i;
        anchor[i] = (((mt / WD_TH) * WD_BH + (mt % WD_TH)) * WD_BW + n) * VS + g * 16;
    }
    wd_f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[i][m] = wd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        uint4 wn[WLD];
        if (s + 1 < NSTAGE) {
#pragma unroll
            for (int r = 0; r < WLD; ++r) {
                const int idx = tid + 512 * r;
                wn[r] = idx < SB / 16 ? a.wpk[(s + 1) * (SB / 16) + idx] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        const unsigned char* wb = wbuf + (s & 1) * SB + lane * 16;
#pragma unroll
        for (int ks = 0; ks < S; ++ks) {
            constexpr int BPT = CIN / 32;                        // k-steps per tap
            const int st = s * S + ks;
            const int tap = st / BPT, cb = st % BPT;
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int koff = ((kd * WD_BH + kh) * WD_BW + kw) * VS + cb * 64;
            uint4 af[NT];
#pragma unroll
            for (int m = 0; m < NT; ++m) af[m] = *reinterpret_cast<const uint4*>(wb + (ks * NT + m) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 xf = *reinterpret_cast<const uint4*>(smem + anchor[i] + koff);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[i][m] = WdMfma<H>::run(af[m], xf, acc[i][m]);
            }
        }
        if (s + 1 < NSTAGE) {
#pragma unroll
            for (int r = 0; r < WLD; ++r) {
                const int idx = tid + 512 * r;
                if (idx < SB / 16) *reinterpret_cast<uint4*>(wbuf + ((s + 1) & 1) * SB + idx * 16) = wn[r];
            }
            __syncthreads();       // stage s + 1 is in place; nobody reads buffer s & 1 any more when stage s + 2 overwrites it
        }
    }

    // ---- epilogue: lane (n, g) owns channels 16 m + 4 g .. + 3 of voxel (row, t0w + n); same operation chain as conv3d.hip ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int od = t0d + (mt0 + i) / WD_TH, oh = t0h + (mt0 + i) % WD_TH;
        if (od >= a.D || oh >= a.Hh || !col_ok) continue;
        const long vox = (((long)b * a.D + od) * a.Hh + oh) * a.W + ow;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            float y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = relu_floor(fmaf(acc[i][m][k], sc[m][k], bi[m][k]), fl[m][k]);
            // (always added, +0 without a skip tensor: the brick kernel's chain, down to the sign of a zero)
            y[0] = relu_floor(y[0] + Half16<H>::lo(skv[i][m].x), lo_post); y[1] = relu_floor(y[1] + Half16<H>::hi(skv[i][m].x), lo_post);
            y[2] = relu_floor(y[2] + Half16<H>::lo(skv[i][m].y), lo_post); y[3] = relu_floor(y[3] + Half16<H>::hi(skv[i][m].y), lo_post);
            if (a.out_f32)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + vox * a.out_cs + a.out_co + m * 16 + g * 4) = make_float4(y[0], y[1], y[2], y[3]);
            else
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + vox * a.out_cs + a.out_co + m * 16 + g * 4) =
                    make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
        }
    }
}

// ---- 64 input channels, reduction split over the two wave halves ("conv_wide" = 3; measured, NOT the default) -----------------------
// Hypothesis tested in round 5: the kernel above is bound by LDS reads (two rows per wave: NT A reads + 2 B reads per 2 NT MFMAs =
// 0.75 KB per MFMA, 192 B/clk per CU at the MFMA peak against the 256 B/clk `ds_read_b128` delivers at best).  Result: a third less
// LDS traffic for the same MFMAs buys nothing -- 114.7 us against 105.9 us per 64 -> 64 launch of configuration 4 on the same box -- so
// LDS bandwidth is not the limit; both sit at 0.28-0.30 of the nominal MFMA peak (~0.5 of what the part sustains at its MFMA clock),
// like the 64-channel 2-D kernel with LDS-resident weights (0.35).  Kept as a measured variant.  Here waves 0-3 contract the EVEN k-steps
// (channels 0-31 of every tap) and waves 4-7 the ODD ones (channels 32-63), each over FOUR rows and all output tiles: NT A reads + 4
// B reads feed 4 NT MFMAs (0.5 KB per MFMA, a third less LDS traffic for the same MFMAs); the two partial sums of a row meet once,
// through LDS (each wave hands over the half of the output tiles it does not finish: 64 KB over the then idle brick), and every wave
// runs the epilogue of four rows x NT / 2 output tiles.  The sum is (even k-steps) + (odd k-steps): within fp32 rounding of the
// sequential order, not bit-identical to the brick kernel (tests: 1e-6 relative, and ATen).
template <typename H, int NT>
__global__ __launch_bounds__(512, 2) void conv3d_wide2_kernel(const WideArgs a) {
    constexpr int CIN = 64, VS = wd_vs(CIN), CCH = CIN / 8, S = 6, NSTEPS = 54, NSTAGE = 9, NH = NT / 2;
    constexpr int BRICK = (WD_NVOX * VS + 15) & ~15, SB = S * NT * 1024;
    constexpr int NCH = WD_NVOX * CCH, NLD = (NCH + 511) / 512;
    constexpr int WLD = (SB / 16 + 511) / 512;
    static_assert(NT % 2 == 0, "the output tiles split over the two wave halves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem + BRICK;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot;
    const int tw_i = fast_divmod(wg, a.ntw, a.mg_tw);
    const int th_i = fast_divmod(wg, a.nth, a.mg_th);
    const int td_i = fast_divmod(wg, a.ntd, a.mg_td);
    const int b = wg;
    const int t0d = td_i * WD_TD, t0h = th_i * WD_TH, t0w = tw_i * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int kpar = wave >> 2;                                 // 0: even k-steps, 1: odd k-steps
    const int od = t0d + (wave & 3);                            // this wave's four rows: plane od, rows t0h .. t0h + 3
    const int mb = kpar * NH;                                   // first output tile this wave finishes

    {
        uint4 val[NLD];
        const long plane = (long)a.Hh * a.W;
        const uint16_t* inb = a.in + (long)b * a.D * plane * a.in_cs + a.in_co;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            const int bd = v / (WD_BH * WD_BW), rem = v - bd * (WD_BH * WD_BW);
            const int bh = rem / WD_BW, bw = rem - bh * WD_BW;
            const int gd = t0d - 1 + bd, gh = t0h - 1 + bh, gw = t0w - 1 + bw;
            val[i] = make_uint4(0u, 0u, 0u, 0u);
            if (c < NCH && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W)
                val[i] = *reinterpret_cast<const uint4*>(inb + ((long)gd * plane + (long)gh * a.W + gw) * a.in_cs + cc * 8);
        }
        uint4 w0[WLD];
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            w0[r] = idx < SB / 16 ? a.wpk[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            if (c < NCH) *reinterpret_cast<uint4*>(smem + v * VS + cc * 16) = val[i];
        }
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            if (idx < SB / 16) *reinterpret_cast<uint4*>(wbuf + idx * 16) = w0[r];
        }
    }
    float sc[NH][4], bi[NH][4], fl[NH][4];
#pragma unroll
    for (int m = 0; m < NH; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (mb + m) * 16 + g * 4 + k;
            sc[m][k] = a.scale ? a.scale[c] : 1.0f;
            bi[m][k] = a.bias ? a.bias[c] : 0.0f;
            fl[m][k] = (a.epi & PSCV_EPI_RELU_PRE) ? (a.floor ? a.floor[c] : 0.0f) : -__builtin_inff();
        }
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    const int ow = t0w + n;
    const bool col_ok = ow < a.W;
    uint2 skv[4][NH];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oh = t0h + i;
#pragma unroll
        for (int m = 0; m < NH; ++m) {
            skv[i][m] = make_uint2(0u, 0u);
            if (a.skip && col_ok && od < a.D && oh < a.Hh)
                skv[i][m] = *reinterpret_cast<const uint2*>(a.skip + ((((long)b * a.D + od) * a.Hh + oh) * a.W + ow) * a.skip_cs + a.skip_co + (mb + m) * 16 + g * 4);
        }
    }
    __syncthreads();

    // k-step st = 2 (3 s + j) + kpar: tap 3 s + j (compile time), channel block kpar
    int anchor[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) anchor[i] = ((((wave & 3)) * WD_BH + i) * WD_BW + n) * VS + g * 16 + kpar * 64;
    wd_f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[i][m] = wd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        uint4 wn[WLD];
        if (s + 1 < NSTAGE) {
#pragma unroll
            for (int r = 0; r < WLD; ++r) {
                const int idx = tid + 512 * r;
                wn[r] = idx < SB / 16 ? a.wpk[(s + 1) * (SB / 16) + idx] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        const unsigned char* wb = wbuf + (s & 1) * SB + lane * 16 + kpar * (NT * 1024);
#pragma unroll
        for (int j = 0; j < S / 2; ++j) {
            const int tap = s * (S / 2) + j;
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int koff = ((kd * WD_BH + kh) * WD_BW + kw) * VS;
            uint4 af[NT];
#pragma unroll
            for (int m = 0; m < NT; ++m) af[m] = *reinterpret_cast<const uint4*>(wb + (2 * j * NT + m) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 xf = *reinterpret_cast<const uint4*>(smem + anchor[i] + koff);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[i][m] = WdMfma<H>::run(af[m], xf, acc[i][m]);
            }
        }
        if (s + 1 < NSTAGE) {
#pragma unroll
            for (int r = 0; r < WLD; ++r) {
                const int idx = tid + 512 * r;
                if (idx < SB / 16) *reinterpret_cast<uint4*>(wbuf + ((s + 1) & 1) * SB + idx * 16) = wn[r];
            }
            __syncthreads();
        }
    }
    // ---- the two halves of the reduction meet: every wave hands the output tiles it does not finish to its partner (wave ^ 4) ----
    __syncthreads();                                            // every wave is done with the brick
    wd_f32x4* xb = reinterpret_cast<wd_f32x4*>(smem);           // [wave][row 0..3][tile 0..NH-1][lane]
    const int give = (1 - kpar) * NH;                           // first tile handed over
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NH; ++m) xb[((wave * 4 + i) * NH + m) * 64 + lane] = kpar ? acc[i][m] : acc[i][NH + m];
    __syncthreads();
    wd_f32x4 fin[4][NH];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NH; ++m) {
            const wd_f32x4 other = xb[(((wave ^ 4) * 4 + i) * NH + m) * 64 + lane];
            const wd_f32x4 mine = kpar ? acc[i][NH + m] : acc[i][m];
            fin[i][m] = kpar ? other + mine : mine + other;        // (even k-steps) + (odd k-steps)
        }
    (void)give;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oh = t0h + i;
        if (od >= a.D || oh >= a.Hh || !col_ok) continue;
        const long vox = (((long)b * a.D + od) * a.Hh + oh) * a.W + ow;
#pragma unroll
        for (int m = 0; m < NH; ++m) {
            float y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = relu_floor(fmaf(fin[i][m][k], sc[m][k], bi[m][k]), fl[m][k]);
            y[0] = relu_floor(y[0] + Half16<H>::lo(skv[i][m].x), lo_post); y[1] = relu_floor(y[1] + Half16<H>::hi(skv[i][m].x), lo_post);
            y[2] = relu_floor(y[2] + Half16<H>::lo(skv[i][m].y), lo_post); y[3] = relu_floor(y[3] + Half16<H>::hi(skv[i][m].y), lo_post);
            if (a.out_f32)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + vox * a.out_cs + a.out_co + (mb + m) * 16 + g * 4) = make_float4(y[0], y[1], y[2], y[3]);
            else
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + vox * a.out_cs + a.out_co + (mb + m) * 16 + g * 4) =
                    make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
        }
    }
}

template <typename H, int NT>
static int wide2_launch(const WideArgs& a, long nblk, hipStream_t st) {
    constexpr int LDS = wd_lds(64, NT);
    auto kern = conv3d_wide2_kernel<H, NT>;
    hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS);
    if (e != hipSuccess) { set_error("pscv_conv3d(wide2): hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), LDS, st, a);
    return 0;
}

template <typename H, int CIN, int NT>
static int wide_launch(const WideArgs& a, long nblk, hipStream_t st) {
    constexpr int LDS = wd_lds(CIN, NT);
    static_assert(LDS <= 160 * 1024, "brick + weight double buffer do not fit the LDS");
    auto kern = conv3d_wide_kernel<H, CIN, NT>;
    hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS);
    if (e != hipSuccess) { set_error("pscv_conv3d(wide): hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), LDS, st, a);
    return 0;
}

}  // namespace pscv

// Returns 0 if launched, 1 if the layer / size is not covered (the caller runs the brick kernel), negative on error.
int pscv_conv3d_wide_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                            const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                            int out_cstride, int out_coff, int out_dtype, int B, int D, int Hh, int W, int c_in, int c_out, int epi_flags,
                            hipStream_t st) {
    using namespace pscv;
    if (!g_conv_wide) return 1;
    if (!((c_in == 32 || c_in == 64) && (c_out == 32 || c_out == 64))) return 1;
    // measured at CVP's configuration 4 (scripts/dev/config_kernels.py 4): 64 -> 64 128 -> 110 us, 64 -> 32 91 -> 75 us per launch;
    // the 32-input layers (27 k-steps only) are no faster than on the brick kernel (32 -> 32 38 vs 35..43 us, 32 -> 64 62 vs 68 us):
    // they take this kernel only when it is forced ("conv_wide" = 2: tests)
    if (c_in == 32 && g_conv_wide < 2) return 1;
    if ((out_cstride | out_coff) & 3 || (skip && ((skip_cstride | skip_coff) & 3))) return 1;
    WideArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = reinterpret_cast<const uint4*>(packed);
    a.scale = scale; a.bias = bias; a.floor = floor;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff; a.out_cs = out_cstride; a.out_co = out_coff;
    a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.D = D; a.Hh = Hh; a.W = W; a.epi = epi_flags;
    a.ntd = (D + WD_TD - 1) / WD_TD; a.nth = (Hh + WD_TH - 1) / WD_TH; a.ntw = (W + 15) / 16;
    a.mg_td = fast_div_magic(a.ntd); a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw);
    const long nblk = (long)B * a.ntd * a.nth * a.ntw;
    // one 8-wave workgroup per CU: volumes with fewer tiles than CUs stay on the brick kernel's small tiles (more, lighter workgroups)
    if (nblk < (g_conv_wide >= 2 ? 1 : 512) || nblk > 0x7fffffffL) return 1;      // ("conv_wide" >= 2: at any size)
    const int nt = c_out / 16;
    if (c_in == 64 && g_conv_wide == 3) {        // ("conv_wide" = 3: the reduction split over the wave halves -- measured, not the default)
        if (dtype == PSCV_BF16) return nt == 4 ? wide2_launch<bf16_t, 4>(a, nblk, st) : wide2_launch<bf16_t, 2>(a, nblk, st);
        return nt == 4 ? wide2_launch<f16_t, 4>(a, nblk, st) : wide2_launch<f16_t, 2>(a, nblk, st);
    }
#define PSCV_WIDE_CASE(HT, CI, NTV) if (c_in == CI && nt == NTV) return wide_launch<HT, CI, NTV>(a, nblk, st);
    if (dtype == PSCV_BF16) { PSCV_WIDE_CASE(bf16_t, 64, 4) PSCV_WIDE_CASE(bf16_t, 64, 2) PSCV_WIDE_CASE(bf16_t, 32, 4) PSCV_WIDE_CASE(bf16_t, 32, 2) }
    else { PSCV_WIDE_CASE(f16_t, 64, 4) PSCV_WIDE_CASE(f16_t, 64, 2) PSCV_WIDE_CASE(f16_t, 32, 4) PSCV_WIDE_CASE(f16_t, 32, 2) }
#undef PSCV_WIDE_CASE
    return 1;
}
