// 3x3x3 stride-1 convolution for WIDE layers (32 | 64 input channels -> 32 | 64 output channels) on large volumes: CVP-MVSNet's
// conv2 / conv2a (32 -> 32), conv3 (32 -> 64), conv4 / conv4a (64 -> 64) and conv5 (64 -> 32, stride-1 transposed = a stride-1
// convolution with the flipped kernel).  gfx950.
//
// The brick kernel (conv3d.hip) runs these layers with 256-thread workgroups whose waves each stream ALL A fragments of the layer
// from global memory (216 KB per wave at 64 -> 64, 864 KB per 256 output voxels): with a 93 KB input brick only ONE such workgroup
// fits a CU, i.e. one wave per SIMD whose MFMAs wait on its own weight loads -- 0.25 of the MFMA peak, 0.09 of HBM, bound by the
// L2 -> CU path (4.4 GB of weight re-reads per launch at 4 x 512 x 640).  Here (conv3d_widep_kernel):
//
//   workgroup = 512 threads = 8 waves on the same 4 x 4 x 16 output tile (16 rows of 16 x-adjacent voxels, two per wave, all
//     output channels): two waves per SIMD; ONE workgroup per CU, persistent over its share of the tiles;
//   the weights pass through an LDS double buffer shared by the eight waves: a stage = 6 | 3 (C_in 64) or 7 | 6 (C_in 32) k-steps of
//     all output tiles = one contiguous piece of the packed weights, copied cooperatively global -> registers -> LDS two stages ahead
//     of its use, one barrier per stage: the weights enter the CU once per tile (216 KB at 64 -> 64), the A fragments are
//     ds_read_b128 like the B fragments;
//   per k-step and wave: NT A reads + 2 B reads feed 2 NT MFMAs, the reads of k-step s + 1 issued before the MFMAs of k-step s;
//     every accumulator sums its k-steps in ascending order like the brick kernel's, and the epilogue is the same operation chain:
//     same stored bits (tests/test_gpu_conv3d.py).
//
// Measured (4 x 512 x 640, 64 -> 64, fp16; profiles/r05_wide_kernel.txt): brick kernel 385-390 us, one tile per workgroup 381-415 us,
// persistent 345-365 us (0.32-0.34 of the nominal MFMA peak); the counters and ablations that led here are in the same file.
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

PSCV_PROF_BUFFER(wide)
Knob g_conv_wide = {1, KNOB_CONV_WIDE};       // pscv_set_tuning("conv_wide", 0): these layers back on the brick kernel (A/B runs, bit comparison)

// ---- the kernel ------------------------------------------------------------------------------------------------------------------
// One workgroup per CU walks its share of the tiles.  With one tile per workgroup (this file's first form, round 5) the phase stamps
// showed 28 % of a workgroup's life requesting and awaiting its 83 KB brick and 13 % in the epilogue and drain, with nothing else
// resident on the CU to fill either (one workgroup's LDS is the CU's).  Here
//   the NEXT tile's brick is requested (buffer loads, out-of-volume chunks read as zeros) when the contraction of the current tile
//     starts and sits in registers (11 x 16 B per thread at 64 channels) until the last k-step, where -- the barrier that hands the
//     last weight stage over also being the last read of the brick -- it is written over the brick while the last MFMAs run;
//   the weight stages keep cycling: the stage count is even (6 6 6 6 6 6 6 6 3 3 k-steps at 64 channels, 7 7 7 6 at 32), so the
//     double buffer's parity is the same for every tile and stage 0 of the next tile is staged during the last stage of this one;
//   scale / bias / floor sit in LDS instead of 48 registers;
//   16-bit outputs leave through a per-wave LDS row as whole-voxel 16-byte stores.
// Ablations on the 345 us launch (stores / brick traffic / weight staging removed one at a time): -44 / -42 / -33 us -- what is left
// of the three non-MFMA streams of a tile (32 KB out, 83 KB in, 216 KB of weights through VGPRs into LDS); the LDS array is
// co-critical with the MFMA pipe (10.4 k cycles of fragment reads + 3.9 k of stage and brick writes against 13.8 k MFMA cycles per
// tile and SIMD), which is why fewer reads per MFMA alone (the reduction split over the wave halves tried in round 5: DESIGN.md) bought nothing.
__host__ __device__ constexpr int wp_nstage(int cin) { return cin == 64 ? 10 : 4; }
__host__ __device__ constexpr int wp_slen(int cin, int s) { return cin == 64 ? (s < 8 ? 6 : 3) : (s < 3 ? 7 : 6); }
__host__ __device__ constexpr int wp_sbase(int cin, int s) { return cin == 64 ? (s < 8 ? 6 * s : 48 + 3 * (s - 8)) : 7 * s; }
__host__ __device__ constexpr int wp_smax(int cin) { return cin == 64 ? 6 : 7; }
// brick voxels: 64 channels 128 B apart with the eight 16-byte chunks of voxel v at chunk ^ (v & 7) (conv2d.hip, C2WGeom: the sixteen
// granules of a `ds_read_b128` lane group are all different, without padding); 32 channels at the 96-byte stride of conv_vs
__host__ __device__ constexpr int wp_vs(int cin) { return cin == 64 ? 128 : 96; }
__host__ __device__ constexpr int wp_opitch(int nt) { return nt * 32 + 16; }     // bytes per voxel of a wave's output staging row (+16: conflict-free 8-byte writes)
__host__ __device__ constexpr int wp_lds(int cin, int nt) { return WD_NVOX * wp_vs(cin) + 2 * wp_smax(cin) * nt * 1024 + 3 * nt * 64 + 8 * 16 * wp_opitch(nt); }

template <typename H, int CIN, int NT>
__global__ __launch_bounds__(512, 2) void conv3d_widep_kernel(const WideArgs a, const int ntiles) {
    constexpr int VS = wp_vs(CIN), CCH = CIN / 8, NSTEPS = 27 * CIN / 32, NSTAGE = wp_nstage(CIN), SMAX = wp_smax(CIN);
    constexpr int BRICK = WD_NVOX * VS, SB = SMAX * NT * 1024, OPITCH = wp_opitch(NT);   // bytes: input brick, one weight stage buffer, staged output voxel
    constexpr int NCH = WD_NVOX * CCH, NLD = (NCH + 511) / 512;                    // brick chunks (16 B), loads per thread
    constexpr int WLD = (SB / 16 + 511) / 512;                                     // weight-stage loads per thread
    constexpr int BPT = CIN / 32;                                                  // k-steps per tap
    static_assert(wp_sbase(CIN, NSTAGE - 1) + wp_slen(CIN, NSTAGE - 1) == NSTEPS && NSTAGE % 2 == 0, "an even number of stages tiles the k-steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem + BRICK;
    float* const epi = reinterpret_cast<float*>(smem + BRICK + 2 * SB);           // [scale | bias | floor][NT x 16]
    unsigned char* const so = smem + BRICK + 2 * SB + 3 * NT * 64 + (threadIdx.x >> 6) * (16 * OPITCH);   // this wave's output staging row
    auto lds_of = [](int v, int chunk) { return CIN == 64 ? v * VS + ((chunk ^ (v & 7)) << 4) : v * VS + chunk * 16; };
    PSCV_PROF_BEGIN   // (profile builds, summed over the tiles: tile start | - | - | contraction | epilogue | barrier)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    // ---- this workgroup's tiles: XCD x (= blockIdx & 7) owns a contiguous run of tiles, its workgroups walk it with their count as stride ----
    const int G = gridDim.x, bid = blockIdx.x;
    int t_cur, t_end, t_step;
    if ((G & 7) == 0) {
        const int xcd = bid & 7, q = ntiles >> 3, r_ = ntiles & 7;
        const int base = xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q;
        t_step = G >> 3; t_cur = base + (bid >> 3); t_end = base + q + (xcd < r_ ? 1 : 0);
    } else { t_step = G; t_cur = bid; t_end = ntiles; }
    if (t_cur >= t_end) return;

    const unsigned row_b = (unsigned)a.W * a.in_cs * 2, plane_b = (unsigned)a.Hh * row_b, vol_b = (unsigned)a.D * plane_b;   // (< 2 GiB: host)
    struct Tile { int b, t0d, t0h, t0w; };
    auto tile_of = [&](int t) {
        Tile r;
        r.t0w = fast_divmod(t, a.ntw, a.mg_tw) * 16;
        r.t0h = fast_divmod(t, a.nth, a.mg_th) * WD_TH;
        r.t0d = fast_divmod(t, a.ntd, a.mg_td) * WD_TD;
        r.b = t;
        return r;
    };
    // the brick of a tile -> registers: a chunk outside the volume (the convolution's padding) or beyond the brick takes an offset
    // outside the descriptor and reads as zeros
    // (the chunk decode is recomputed per tile from an opaque copy of the thread index: hoisted out of the tile loop it would hold
    //  ~40 registers through the contraction)
    auto bfetch = [&](const Tile& T, uint4 (&val)[NLD]) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.in) + ((unsigned long)T.b * vol_b + (unsigned long)a.in_co * 2)), (short)0,
            (int)(vol_b - (unsigned)a.in_co * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid_ + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            const int bd = v / (WD_BH * WD_BW), rem = v - bd * (WD_BH * WD_BW);
            const int bh = rem / WD_BW, bw = rem - bh * WD_BW;
            const int gd = T.t0d - 1 + bd, gh = T.t0h - 1 + bh, gw = T.t0w - 1 + bw;
            const bool ok = c < NCH && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)gd * plane_b + (unsigned)gh * row_b + ((unsigned)gw * a.in_cs + cc * 8) * 2u : 0x7ffffff0u;
            val[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        }
    };
    auto bstash = [&](const uint4 (&val)[NLD]) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid_ + 512 * i;
            const int v = c / CCH, cc = c - v * CCH;
            if (c < NCH) *reinterpret_cast<uint4*>(smem + lds_of(v, cc)) = val[i];
        }
    };
    uint4 wn[2][WLD];
    // (weights through a buffer descriptor: the stage's base is the scalar offset, a thread's chunk one 32-bit register)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wpk), (short)0, NSTEPS * NT * 1024, 0x00020000);
    auto wfetch = [&](const int stage, const int buf) {
        const int cnt = wp_slen(CIN, stage) * NT * 64, base = wp_sbase(CIN, stage) * NT * 64;
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            if (512 * r + 511 < cnt) wn[buf][r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, idx * 16, base * 16, 0));
            else if (512 * r < cnt) wn[buf][r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, idx < cnt ? idx * 16 : 0x7ffffff0, base * 16, 0));
        }
    };
    auto wstash = [&](const int stage, const int buf) {
        const int cnt = wp_slen(CIN, stage) * NT * 64;
#pragma unroll
        for (int r = 0; r < WLD; ++r) {
            const int idx = tid + 512 * r;
            if (idx < cnt) *reinterpret_cast<uint4*>(wbuf + (stage & 1) * SB + idx * 16) = wn[buf][r];
        }
    };

    // ---- prologue: first brick, weight stage 0 (stage 1 stays in registers), epilogue constants ----
    Tile T = tile_of(t_cur);
    uint4 val[NLD];
    bfetch(T, val);
    wfetch(0, 0);
    wfetch(1, 1);
    if (tid < NT * 16) {
        epi[tid] = a.scale ? a.scale[tid] : 1.0f;
        epi[NT * 16 + tid] = a.bias ? a.bias[tid] : 0.0f;
        epi[2 * NT * 16 + tid] = (a.epi & PSCV_EPI_RELU_PRE) ? (a.floor ? a.floor[tid] : 0.0f) : -__builtin_inff();
    }
    bstash(val);
    wstash(0, 0);
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    const int mt0 = 2 * wave;                                   // this wave's rows (M-tiles) mt0, mt0 + 1: (d, h) = (mt / 4, mt % 4)
    // B-fragment addresses.  32 channels: anchor + constant.  64 channels (swizzled): voxel v = anchor voxel + tap voxel, chunk 4 cb + g at
    // (chunk ^ (v & 7)) -- (v & 7) = (anchor & 7) + (tap voxel & 7) mod 8, so a lane keeps one byte offset per residue r of the tap voxel
    // (anchor * 128 + the swizzled chunk of channel block 0; block 1 flips bit 6) and the tap voxel itself goes into the instruction's
    // offset field.
    constexpr int NRES = CIN == 64 ? 8 : 1;
    int anchor[2][NRES];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = mt0 + i;
        const int av = ((mt / WD_TH) * WD_BH + (mt % WD_TH)) * WD_BW + n;
#pragma unroll
        for (int r = 0; r < NRES; ++r) anchor[i][r] = CIN == 64 ? av * VS + ((g ^ ((av + r) & 7)) << 4) : av * VS + g * 16;
    }
    uint4 af[2][NT], xf[2][2];
    auto issue = [&](const int st, const int buf) {
        int s = 0;
#pragma unroll
        for (int j = 1; j < NSTAGE; ++j) s += st >= wp_sbase(CIN, j) ? 1 : 0;
        const int ks = st - wp_sbase(CIN, s);
        const int tap = st / BPT, cb = st % BPT;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int kvox = (kd * WD_BH + kh) * WD_BW + kw;
        const unsigned char* wb = wbuf + (s & 1) * SB + lane * 16;
#pragma unroll
        for (int m = 0; m < NT; ++m) af[buf][m] = *reinterpret_cast<const uint4*>(wb + (ks * NT + m) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            xf[buf][i] = *reinterpret_cast<const uint4*>(smem + (CIN == 64 ? (anchor[i][kvox & 7] ^ (cb * 64)) : anchor[i][0]) + kvox * VS);
    };
    __syncthreads();
    int t_next = t_cur + t_step;
    Tile Tn = T;
    if (t_next < t_end) { Tn = tile_of(t_next); bfetch(Tn, val); }
    issue(0, 0);
    PSCV_STAMP(0)

    for (;;) {
        const bool have_next = t_next < t_end;                  // (workgroup-uniform)
        wd_f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[i][m] = wd_f32x4{0.f, 0.f, 0.f, 0.f};
        const int ow = T.t0w + n;
        const bool col_ok = ow < a.W;
        uint2 skv[2][NT];
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            wfetch((s + 2) % NSTAGE, s & 1);                     // two stages ahead (the next tile's stages 0 / 1 from the last two)
            if (s == NSTAGE - 1) {
                // skip values of this wave's two rows, due in the epilogue
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int od = T.t0d + (mt0 + i) / WD_TH, oh = T.t0h + (mt0 + i) % WD_TH;
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        skv[i][m] = make_uint2(0u, 0u);
                        if (a.skip && col_ok && od < a.D && oh < a.Hh)
                            skv[i][m] = *reinterpret_cast<const uint2*>(a.skip + ((((long)T.b * a.D + od) * a.Hh + oh) * a.W + ow) * a.skip_cs + a.skip_co + m * 16 + g * 4);
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < wp_slen(CIN, s); ++ks) {
                const int st = wp_sbase(CIN, s) + ks;
                if (ks == wp_slen(CIN, s) - 1) {
                    wstash((s + 1) % NSTAGE, (s + 1) & 1);
                    __syncthreads();   // stage s + 1 is in place (its buffer was last read before the previous barrier); at the last stage: nobody reads the brick any more
                }
                if (st + 1 < NSTEPS) issue(st + 1, (st + 1) & 1);
                else if (have_next) bstash(val);                 // the next tile's brick, under the last MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < NT; ++m) acc[i][m] = WdMfma<H>::run(af[st & 1][m], xf[st & 1][i], acc[i][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PSCV_STAMP(3)
        // ---- epilogue: lane (n, g) owns channels 16 m + 4 g .. + 3 of voxel (row, t0w + n); same operation chain as conv3d.hip ----
        // 16-bit outputs leave through a per-wave LDS row as 16-byte stores (a lane: 8 consecutive channels of a voxel, whole 64 / 128-byte
        // voxels per 4 / 8 lanes) -- 8-byte stores per (row, channel tile) are store-issue bound (conv2d.hip, the same remedy); same bits
        const bool staged = !a.out_f32 && !((a.out_cs | a.out_co) & 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int od = T.t0d + (mt0 + i) / WD_TH, oh = T.t0h + (mt0 + i) % WD_TH;
            if (od >= a.D || oh >= a.Hh) continue;              // (wave-uniform)
            const long rowvox = (((long)T.b * a.D + od) * a.Hh + oh) * a.W;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const float4 sc = *reinterpret_cast<const float4*>(epi + m * 16 + g * 4);
                const float4 bi = *reinterpret_cast<const float4*>(epi + NT * 16 + m * 16 + g * 4);
                const float4 fl = *reinterpret_cast<const float4*>(epi + 2 * NT * 16 + m * 16 + g * 4);
                float y[4] = {relu_floor(fmaf(acc[i][m][0], sc.x, bi.x), fl.x), relu_floor(fmaf(acc[i][m][1], sc.y, bi.y), fl.y),
                              relu_floor(fmaf(acc[i][m][2], sc.z, bi.z), fl.z), relu_floor(fmaf(acc[i][m][3], sc.w, bi.w), fl.w)};
                // (always added, +0 without a skip tensor: the brick kernel's chain, down to the sign of a zero)
                y[0] = relu_floor(y[0] + Half16<H>::lo(skv[i][m].x), lo_post); y[1] = relu_floor(y[1] + Half16<H>::hi(skv[i][m].x), lo_post);
                y[2] = relu_floor(y[2] + Half16<H>::lo(skv[i][m].y), lo_post); y[3] = relu_floor(y[3] + Half16<H>::hi(skv[i][m].y), lo_post);
                if (staged)
                    *reinterpret_cast<uint2*>(so + n * OPITCH + m * 32 + g * 8) = make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                else if (col_ok) {
                    if (a.out_f32)
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (rowvox + ow) * a.out_cs + a.out_co + m * 16 + g * 4) = make_float4(y[0], y[1], y[2], y[3]);
                    else
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + (rowvox + ow) * a.out_cs + a.out_co + m * 16 + g * 4) =
                            make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                }
            }
            if (staged) {
                __builtin_amdgcn_wave_barrier();                 // (LDS executes a wave's operations in order: the reads below see the row)
#pragma unroll
                for (int q0 = 0; q0 < 16 * NT * 2; q0 += 64) {
                    const int q = q0 + lane, vx = q / (NT * 2), c = q - vx * (NT * 2);
                    const uint4 v = *reinterpret_cast<const uint4*>(so + vx * OPITCH + c * 16);
                    if (T.t0w + vx < a.W)
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (rowvox + T.t0w + vx) * a.out_cs + a.out_co + c * 8) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        PSCV_STAMP(4)
        if (!have_next) break;
        __syncthreads();                                         // the next brick is in place
        T = Tn;
        t_next += t_step;
        if (t_next < t_end) { Tn = tile_of(t_next); bfetch(Tn, val); }
        issue(0, 0);
        PSCV_STAMP(5)
    }
    PSCV_PROF_END(wide, blockIdx.x)
}

template <typename H, int CIN, int NT>
static int widep_launch(const WideArgs& a, long nblk, hipStream_t st) {
    constexpr int LDS = wp_lds(CIN, NT);
    static_assert(LDS <= 160 * 1024, "brick + weight double buffer do not fit the LDS");
    auto kern = conv3d_widep_kernel<H, CIN, NT>;
    hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS);
    if (e != hipSuccess) { set_error("pscv_conv3d(wide): hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
    const int n_cu = device_cu_count();            // of the current device (several GPUs in one process: not a per-process constant)
    if (n_cu <= 0) { set_error("pscv_conv3d(wide): device query failed"); return -2; }
    // one workgroup per CU; a multiple of 8 where the device has that many (the XCD-contiguous walk), the kernel takes any grid
    const long per_dev = n_cu >= 8 ? (long)(n_cu & ~7) : (long)n_cu;
    const long grid = nblk < per_dev ? nblk : per_dev;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, st, a, (int)nblk);
    return 0;
}

}  // namespace pscv

PSCV_PROF_EXPORT(wide)

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
    // a batch item's input volume is addressed through a 32-bit buffer descriptor
    if ((long)D * Hh * W * in_cstride * 2 >= 0x7fffffffL) return 1;
#define PSCV_WIDE_CASE(HT, CI, NTV) if (c_in == CI && nt == NTV) return widep_launch<HT, CI, NTV>(a, nblk, st);
    if (dtype == PSCV_BF16) { PSCV_WIDE_CASE(bf16_t, 64, 4) PSCV_WIDE_CASE(bf16_t, 64, 2) PSCV_WIDE_CASE(bf16_t, 32, 4) PSCV_WIDE_CASE(bf16_t, 32, 2) }
    else { PSCV_WIDE_CASE(f16_t, 64, 4) PSCV_WIDE_CASE(f16_t, 64, 2) PSCV_WIDE_CASE(f16_t, 32, 4) PSCV_WIDE_CASE(f16_t, 32, 2) }
#undef PSCV_WIDE_CASE
    return 1;
}
