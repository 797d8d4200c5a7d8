// Backward of the fused plane-sweep warp + cost aggregation (gfx950): gradient of the loss with respect to the
// reference and source FEATURE MAPS given the gradient of the cost volume.
//
// Reference semantics (fdarmon/wild_deep_mvs): the sampling grid is built under torch.no_grad()
// (models/MVSNet/module.py:127, models/CVP_MVSNet/models/modules.py:83, models/VisMVSNet/homography.py has no grad
// path to the cameras in train.py either), so autograd reaches only `src_fea` through F.grid_sample (bilinear, zeros)
// and the features through the cost statistic (models/MVSNet/model.py:109-176).  This kernel is that backward, fused:
// the per-view warped volumes and their gradients (503 MB fp32 each at the headline size) never exist.
//
// Per voxel: pass 1 re-samples the sources to rebuild the statistic the forward kept only in registers (sum for the
// variance, the soft-min numerator / weights), pass 2 re-samples each source, forms d cost / d warped in registers and
// scatters it to the four taps of the fp32 gradient map with hardware float atomics (global_atomic_add_f32, no
// return); the reference-feature gradient is accumulated over the block's depth planes in registers first.
// Same block / lane mapping and XCD-banded block order as the forward (warp_cost.hip).
#include <limits.h>
#include <type_traits>

#include "warp_common.h"

namespace pscv {

struct WarpBwdArgs {
    WarpArgs w;                    // geometry, features (ref / src), depth planes; w.out unused
    const void* g;                 // gradient of the forward's output, same layout as that output
    long g_view_stride;
    float* dref;                   // [B,h,w,C] fp32, accumulated (caller zero-fills), or NULL
    float* dsrc[PSCV_MAX_SRC];     // [B,hs,ws,C] fp32 each, accumulated (caller zero-fills)
    float* dtemp;                  // [1] fp32, accumulated (SOFTMIN), or NULL
    int ntx, nty;                  // tiled kernel: reference-pixel tiles per row / column
};

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

template <typename TIn, typename TG, int C, int LPV, int GEOM, int COST>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const WarpBwdArgs A) {
    const WarpArgs& a = A.w;
    constexpr int CPL = C / LPV;
    constexpr int PPB = 256 / LPV;
    constexpr int PIXB = C * (int)sizeof(TIn);
    constexpr bool VAR = (COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP);

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int pb = wg / a.n_dchunks;
    const int dc = wg - pb * a.n_dchunks;
    const int b = pb / a.npb_batch;
    const int pbb = pb - b * a.npb_batch;

    const int tid = threadIdx.x;
    __shared__ float cam_lds[PSCV_MAX_SRC * PSCV_CAM_FLOATS];
    __shared__ float red_lds[4];
    for (int i = tid; i < a.n_src * PSCV_CAM_FLOATS; i += 256) {
        const int v = i / PSCV_CAM_FLOATS, k = i - v * PSCV_CAM_FLOATS;
        cam_lds[i] = a.cams[((long)v * a.B + b) * PSCV_CAM_FLOATS + k];
    }
    __syncthreads();

    const int hw = a.h * a.w;
    const int pl = tid / LPV;
    int pflat = pbb * PPB + pl;
    const bool active = pflat < hw;
    pflat = active ? pflat : hw - 1;
    const int choff = (tid % LPV) * CPL;
    const unsigned chb = (unsigned)choff * (unsigned)sizeof(TIn);
    const long pix = (long)b * hw + pflat;
    const int y = pflat / a.w;
    const int x = pflat - y * a.w;
    const float off = (GEOM == PSCV_GEOM_HOMOG) ? 0.5f : 0.0f;
    const float px = (float)x + off, py = (float)y + off;

    VecF<CPL> rf, gref;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { rf.v[j] = 0.0f; gref.v[j] = 0.0f; }
    if (COST != PSCV_COST_WARP_ONLY) rf = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(a.ref) + pix * C + choff);

    const int d0 = dc * a.ppd;
    const int d1 = min(a.D, d0 + a.ppd);
    const float N = (float)(a.n_src + 1);
    const float invN = 1.0f / N;
    const long img_elems = (long)b * a.hs * a.ws * C;
    const TG* gp = reinterpret_cast<const TG*>(A.g);
    float dtemp_acc = 0.0f;

    // bilinear taps of (voxel, view) in the general (border-aware) form: invalid taps carry weight 0
    auto sample = [&](int v, float dval, Taps& taps) -> VecF<CPL> {
        float ix, iy;
        sweep_index<GEOM>(cam_lds + v * PSCV_CAM_FLOATS, px, py, dval, a, ix, iy);
        const float x0f = floorf(ix), y0f = floorf(iy);
        make_taps<false, PIXB>(ix - x0f, iy - y0f, (int)x0f, (int)y0f, a.hs, a.ws, chb, taps);
        const char* img = reinterpret_cast<const char*>(a.src[v]) + img_elems * (long)sizeof(TIn);
        return blend_taps<TIn, CPL, false, PIXB>(img, taps);
    };
    auto scatter = [&](int v, const Taps& taps, const VecF<CPL>& gw) {
        if (!active) return;
        float* base = A.dsrc[v] + img_elems;
        float* p00 = base + taps.o00 / sizeof(TIn);
        float* p01 = base + taps.o01 / sizeof(TIn);
        float* p10 = base + taps.o10 / sizeof(TIn);
        float* p11 = base + taps.o11 / sizeof(TIn);
        if (taps.w00 != 0.0f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(p00 + j, gw.v[j] * taps.w00);
        }
        if (taps.w01 != 0.0f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(p01 + j, gw.v[j] * taps.w01);
        }
        if (taps.w10 != 0.0f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(p10 + j, gw.v[j] * taps.w10);
        }
        if (taps.w11 != 0.0f) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(p11 + j, gw.v[j] * taps.w11);
        }
    };
    auto lane_sum = [&](float part) -> float {   // sum over the LPV lanes that share a voxel
#pragma unroll
        for (int m = 1; m < LPV; m <<= 1) part += __shfl_xor(part, m, 64);
        return part;
    };

    for (int d = d0; d < d1; ++d) {
        const float dval = a.depth_per_pixel ? a.depth[(long)b * a.depth_bstride + (long)d * hw + pflat]
                                             : a.depth[(long)b * a.depth_bstride + d];
        const long vox = ((long)b * a.D + d) * hw + pflat;
        Taps taps;

        if (VAR) {
            // cost = S2 / N - S1^2 / N^2  ->  d cost / d f = (2 / N) (f - S1 / N)      model.py:134, net.py:148
            VecF<CPL> s1 = rf;
            for (int v = 0; v < a.n_src; ++v) {
                const VecF<CPL> wv = sample(v, dval, taps);
#pragma unroll
                for (int j = 0; j < CPL; ++j) s1.v[j] += wv.v[j];
            }
            VecF<CPL> G = load_chan<TG, CPL>(gp + vox * C + choff);
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                G.v[j] *= 2.0f * invN;
                s1.v[j] *= invN;
                gref.v[j] = fmaf(G.v[j], rf.v[j] - s1.v[j], gref.v[j]);
            }
            for (int v = 0; v < a.n_src; ++v) {
                const VecF<CPL> wv = sample(v, dval, taps);
                VecF<CPL> gw;
#pragma unroll
                for (int j = 0; j < CPL; ++j) gw.v[j] = G.v[j] * (wv.v[j] - s1.v[j]);
                scatter(v, taps, gw);
            }
        } else if (COST == PSCV_COST_SOFTMIN) {
            // cost_c = num_c / Z, num_c = sum_v e_v diff_vc, Z = sum_v e_v + 1e-6, e_v = exp(-temp S_v), S_v = sum_c diff_vc
            VecF<CPL> num;
#pragma unroll
            for (int j = 0; j < CPL; ++j) num.v[j] = 0.0f;
            float Z = 1e-6f;
            for (int v = 0; v < a.n_src; ++v) {
                const VecF<CPL> wv = sample(v, dval, taps);
                VecF<CPL> diff;
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) { const float t = rf.v[j] - wv.v[j]; diff.v[j] = t * t; part += diff.v[j]; }
                const float e = __expf(-a.temp * lane_sum(part));
                Z += e;
#pragma unroll
                for (int j = 0; j < CPL; ++j) num.v[j] = fmaf(e, diff.v[j], num.v[j]);
            }
            const float invZ = 1.0f / Z;
            const VecF<CPL> G = load_chan<TG, CPL>(gp + vox * C + choff);
            float gc = 0.0f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) gc = fmaf(G.v[j], num.v[j] * invZ, gc);
            gc = lane_sum(gc);
            for (int v = 0; v < a.n_src; ++v) {
                const VecF<CPL> wv = sample(v, dval, taps);
                VecF<CPL> t;
                float part = 0.0f, gd = 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    t.v[j] = rf.v[j] - wv.v[j];
                    const float df = t.v[j] * t.v[j];
                    part += df;
                    gd = fmaf(G.v[j], df, gd);
                }
                const float S = lane_sum(part);
                gd = lane_sum(gd);
                const float e = __expf(-a.temp * S);
                const float dLde = (gd - gc) * invZ;
                if (active && (tid % LPV) == 0) dtemp_acc -= dLde * S * e;
                VecF<CPL> gw;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const float coef = 2.0f * e * (G.v[j] * invZ - a.temp * dLde) * t.v[j];   // d L / d diff * d diff / d r
                    gref.v[j] += coef;
                    gw.v[j] = -coef;
                }
                scatter(v, taps, gw);
            }
        } else if constexpr (COST == PSCV_COST_GROUPCORR) {
            // corr_g = sum_{c in group g} ref_c warp_c                                    nn_utils.py:473-490
            static_assert(C == 32, "group-wise correlation backward: 32 channels -> 8 groups (Vis-MVSNet)");
            constexpr int GC = C / 4;
            for (int v = 0; v < a.n_src; ++v) {
                const VecF<CPL> wv = sample(v, dval, taps);
                const TG* gv = gp + (long)v * A.g_view_stride + vox * GC + choff / 4;
                VecF<CPL> gw;
                const VecF<8> G8 = load_chan<TG, 8>(gv - (choff / 4) % 8);   // aligned 8-group chunk holding this lane's groups
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const float Gg = G8.v[(choff / 4) % 8 + j / 4];
                    gw.v[j] = Gg * rf.v[j];
                    gref.v[j] = fmaf(Gg, wv.v[j], gref.v[j]);
                }
                scatter(v, taps, gw);
            }
        } else {   // WARP_ONLY: the gradient of the warped volume goes straight to the taps
            for (int v = 0; v < a.n_src; ++v) {
                (void)sample(v, dval, taps);
                const VecF<CPL> gw = load_chan<TG, CPL>(gp + (long)v * A.g_view_stride + vox * C + choff);
                scatter(v, taps, gw);
            }
        }
    }

    if (COST != PSCV_COST_WARP_ONLY && A.dref && active) {
        float* dr = A.dref + pix * C + choff;
#pragma unroll
        for (int j = 0; j < CPL; ++j) atomic_add_f32(dr + j, gref.v[j]);
    }
    if (COST == PSCV_COST_SOFTMIN && A.dtemp) {
        float s = dtemp_acc;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        if ((tid & 63) == 0) red_lds[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) atomic_add_f32(A.dtemp, (red_lds[0] + red_lds[1]) + (red_lds[2] + red_lds[3]));
    }
}


// ---- LDS-privatised variant (default) ------------------------------------------------------------------------------
// The direct kernel above issues one global float atomic per (voxel, view, tap, channel): 2.0 G of them at the headline
// size, 64 scattered 4-byte targets per wave instruction -- measured 91 ms on MI355X (22 G atomics/s), 85 % of a whole
// training step.  Here a workgroup owns a 16 x 8 tile of reference pixels x 4 depth planes; per source view it
//   1. finds the exact bounding box of the texels its samples touch (block-wide min / max of the sample positions),
//   2. accumulates the tap gradients into an LDS patch of that box in 32-bit FIXED POINT with ds_add_u32 (texel stride
//      C + 1 words: the lanes of a wave spread over the banks).  Measured on MI355X (scripts/ubench/lds_atomic_rate.hip):
//      ds_add_f32 retires one wave64 instruction per ~190 clk per CU whatever the address pattern (a first version with
//      float LDS atomics took 8.3 ms), ds_add_u32 one per ~4 clk.  The scale is a power of two chosen per (workgroup, view)
//      from a bound on |d cost / d warped| that needs no extra pass (max |upstream gradient| x twice the largest feature
//      value the lane sampled), with 9 bits of headroom for the at most 128 pixels x 4 planes that can hit one texel: 21
//      bits below that bound are kept, and the LDS sums are order-independent (bit-reproducible),
//   3. flushes the patch with global float atomics in texel-major order: consecutive lanes hit consecutive addresses, so a
//      wave instruction is two full 128-byte lines instead of 64 scattered words, and each texel of the box is touched
//      once per workgroup instead of once per tap.
// Taps that fall outside the (capacity-clipped) box take the direct global atomic, so any geometry stays correct.
// The statistic of the forward (variance: sum; soft-min: numerator / weights) and the upstream gradient of the four planes
// live in registers across the view loop.
constexpr int BT_TW = 16, BT_TH = 8, BT_PLN = 4;
// patch capacity in texels: ~68 KB of LDS either way (two workgroups per CU): 512 texels of 32 channels, 1024 of 16 (CVP-MVSNet's per-pixel
// hypotheses around a still noisy depth estimate spread a tile's samples over more texels than a plane sweep does)
template <int C> constexpr int bt_texels() { return C <= 16 ? 1024 : 512; }
// -DPSCV_ABLATE builds read measurement flags from pscv_set_tuning("fuse_c0", bits): 1 no global flush atomics, 2 no LDS atomics,
// 4 no phase-A re-sampling, 8 no phase-B sampling, 16 no box, 32 no flush scan, 64 no reference-gradient adds, 128 no patch zeroing,
// 256 no direct adds for taps outside the patch (results are wrong with any bit set; scripts/dev/wbwd_ablate.py)
#ifdef PSCV_ABLATE
#define BT_ABL(bit) (a.variant & (bit))
#else
#define BT_ABL(bit) false
#endif

template <typename TIn, typename TG, int C, int GEOM, int COST>
__global__ __launch_bounds__(256) void warp_bwd_tile_kernel(const WarpBwdArgs A) {
    const WarpArgs& a = A.w;
    constexpr int LPV = 2;
    constexpr int CPL = C / LPV;
    constexpr int PIXB = C * (int)sizeof(TIn);
    constexpr int TS = C + 1;                   // patch floats per texel
    constexpr int BT_TEXELS = bt_texels<C>();
    constexpr bool VAR = (COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP);

    extern __shared__ __attribute__((aligned(16))) int patch[];   // [BT_TEXELS][TS] fixed point
    __shared__ float cam_lds[PSCV_MAX_SRC * PSCV_CAM_FLOATS];
    __shared__ int mm[PSCV_MAX_SRC][5];   // (min x, min y, max x, max y, bits of the magnitude bound)
    __shared__ int mm2[4];                // box of one row group when the tile's box exceeds the patch (sub-passes)
    __shared__ float red_lds[4];

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int pb = wg / a.n_dchunks;
    const int dc = wg - pb * a.n_dchunks;
    const int b = pb / a.npb_batch;
    const int pt = pb - b * a.npb_batch;
    const int ty_i = pt / A.ntx, tx_i = pt - ty_i * A.ntx;

    const int tid = threadIdx.x;
    for (int i = tid; i < a.n_src * PSCV_CAM_FLOATS; i += 256) {
        const int v = i / PSCV_CAM_FLOATS, k = i - v * PSCV_CAM_FLOATS;
        cam_lds[i] = a.cams[((long)v * a.B + b) * PSCV_CAM_FLOATS + k];
    }
    if (tid < a.n_src * 4) mm[tid >> 2][tid & 3] = (tid & 2) ? INT_MIN : INT_MAX;
    if (tid < a.n_src) mm[tid][4] = 0;
    if (!BT_ABL(128)) for (int i = tid; i < BT_TEXELS * TS; i += 256) patch[i] = 0;
    __syncthreads();

    const int hw = a.h * a.w;
    const int pl = tid >> 1;
    int x = tx_i * BT_TW + (pl & (BT_TW - 1)), y = ty_i * BT_TH + pl / BT_TW;
    const bool active = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int pflat = y * a.w + x;
    const int choff = (tid & 1) * CPL;
    const unsigned chb = (unsigned)choff * (unsigned)sizeof(TIn);
    const long pix = (long)b * hw + pflat;
    const float off = (GEOM == PSCV_GEOM_HOMOG) ? 0.5f : 0.0f;
    const float px = (float)x + off, py = (float)y + off;

    VecF<CPL> rf, gref;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { rf.v[j] = 0.0f; gref.v[j] = 0.0f; }
    if (COST != PSCV_COST_WARP_ONLY) rf = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(a.ref) + pix * C + choff);

    const int d0 = dc * BT_PLN;
    const float N = (float)(a.n_src + 1);
    const float invN = 1.0f / N;
    const long img_elems = (long)b * a.hs * a.ws * C;
    const TG* gp = reinterpret_cast<const TG*>(A.g);
    float dtemp_acc = 0.0f;

    float dval[BT_PLN];
    bool dok[BT_PLN];
    long vox[BT_PLN];
#pragma unroll
    for (int i = 0; i < BT_PLN; ++i) {
        const int d = min(d0 + i, a.D - 1);
        dok[i] = active && (d0 + i) < a.D;
        dval[i] = a.depth_per_pixel ? a.depth[(long)b * a.depth_bstride + (long)d * hw + pflat] : a.depth[(long)b * a.depth_bstride + d];
        vox[i] = ((long)b * a.D + d) * hw + pflat;
    }

    auto sample = [&](int v, float dv, Taps& taps, int& x0, int& y0) -> VecF<CPL> {
        float ix, iy;
        sweep_index<GEOM>(cam_lds + v * PSCV_CAM_FLOATS, px, py, dv, a, ix, iy);
        const float x0f = floorf(ix), y0f = floorf(iy);
        x0 = (int)x0f; y0 = (int)y0f;
        make_taps<false, PIXB>(ix - x0f, iy - y0f, x0, y0, a.hs, a.ws, chb, taps);
        const char* img = reinterpret_cast<const char*>(a.src[v]) + img_elems * (long)sizeof(TIn);
        return blend_taps<TIn, CPL, false, PIXB>(img, taps);
    };
    auto lane_sum = [&](float part) -> float { return part + __shfl_xor(part, 1, 64); };

    // ---- phase A: rebuild the forward's statistic for the four planes; upstream gradient into registers ----
    VecF<CPL> st[BT_PLN];    // variance: S1 / N;  soft-min: unused
    VecF<CPL> G[BT_PLN];     // variance: G * 2 / N;  soft-min: G;  plain warp: the view's upstream gradient
    float invZ[BT_PLN], gcs[BT_PLN];
    float fmaxabs = 0.0f;    // largest |feature value| this lane holds or sampled: |warped - mean| <= 2 fmaxabs
#pragma unroll
    for (int j = 0; j < CPL; ++j) fmaxabs = fmaxf(fmaxabs, fabsf(rf.v[j]));
    if (VAR || COST == PSCV_COST_SOFTMIN) {
#pragma unroll
        for (int i = 0; i < BT_PLN; ++i) {
            Taps taps;
            int x0, y0;
            if (VAR) {
                VecF<CPL> s1 = rf;
                for (int v = 0; v < (BT_ABL(4) ? 0 : a.n_src); ++v) {
                    const VecF<CPL> wv = sample(v, dval[i], taps, x0, y0);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) { s1.v[j] += wv.v[j]; fmaxabs = fmaxf(fmaxabs, fabsf(wv.v[j])); }
                }
                G[i] = load_chan<TG, CPL>(gp + vox[i] * C + choff);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    G[i].v[j] = dok[i] ? G[i].v[j] * 2.0f * invN : 0.0f;
                    st[i].v[j] = s1.v[j] * invN;
                    gref.v[j] = fmaf(G[i].v[j], rf.v[j] - st[i].v[j], gref.v[j]);
                }
            } else {
                VecF<CPL> num;
#pragma unroll
                for (int j = 0; j < CPL; ++j) num.v[j] = 0.0f;
                float Z = 1e-6f;
                for (int v = 0; v < a.n_src; ++v) {
                    const VecF<CPL> wv = sample(v, dval[i], taps, x0, y0);
                    VecF<CPL> diff;
                    float part = 0.0f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const float t = rf.v[j] - wv.v[j];
                        diff.v[j] = t * t;
                        part += diff.v[j];
                        fmaxabs = fmaxf(fmaxabs, fabsf(wv.v[j]));
                    }
                    const float e = __expf(-a.temp * lane_sum(part));
                    Z += e;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) num.v[j] = fmaf(e, diff.v[j], num.v[j]);
                }
                invZ[i] = 1.0f / Z;
                G[i] = load_chan<TG, CPL>(gp + vox[i] * C + choff);
                float gc = 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    G[i].v[j] = dok[i] ? G[i].v[j] : 0.0f;
                    gc = fmaf(G[i].v[j], num.v[j] * invZ[i], gc);
                }
                gcs[i] = lane_sum(gc);
            }
        }
    }

    // soft-min: d L / d warped of plane slot i for a sampled view (and, when `acc`, the reference / temperature parts)
    auto softmin_gw = [&](int i, const VecF<CPL>& wv, VecF<CPL>& gw, bool acc) {
        VecF<CPL> t;
        float part = 0.0f, gd = 0.0f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            t.v[j] = rf.v[j] - wv.v[j];
            const float df = t.v[j] * t.v[j];
            part += df;
            gd = fmaf(G[i].v[j], df, gd);
        }
        const float S = lane_sum(part);
        gd = lane_sum(gd);
        const float e = __expf(-a.temp * S);
        const float dLde = (gd - gcs[i]) * invZ[i];
        if (acc && dok[i] && (tid & 1) == 0) dtemp_acc -= dLde * S * e;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const float coef = 2.0f * e * (G[i].v[j] * invZ[i] - a.temp * dLde) * t.v[j];   // d L / d diff * d diff / d ref
            if (acc) gref.v[j] += coef;
            gw.v[j] = -coef;
        }
    };

    // ---- phase B: per source view, scatter through the LDS patch ----
    for (int v = 0; v < a.n_src; ++v) {
        // upper bound of |d cost / d warped| over this lane's four planes (no sampling needed)
        float bnd = 0.0f;
        if (VAR) {
            float gmax = 0.0f;
#pragma unroll
            for (int i = 0; i < BT_PLN; ++i)
#pragma unroll
                for (int j = 0; j < CPL; ++j) gmax = fmaxf(gmax, fabsf(G[i].v[j]));
            bnd = gmax * 2.0f * fmaxabs;
        } else if (COST == PSCV_COST_SOFTMIN) {
            // the soft-min coefficient has no useful closed-form bound (it mixes G, the weights and the channel sums):
            // evaluate it once without scattering -- MVSNet-s is the small model, the extra sampling pass is cheap
#pragma unroll
            for (int i = 0; i < BT_PLN; ++i) {
                Taps taps;
                int x0, y0;
                const VecF<CPL> wv = sample(v, dval[i], taps, x0, y0);
                VecF<CPL> gw;
                softmin_gw(i, wv, gw, false);
#pragma unroll
                for (int j = 0; j < CPL; ++j) bnd = fmaxf(bnd, fabsf(gw.v[j]));
            }
        } else if constexpr (COST == PSCV_COST_GROUPCORR) {
#pragma unroll
            for (int i = 0; i < BT_PLN; ++i) {
                const VecF<8> G8 = load_chan<TG, 8>(gp + (long)v * A.g_view_stride + vox[i] * 8);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    G[i].v[j] = dok[i] ? G8.v[choff / 4 + j / 4] : 0.0f;
                    bnd = fmaxf(bnd, fabsf(G[i].v[j]) * fabsf(rf.v[j]));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BT_PLN; ++i) {
                G[i] = load_chan<TG, CPL>(gp + (long)v * A.g_view_stride + vox[i] * C + choff);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    G[i].v[j] = dok[i] ? G[i].v[j] : 0.0f;
                    bnd = fmaxf(bnd, fabsf(G[i].v[j]));
                }
            }
        }
        bnd = fminf(bnd, 3.0e38f);   // (an inf / NaN upstream gradient must not poison the scale; it reaches the output through the taps)
        int bbits = active ? (int)__float_as_uint(bnd) : 0;
        // exact bounding box of the texels this workgroup's samples of view v touch
        int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
#pragma unroll
        for (int i = 0; i < (BT_ABL(16) ? 0 : BT_PLN); ++i) {
            float ix, iy;
            sweep_index<GEOM>(cam_lds + v * PSCV_CAM_FLOATS, px, py, dval[i], a, ix, iy);
            const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
            if (dok[i] && x0 >= -1 && x0 < a.ws && y0 >= -1 && y0 < a.hs) {
                mnx = min(mnx, x0); mny = min(mny, y0); mxx = max(mxx, x0); mxy = max(mxy, y0);
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            mnx = min(mnx, __shfl_xor(mnx, m, 64)); mny = min(mny, __shfl_xor(mny, m, 64));
            mxx = max(mxx, __shfl_xor(mxx, m, 64)); mxy = max(mxy, __shfl_xor(mxy, m, 64));
            bbits = max(bbits, __shfl_xor(bbits, m, 64));   // non-negative floats order like their bit patterns
        }
        if ((tid & 63) == 0) {
            atomicMin(&mm[v][0], mnx); atomicMin(&mm[v][1], mny);
            atomicMax(&mm[v][2], mxx); atomicMax(&mm[v][3], mxy);
            atomicMax(&mm[v][4], bbits);
        }
        __syncthreads();   // also: the previous view's flush (which re-zeroes the patch) is complete
        int bx0 = max(mm[v][0], 0), by0 = max(mm[v][1], 0);
        int bx1 = (mm[v][2] == INT_MIN) ? -1 : min(mm[v][2] + 1, a.ws - 1);
        int by1 = (mm[v][3] == INT_MIN) ? -1 : min(mm[v][3] + 1, a.hs - 1);
        int bw = max(bx1 - bx0 + 1, 0), bh = max(by1 - by0 + 1, 0);
        // A box beyond the patch capacity (CVP-MVSNet's per-pixel hypotheses around a noisy depth estimate; strongly zoomed views):
        // instead of clipping it -- every tap outside takes scattered global atomics, 16 per lane and tap, ~20 G lanes/s -- the
        // tile's rows are processed in 2 or 4 sub-passes (row groups of 4 / 2), each with its own box, scatter and flush.
#ifndef BT_SUB_X10
#define BT_SUB_X10 10      // sub-passes from (BT_SUB_X10 / 10) x the capacity on: below that the clipped box + a few direct taps is cheaper
#endif
        const int area10 = bw * bh * 10;
        const int nsub = area10 <= BT_TEXELS * BT_SUB_X10 ? 1 : (area10 <= 2 * BT_TEXELS * BT_SUB_X10 ? 2 : 4);      // (workgroup-uniform: from mm[v])
        float* dsrc_v = A.dsrc[v] + img_elems;
        // fixed-point scale 2^(20 - exponent(bound)): |value * scale| < 2^21, and 2^21 * 512 adds stays inside int32
        const int bexp = ((mm[v][4] >> 23) & 255) - 127;
        const int sexp = min(max(20 - bexp, -100), 100);
        const float scale = __uint_as_float((unsigned)(127 + sexp) << 23);
        const float inv_scale = __uint_as_float((unsigned)(127 - sexp) << 23);

        // one scatter + flush pass over the lanes that are `mine` (ALL: every lane, the common case -- compiled without the per-lane
        // test, which would otherwise sit between the four planes' gathers and serialise their latencies: +70 % on the whole kernel)
        auto scatter_flush = [&](auto allc, const bool mine) {
        constexpr bool ALL = decltype(allc)::value;
        if (bw * bh > BT_TEXELS) {            // still beyond the capacity: clip, the rest goes straight to global memory
            if (bw > BT_TEXELS) bw = BT_TEXELS;
            bh = BT_TEXELS / bw;
        }
        auto put = [&](int tx, int ty, unsigned o, float wgt, const VecF<CPL>& gw) {
            if (wgt == 0.0f) return;
            const int lx = tx - bx0, ly = ty - by0;
            if ((unsigned)lx < (unsigned)bw && (unsigned)ly < (unsigned)bh) {
                int* p = patch + (ly * bw + lx) * TS + choff;
                const float ws_ = wgt * scale;
                if (BT_ABL(2)) return;
#pragma unroll
                for (int j = 0; j < CPL; ++j) atomicAdd(p + j, __float2int_rn(gw.v[j] * ws_));
            } else if (!BT_ABL(256)) {
                float* p = dsrc_v + o / sizeof(TIn);
#pragma unroll
                for (int j = 0; j < CPL; ++j) atomic_add_f32(p + j, gw.v[j] * wgt);
            }
        };

#pragma unroll
        for (int i = 0; i < BT_PLN; ++i) {
            Taps taps;
            int x0, y0;
            if (BT_ABL(8) || (!ALL && !mine)) continue;      // (a lane is `mine` in exactly one sub-pass: its sums below are taken once)
            const VecF<CPL> wv = sample(v, dval[i], taps, x0, y0);
            VecF<CPL> gw;
            if (VAR) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) gw.v[j] = G[i].v[j] * (wv.v[j] - st[i].v[j]);
            } else if (COST == PSCV_COST_SOFTMIN) {
                softmin_gw(i, wv, gw, true);
            } else if constexpr (COST == PSCV_COST_GROUPCORR) {
                static_assert(C == 32, "group-wise correlation backward: 32 channels -> 8 groups (Vis-MVSNet)");
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    gw.v[j] = G[i].v[j] * rf.v[j];
                    gref.v[j] = fmaf(G[i].v[j], wv.v[j], gref.v[j]);
                }
            } else {
                gw = G[i];
            }
            if (dok[i] && (ALL || mine)) {
                put(x0, y0, taps.o00, taps.w00, gw);
                put(x0 + 1, y0, taps.o01, taps.w01, gw);
                put(x0, y0 + 1, taps.o10, taps.w10, gw);
                put(x0 + 1, y0 + 1, taps.o11, taps.w11, gw);
            }
        }
        __syncthreads();
        // flush (texel-major: a wave writes whole 128-byte runs) and re-zero what was used
        const int nflush = BT_ABL(32) ? 0 : bw * bh * C;
        for (int idx = tid; idx < nflush; idx += 256) {
            const int texel = idx / C, c = idx - texel * C;
            const int ly = texel / bw, lx = texel - ly * bw;
            int* p = patch + texel * TS + c;
            const int qv = *p;
            if (qv != 0) {
                if (!BT_ABL(1)) atomic_add_f32(dsrc_v + ((long)(by0 + ly) * a.ws + bx0 + lx) * C + c, (float)qv * inv_scale);
                *p = 0;
            }
        }
        };
        if (nsub == 1) {
            scatter_flush(std::true_type{}, true);
        } else {
        for (int sub = 0; sub < nsub; ++sub) {
        const bool mine = ((pl / BT_TW) * nsub) / BT_TH == sub;
        {
            // this row group's own box (same steps as above, over its lanes only)
            if (tid < 4) mm2[tid] = (tid & 2) ? INT_MIN : INT_MAX;
            __syncthreads();          // also: the previous sub-pass' flush is complete
            int snx = INT_MAX, sny = INT_MAX, sxx = INT_MIN, sxy = INT_MIN;
#pragma unroll
            for (int i = 0; i < BT_PLN; ++i) {
                float ix, iy;
                sweep_index<GEOM>(cam_lds + v * PSCV_CAM_FLOATS, px, py, dval[i], a, ix, iy);
                const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
                if (mine && dok[i] && x0 >= -1 && x0 < a.ws && y0 >= -1 && y0 < a.hs) {
                    snx = min(snx, x0); sny = min(sny, y0); sxx = max(sxx, x0); sxy = max(sxy, y0);
                }
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                snx = min(snx, __shfl_xor(snx, m, 64)); sny = min(sny, __shfl_xor(sny, m, 64));
                sxx = max(sxx, __shfl_xor(sxx, m, 64)); sxy = max(sxy, __shfl_xor(sxy, m, 64));
            }
            if ((tid & 63) == 0) {
                atomicMin(&mm2[0], snx); atomicMin(&mm2[1], sny);
                atomicMax(&mm2[2], sxx); atomicMax(&mm2[3], sxy);
            }
            __syncthreads();
            bx0 = max(mm2[0], 0); by0 = max(mm2[1], 0);
            bx1 = (mm2[2] == INT_MIN) ? -1 : min(mm2[2] + 1, a.ws - 1);
            by1 = (mm2[3] == INT_MIN) ? -1 : min(mm2[3] + 1, a.hs - 1);
            bw = max(bx1 - bx0 + 1, 0); bh = max(by1 - by0 + 1, 0);
            __syncthreads();          // everybody has read mm2 before the next sub-pass resets it
        }
        scatter_flush(std::false_type{}, mine);
        }   // sub-passes
        }

    }

    // Reference-feature gradient of the tile: through LDS (the patch is free now) so that a wave's 64 lanes add to 64 CONSECUTIVE floats
    // (two pixels x 32 channels = two full 128-byte lines).  Issued straight from the registers -- lane = (pixel, channel half), 16
    // instructions with 64 lanes 64 bytes apart -- these 31 M adds (48 depth chunks meet on every pixel) ran at ~19 G lanes/s and
    // were 1.67 ms of the kernel's 2.58 ms at the headline size (ablation: scripts/dev/wbwd_ablate.py); coalesced they are ~0.1 ms.
    if (COST != PSCV_COST_WARP_ONLY && A.dref && !BT_ABL(64)) {
        __syncthreads();                                  // the last view's flush (which reads and re-zeroes the patch) is complete
        float* const stage = reinterpret_cast<float*>(patch);          // [BT_TW * BT_TH pixels][TS]: conflict-free both ways
#pragma unroll
        for (int j = 0; j < CPL; ++j) stage[pl * TS + choff + j] = active ? gref.v[j] : 0.0f;
        __syncthreads();
        float* const dref_b = A.dref + (long)b * hw * C;
        for (int idx = tid; idx < BT_TW * BT_TH * C; idx += 256) {
            const int p = idx / C, c = idx - p * C;
            const int gx = tx_i * BT_TW + (p & (BT_TW - 1)), gy = ty_i * BT_TH + p / BT_TW;
            const float v = stage[p * TS + c];
            if (gx < a.w && gy < a.h && v != 0.0f) atomic_add_f32(dref_b + ((long)gy * a.w + gx) * C + c, v);
        }
    }
    if (COST == PSCV_COST_SOFTMIN && A.dtemp) {
        float s = dtemp_acc;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        if ((tid & 63) == 0) red_lds[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) atomic_add_f32(A.dtemp, (red_lds[0] + red_lds[1]) + (red_lds[2] + red_lds[3]));
    }
}

Knob g_warp_bwd_direct = {0, KNOB_WARP_BWD_DIRECT};   // pscv_set_tuning("warp_bwd_direct", 1): the one-global-atomic-per-tap kernel (measurement / tests)

template <typename TIn, typename TG, int C>
static int bwd_tile_launch(WarpBwdArgs& A, int geom, int cost, hipStream_t st) {
    WarpArgs& a = A.w;
    A.ntx = (a.w + BT_TW - 1) / BT_TW;
    A.nty = (a.h + BT_TH - 1) / BT_TH;
    a.npb_batch = A.ntx * A.nty;
#ifdef PSCV_ABLATE
    { extern Knob g_fuse_c0; a.variant = g_fuse_c0; }
#endif
    a.ppd = BT_PLN;
    a.n_dchunks = (a.D + BT_PLN - 1) / BT_PLN;
    const long nblk = (long)a.npb_batch * a.B * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost_bwd: bad grid size %ld", nblk); return -1; }
    constexpr int LDS = bt_texels<C>() * (C + 1) * 4;
#define PSCV_BWDT(GEOMV, COSTV)                                                                                           \
    if (geom == GEOMV && cost == COSTV) {                                                                                 \
        auto kern = warp_bwd_tile_kernel<TIn, TG, C, GEOMV, COSTV>;                                                       \
        {                                                                                                 \
            hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS); \
            if (e != hipSuccess) { set_error("pscv_warp_cost_bwd: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; } \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), LDS, st, A);                                            \
        return 0;                                                                                                         \
    }
    PSCV_BWDT(PSCV_GEOM_PROJ, PSCV_COST_VARIANCE)
    PSCV_BWDT(PSCV_GEOM_PROJ, PSCV_COST_VARIANCE_CVP)
    PSCV_BWDT(PSCV_GEOM_PROJ, PSCV_COST_SOFTMIN)
    PSCV_BWDT(PSCV_GEOM_PROJ, PSCV_COST_WARP_ONLY)
    if constexpr (C == 32) {
        PSCV_BWDT(PSCV_GEOM_HOMOG, PSCV_COST_GROUPCORR)
    }
    PSCV_BWDT(PSCV_GEOM_HOMOG, PSCV_COST_WARP_ONLY)
#undef PSCV_BWDT
    set_error("pscv_warp_cost_bwd: cost mode %d is not available with geometry %d", cost, geom);
    return -1;
}

template <typename TIn, typename TG, int C, int LPV>
static int bwd_launch(WarpBwdArgs& A, int geom, int cost, hipStream_t st) {
    WarpArgs& a = A.w;
    constexpr int PPB = 256 / LPV;
    a.npb_batch = (a.h * a.w + PPB - 1) / PPB;
    const long n_pixblocks = (long)a.npb_batch * a.B;
    int ppd = 8;
    while (ppd > 1 && n_pixblocks * ((a.D + ppd - 1) / ppd) < 4096) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    const long nblk = n_pixblocks * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost_bwd: bad grid size %ld", nblk); return -1; }
#define PSCV_BWD(GEOMV, COSTV)                                                                                          \
    if (geom == GEOMV && cost == COSTV) {                                                                               \
        hipLaunchKernelGGL((warp_bwd_kernel<TIn, TG, C, LPV, GEOMV, COSTV>), dim3((unsigned)nblk), dim3(256), 0, st, A);  \
        return 0;                                                                                                       \
    }
    PSCV_BWD(PSCV_GEOM_PROJ, PSCV_COST_VARIANCE)
    PSCV_BWD(PSCV_GEOM_PROJ, PSCV_COST_VARIANCE_CVP)
    PSCV_BWD(PSCV_GEOM_PROJ, PSCV_COST_SOFTMIN)
    PSCV_BWD(PSCV_GEOM_PROJ, PSCV_COST_WARP_ONLY)
    if constexpr (C == 32) {
        PSCV_BWD(PSCV_GEOM_HOMOG, PSCV_COST_GROUPCORR)
    }
    PSCV_BWD(PSCV_GEOM_HOMOG, PSCV_COST_WARP_ONLY)
#undef PSCV_BWD
    set_error("pscv_warp_cost_bwd: cost mode %d is not available with geometry %d", cost, geom);
    return -1;
}

template <typename TIn, typename TG>
static int bwd_channels(WarpBwdArgs& A, int C, int geom, int cost, hipStream_t st) {
    if (!g_warp_bwd_direct) {
        if (C == 32) return bwd_tile_launch<TIn, TG, 32>(A, geom, cost, st);
        if (C == 16) return bwd_tile_launch<TIn, TG, 16>(A, geom, cost, st);
    }
    if (C == 32) return bwd_launch<TIn, TG, 32, 2>(A, geom, cost, st);
    if (C == 16) return bwd_launch<TIn, TG, 16, 2>(A, geom, cost, st);
    set_error("pscv_warp_cost_bwd: unsupported channel count C=%d (16 or 32)", C);
    return -1;
}

}  // namespace pscv

extern "C" int pscv_warp_cost_bwd(const void* ref, const void* const* srcs, int n_src, const float* cams, const float* depth,
                                  long depth_bstride, int depth_per_pixel, int geom, int cost, float temp, const void* grad_out,
                                  float* dref, float* const* dsrcs, float* dtemp, int B, int C, int h, int w, int hs, int ws,
                                  int D, int in_dtype, int grad_dtype, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(n_src >= 1 && n_src <= PSCV_MAX_SRC, "pscv_warp_cost_bwd: n_src=%d outside [1,%d]", n_src, PSCV_MAX_SRC);
    PSCV_CHECK_ARG(srcs && cams && depth && grad_out && dsrcs, "pscv_warp_cost_bwd: null pointer argument");
    PSCV_CHECK_ARG(cost == PSCV_COST_WARP_ONLY || ref, "pscv_warp_cost_bwd: ref is required for cost mode %d", cost);
    PSCV_CHECK_ARG(B > 0 && h > 0 && w > 0 && hs > 1 && ws > 1 && D > 0, "pscv_warp_cost_bwd: bad sizes");
    PSCV_CHECK_ARG((long)hs * ws < (1L << 24) && (long)hs * ws * C * 4 < (1L << 32), "pscv_warp_cost_bwd: source map %dx%dx%d too large", hs, ws, C);
    WarpBwdArgs A;
    WarpArgs& a = A.w;
    a.ref = ref;
    for (int i = 0; i < PSCV_MAX_SRC; ++i) { a.src[i] = i < n_src ? srcs[i] : nullptr; A.dsrc[i] = i < n_src ? dsrcs[i] : nullptr; }
    for (int i = 0; i < n_src; ++i) PSCV_CHECK_ARG(srcs[i] && dsrcs[i], "pscv_warp_cost_bwd: srcs[%d] / dsrcs[%d] is null", i, i);
    a.cams = cams; a.depth = depth; a.out = nullptr;
    a.depth_bstride = depth_bstride;
    a.n_src = n_src; a.B = B; a.h = h; a.w = w; a.hs = hs; a.ws = ws; a.D = D;
    a.depth_per_pixel = depth_per_pixel;
    a.temp = temp;
    const long vol = (long)B * D * h * w;
    a.out_view_stride = 0;
    A.g = grad_out;
    A.g_view_stride = cost == PSCV_COST_GROUPCORR ? vol * (C / 4) : vol * C;
    A.dref = dref; A.dtemp = dtemp;
    if (geom == PSCV_GEOM_PROJ) {
        a.sx = 1.0f; a.sy = 1.0f;
        a.xlo = -4.5f * (ws - 1); a.xhi = 5.5f * (ws - 1);
        a.ylo = -4.5f * (hs - 1); a.yhi = 5.5f * (hs - 1);
    } else {
        a.sx = (float)(ws - 1) / (float)ws; a.sy = (float)(hs - 1) / (float)hs;
        a.xlo = -0.05f * (ws - 1); a.xhi = 1.05f * (ws - 1);
        a.ylo = -0.05f * (hs - 1); a.yhi = 1.05f * (hs - 1);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (in_dtype == PSCV_BF16 && grad_dtype == PSCV_BF16) rc = bwd_channels<bf16_t, bf16_t>(A, C, geom, cost, st);
    else if (in_dtype == PSCV_F16 && grad_dtype == PSCV_F16) rc = bwd_channels<f16_t, f16_t>(A, C, geom, cost, st);
    else if (in_dtype == PSCV_F32 && grad_dtype == PSCV_F32) rc = bwd_channels<float, float>(A, C, geom, cost, st);
    else if (in_dtype == PSCV_BF16 && grad_dtype == PSCV_F32) rc = bwd_channels<bf16_t, float>(A, C, geom, cost, st);
    else if (in_dtype == PSCV_F16 && grad_dtype == PSCV_F32) rc = bwd_channels<f16_t, float>(A, C, geom, cost, st);
    else { set_error("pscv_warp_cost_bwd: unsupported dtype pair in=%d grad=%d", in_dtype, grad_dtype); return -1; }
    if (rc) return rc;
    PSCV_CHECK_LAUNCH("pscv_warp_cost_bwd");
    return 0;
}
