// Shared pieces of the plane-sweep warp kernels (warp_cost.hip: direct gather; warp_cost_tiled.hip: LDS-staged).
#pragma once
#include "pscv_common.h"

namespace pscv {

struct WarpArgs {
    const void* ref;
    const void* src[PSCV_MAX_SRC];
    const float* cams;   // [n_src][B][18]
    const float* depth;
    void* out;
    long depth_bstride;
    long out_view_stride;  // elements between per-source outputs (GROUPCORR / WARP_ONLY)
    int n_src, B, h, w, hs, ws, D;
    int depth_per_pixel;
    int ppd;             // depth planes per block
    int npb_batch;       // pixel blocks per batch item: ceil(h*w / PPB)
    int n_dchunks;       // ceil(D / ppd)
    int variant;         // measurement switch of the LDS-staged kernel (pscv_set_tuning("warp_tile")): 1 = the round-2 sweep loop
    float temp;
    float sx, sy;        // index scale: PROJ 1, HOMOG (W-1)/W
    float xlo, xhi, ylo, yhi;  // clamp of the pixel index implied by the reference's grid clamp
    int ref_y0;          // row of the FULL reference image the first row of this launch is (pscv_warp_cost_rows: a row slab computes with the
                         // pixel coordinates it has in the whole image); 0 for whole-image launches
    int* mode_hist;      // development aid of the LDS-staged kernel (pscv_debug_wl_mode_hist): [view][mode] counters of its per-(block, view) staging modes; null = off
};

template <int N> struct VecF { float v[N]; };

template <typename T, int CPL> __device__ __forceinline__ VecF<CPL> load_chan(const T* p) {
    VecF<CPL> r;
#pragma unroll
    for (int k = 0; k < CPL / 8; ++k) {
        const f32x8 t = Elem<T>::load8(p + 8 * k);
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[8 * k + j] = t.v[j];
    }
    return r;
}
template <typename T, int CPL> __device__ __forceinline__ void store_chan(T* p, const VecF<CPL>& r) {
#pragma unroll
    for (int k = 0; k < CPL / 8; ++k) {
        f32x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t.v[j] = r.v[8 * k + j];
        Elem<T>::store8(p + 8 * k, t);
    }
}

// Source-image pixel index (ix, iy) of reference pixel (px, py) on plane d for one source camera.
template <int GEOM>
__device__ __forceinline__ void sweep_index(const float* __restrict__ cam, float px, float py, float d,
                                            const WarpArgs& a, float& ix, float& iy) {
    float hx, hy, hz;
    if (GEOM == PSCV_GEOM_PROJ) {
        // q = rot * (x, y, 1) * d + trans                                  module.py:138-144
        const float rx = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ry = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float rz = fmaf(cam[7], py, cam[6] * px) + cam[8];
        hx = fmaf(rx, d, cam[9]);
        hy = fmaf(ry, d, cam[10]);
        hz = fmaf(rz, d, cam[11]);
    } else {
        // hom = A p - (Bm p) / (d + 1e-9)                                  homography.py:63-69
        const float ax = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ay = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float az = fmaf(cam[7], py, cam[6] * px) + cam[8];
        const float bx = fmaf(cam[10], py, cam[9] * px) + cam[11];
        const float by = fmaf(cam[13], py, cam[12] * px) + cam[14];
        const float bz = fmaf(cam[16], py, cam[15] * px) + cam[17];
        const float inv_d = __builtin_amdgcn_rcpf(d + 1e-9f);
        hx = fmaf(-bx, inv_d, ax);
        hy = fmaf(-by, inv_d, ay);
        hz = fmaf(-bz, inv_d, az);
    }
    // perspective divide; points at or behind the source camera go to (-10, -10)   module.py:146-150,
    // homography.py:113-117 (which also clamps the divisor at 1e-9)
    const bool front = hz > 0.0f;
    const float inv_z = __builtin_amdgcn_rcpf(GEOM == PSCV_GEOM_HOMOG ? fmaxf(hz, 1e-9f) : hz);
    float u = front ? hx * inv_z : -10.0f;
    float v = front ? hy * inv_z : -10.0f;
    // normalise -> clamp -> align_corners=True un-normalise collapses to a scaled, clamped index
    ix = fminf(fmaxf(u * a.sx, a.xlo), a.xhi);
    iy = fminf(fmaxf(v * a.sy, a.ylo), a.yhi);
}

// d = float(half of h2) * w + acc in one VALU instruction (op_sel_hi marks src0 as fp16, op_sel picks its high half)
__device__ __forceinline__ float fma_mix_lo(uint32_t h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(uint32_t h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}

// ---- direct-gather kernel helpers (warp_cost.hip) -----------------------------------------------------------------
// The sweep is VALU-issue bound (profiles/), so the per-(plane, view) instruction count is what these are shaped for.

// first blend term: float(half) * w  (fma with an inline 0.0 accumulator: bit-identical to the plain product)
__device__ __forceinline__ float mul_mix_lo(uint32_t h2, float w) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w));
    return d;
}
__device__ __forceinline__ float mul_mix_hi(uint32_t h2, float w) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w));
    return d;
}
__device__ __forceinline__ int med3_i32(int x, int lo, int hi) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(lo), "v"(hi));
    return d;
}

// Bilinear taps of one (voxel, view): byte offsets from the (wave-uniform) image base and the four weights.
struct Taps {
    unsigned o00, o01, o10, o11;
    float w00, w01, w10, w11;
};

// Pixel index -> taps.  INTERIOR: the caller has established (for the whole wave) that all four taps lie inside the
// image, so the validity masks, the index clamps and two of the offsets disappear; the arithmetic that remains is the
// same as in the general form, so both produce identical bits.
template <bool INTERIOR, int PIXB>
__device__ __forceinline__ void make_taps(float fx, float fy, int x0, int y0, int hs, int ws, unsigned chb, Taps& t) {
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    if (INTERIOR) {
        t.w00 = gx * gy; t.w01 = fx * gy; t.w10 = gx * fy; t.w11 = fx * fy;
        t.o00 = (__umul24((unsigned)y0, (unsigned)ws) + (unsigned)x0) * PIXB + chb;
        t.o10 = t.o00 + (unsigned)ws * PIXB;
        t.o01 = 0; t.o11 = 0;   // = o00 + PIXB, o10 + PIXB: folded into the load's immediate offset
    } else {
        const int x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = (unsigned)x0 < (unsigned)ws, vx1 = (unsigned)x1 < (unsigned)ws;
        const bool vy0 = (unsigned)y0 < (unsigned)hs, vy1 = (unsigned)y1 < (unsigned)hs;
        t.w00 = (vx0 && vy0) ? gx * gy : 0.0f;
        t.w01 = (vx1 && vy0) ? fx * gy : 0.0f;
        t.w10 = (vx0 && vy1) ? gx * fy : 0.0f;
        t.w11 = (vx1 && vy1) ? fx * fy : 0.0f;
        const unsigned xc0 = (unsigned)med3_i32(x0, 0, ws - 1), xc1 = (unsigned)med3_i32(x1, 0, ws - 1);
        const unsigned r0 = __umul24((unsigned)med3_i32(y0, 0, hs - 1), (unsigned)ws);
        const unsigned r1 = __umul24((unsigned)med3_i32(y1, 0, hs - 1), (unsigned)ws);
        t.o00 = (r0 + xc0) * PIXB + chb; t.o01 = (r0 + xc1) * PIXB + chb;
        t.o10 = (r1 + xc0) * PIXB + chb; t.o11 = (r1 + xc1) * PIXB + chb;
    }
}

// Zero-padded bilinear blend of this lane's CPL channels.  `base` is wave-uniform (SGPR pair) and the offsets are
// unsigned 32-bit byte offsets, so every tap is a `global_load_dwordx4 v, voff, s[base] offset:imm`: no 64-bit
// vector address arithmetic.  All CPL/8 x 4 loads are issued before the first use.
template <typename TIn, int CPL, bool INTERIOR, int PIXB>
__device__ __forceinline__ VecF<CPL> blend_taps(const char* __restrict__ base, const Taps& t) {
    constexpr int NK = CPL / 8;
    constexpr int CB = 8 * (int)sizeof(TIn);   // bytes of one 8-channel chunk
    const char* p00 = base + t.o00;
    const char* p10 = base + t.o10;
    const char* p01 = INTERIOR ? p00 + PIXB : base + t.o01;
    const char* p11 = INTERIOR ? p10 + PIXB : base + t.o11;
    VecF<CPL> r;
    if constexpr (sizeof(TIn) == 2 && Elem<TIn>::dtype == PSCV_F16) {
        // fp16 taps: v_fma_mix_f32 converts the half operand and does the fp32 FMA in ONE instruction.  The four
        // blend stages run across all channels of a chunk (independent chains back to back: no dependent-issue nops).
        uint4 a[NK], b[NK], c[NK], d[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            a[k] = *reinterpret_cast<const uint4*>(p00 + CB * k);
            b[k] = *reinterpret_cast<const uint4*>(p01 + CB * k);
            c[k] = *reinterpret_cast<const uint4*>(p10 + CB * k);
            d[k] = *reinterpret_cast<const uint4*>(p11 + CB * k);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const uint32_t aw[4] = {a[k].x, a[k].y, a[k].z, a[k].w}, bw[4] = {b[k].x, b[k].y, b[k].z, b[k].w};
            const uint32_t cw[4] = {c[k].x, c[k].y, c[k].z, c[k].w}, dw[4] = {d[k].x, d[k].y, d[k].z, d[k].w};
            float* o = r.v + 8 * k;
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[2 * q] = mul_mix_lo(aw[q], t.w00); o[2 * q + 1] = mul_mix_hi(aw[q], t.w00); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(bw[q], t.w01, o[2 * q]); o[2 * q + 1] = fma_mix_hi(bw[q], t.w01, o[2 * q + 1]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(cw[q], t.w10, o[2 * q]); o[2 * q + 1] = fma_mix_hi(cw[q], t.w10, o[2 * q + 1]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(dw[q], t.w11, o[2 * q]); o[2 * q + 1] = fma_mix_hi(dw[q], t.w11, o[2 * q + 1]); }
        }
    } else {
        const VecF<CPL> f00 = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(p00));
        const VecF<CPL> f01 = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(p01));
        const VecF<CPL> f10 = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(p10));
        const VecF<CPL> f11 = load_chan<TIn, CPL>(reinterpret_cast<const TIn*>(p11));
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            r.v[j] = fmaf(f11.v[j], t.w11, fmaf(f10.v[j], t.w10, fmaf(f01.v[j], t.w01, f00.v[j] * t.w00)));
    }
    return r;
}


}  // namespace pscv
