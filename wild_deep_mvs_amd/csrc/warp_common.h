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
    float temp;
    float sx, sy;        // index scale: PROJ 1, HOMOG (W-1)/W
    float xlo, xhi, ylo, yhi;  // clamp of the pixel index implied by the reference's grid clamp
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

// Zero-padded bilinear gather of this lane's CPL channels.
template <typename TIn, int CPL>
__device__ __forceinline__ VecF<CPL> gather_bilinear(const TIn* __restrict__ img, int b, int hs, int ws, int C,
                                                     int choff, float ix, float iy) {
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (unsigned)x0 < (unsigned)ws, vx1 = (unsigned)x1 < (unsigned)ws;
    const bool vy0 = (unsigned)y0 < (unsigned)hs, vy1 = (unsigned)y1 < (unsigned)hs;
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float w00 = (vx0 && vy0) ? gx * gy : 0.0f;
    const float w01 = (vx1 && vy0) ? fx * gy : 0.0f;
    const float w10 = (vx0 && vy1) ? gx * fy : 0.0f;
    const float w11 = (vx1 && vy1) ? fx * fy : 0.0f;
    const int xc0 = min(max(x0, 0), ws - 1), xc1 = min(max(x1, 0), ws - 1);
    const int yc0 = min(max(y0, 0), hs - 1), yc1 = min(max(y1, 0), hs - 1);
    // wave-uniform batch base (SGPR pair) + 32-bit per-lane element offsets (feature maps are far below 2^31
    // elements): the loads use the scalar-base addressing form, no 64-bit vector address arithmetic
    const TIn* base = img + (long)b * hs * ws * C;
    const int o00 = (yc0 * ws + xc0) * C + choff, o01 = (yc0 * ws + xc1) * C + choff;
    const int o10 = (yc1 * ws + xc0) * C + choff, o11 = (yc1 * ws + xc1) * C + choff;
    VecF<CPL> r;
    if constexpr (sizeof(TIn) == 2 && Elem<TIn>::dtype == PSCV_F16) {
        // fp16 taps: v_fma_mix_f32 converts the half operand and does the fp32 FMA in ONE instruction, so the 64
        // v_cvt_f32_f16 of a 16-channel blend disappear; the arithmetic (exact convert, fp32 FMA chain in the same
        // order) is bit-identical to the generic path below.  The kernel is VALU-issue bound (profiles/).
#pragma unroll
        for (int k = 0; k < CPL / 8; ++k) {
            const uint4 a = *reinterpret_cast<const uint4*>(base + o00 + 8 * k);
            const uint4 bq = *reinterpret_cast<const uint4*>(base + o01 + 8 * k);
            const uint4 c = *reinterpret_cast<const uint4*>(base + o10 + 8 * k);
            const uint4 d = *reinterpret_cast<const uint4*>(base + o11 + 8 * k);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
            const uint32_t cw[4] = {c.x, c.y, c.z, c.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r.v[8 * k + 2 * q] = fma_mix_lo(dw[q], w11, fma_mix_lo(cw[q], w10, fma_mix_lo(bw[q], w01, fma_mix_lo(aw[q], w00, 0.0f))));
                r.v[8 * k + 2 * q + 1] = fma_mix_hi(dw[q], w11, fma_mix_hi(cw[q], w10, fma_mix_hi(bw[q], w01, fma_mix_hi(aw[q], w00, 0.0f))));
            }
        }
    } else {
        const VecF<CPL> f00 = load_chan<TIn, CPL>(base + o00);
        const VecF<CPL> f01 = load_chan<TIn, CPL>(base + o01);
        const VecF<CPL> f10 = load_chan<TIn, CPL>(base + o10);
        const VecF<CPL> f11 = load_chan<TIn, CPL>(base + o11);
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            r.v[j] = fmaf(f11.v[j], w11, fmaf(f10.v[j], w10, fmaf(f01.v[j], w01, f00.v[j] * w00)));
    }
    return r;
}


}  // namespace pscv
