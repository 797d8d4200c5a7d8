// LDS-staged plane-sweep warp + cost (gfx950): the source patches a reference tile can touch are copied into LDS
// once per (tile, depth chunk, view) and every bilinear tap is then an LDS read.
//
// Why: the direct-gather kernel (warp_cost.hip) is bound by the per-CU vector L1 / texture-address path -- 4 taps x
// 64 B x V-1 views per voxel = 4 GB per sweep at the headline size, TA busy 80 % (profiles/) -- while its HBM
// traffic is already the algorithmic minimum.  For a 16 x 8 reference tile and a run of depth planes, the sample
// positions in one source view lie inside the convex hull of the tile corners' projections at the nearest and
// farthest plane of the run (the warp is projective in (x, y) for fixed depth and moves each pixel monotonically
// along its epipolar line in depth), so a texel bounding box computed from 8 corner projections (+1 texel margin)
// covers every tap.  Each texel is fetched from L1/L2 once per block instead of ~3x per plane; the taps move to the
// 4x faster LDS path (conflict-free with the chunk ^= ((texel >> 2) & 1) << 1 swizzle).
//
// Views whose box does not fit the LDS budget, or that have a corner at / behind the source camera, fall back to
// direct global taps for that block (same arithmetic, so results do not depend on the path taken).
//
// Mapping: 256 threads = 128 pixels (16 x 8 tile) x 2 lanes; lane h of a pixel owns 16-byte channel chunks h and
// 2 + h of every 64-byte voxel, so each tap / store instruction moves 32 contiguous bytes per pixel.
#include "warp_common.h"

namespace pscv {

constexpr int WT_TW = 16, WT_TH = 8;
constexpr int WT_INFO_BYTES = 1024;                 // per-view patch descriptors (16 views x 8 ints) + scratch
constexpr int WT_TEXELS = 1152;                     // texel budget per block (72 KiB): two blocks per CU
constexpr int WT_LDS = WT_INFO_BYTES + WT_TEXELS * 64;

__device__ __forceinline__ int wt_chunk_off(int texel, int chunk) {
    return texel * 64 + ((chunk ^ (((texel >> 2) & 1) << 1)) << 4);
}

// unclamped source-image coordinate (already scaled to a pixel index) and the depth of the point in the source frame
template <int GEOM>
__device__ __forceinline__ void sweep_uvz(const float* __restrict__ cam, float px, float py, float d, const WarpArgs& a,
                                          float& u, float& v, float& z) {
    float hx, hy, hz;
    if (GEOM == PSCV_GEOM_PROJ) {
        const float rx = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ry = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float rz = fmaf(cam[7], py, cam[6] * px) + cam[8];
        hx = fmaf(rx, d, cam[9]); hy = fmaf(ry, d, cam[10]); hz = fmaf(rz, d, cam[11]);
    } else {
        const float ax = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ay = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float az = fmaf(cam[7], py, cam[6] * px) + cam[8];
        const float bx = fmaf(cam[10], py, cam[9] * px) + cam[11];
        const float by = fmaf(cam[13], py, cam[12] * px) + cam[14];
        const float bz = fmaf(cam[16], py, cam[15] * px) + cam[17];
        const float inv_d = 1.0f / (d + 1e-9f);
        hx = fmaf(-bx, inv_d, ax); hy = fmaf(-by, inv_d, ay); hz = fmaf(-bz, inv_d, az);
    }
    z = hz;
    const float inv_z = 1.0f / hz;
    u = hx * inv_z * a.sx;
    v = hy * inv_z * a.sy;
}

struct WtPatch { int base, x0, y0, bw, bh, staged; };

template <typename TIn, typename TOut, int GEOM, int COST, int NSRC>
__global__ __launch_bounds__(256, 2) void warp_cost_tiled_kernel(const WarpArgs a) {
    constexpr int C = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    int* info = reinterpret_cast<int*>(wsm);                 // [view][8]: x0, y0, bw, bh, ok
    unsigned char* tex = wsm + WT_INFO_BYTES;

    // ---- work decode: XCD-banded (tile-major, depth-chunk minor) ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r = nwg & 7;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int dc = wg % a.n_dchunks; wg /= a.n_dchunks;
    const int ntx = (a.w + WT_TW - 1) / WT_TW, nty = (a.h + WT_TH - 1) / WT_TH;
    const int txi = wg % ntx; wg /= ntx;
    const int tyi = wg % nty; wg /= nty;
    const int b = wg;

    const int tid = threadIdx.x;
    const int p = tid >> 1, hh = tid & 1, hsel = hh << 4;
    const int x0t = txi * WT_TW, y0t = tyi * WT_TH;
    int x = x0t + (p & (WT_TW - 1)), y = y0t + (p >> 4);
    const bool active = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int hw = a.h * a.w;
    const int pflat = y * a.w + x;
    const long pix = (long)b * hw + pflat;
    const float off = (GEOM == PSCV_GEOM_HOMOG) ? 0.5f : 0.0f;
    const float px = (float)x + off, py = (float)y + off;

    const int d0 = dc * a.ppd, d1 = min(a.D, d0 + a.ppd);

    // ---- 1. depth extremes of this chunk (planes need not be monotone) ----
    float dmin = a.depth[(long)b * a.depth_bstride + d0], dmax = dmin;
    for (int d = d0 + 1; d < d1; ++d) {
        const float dv = a.depth[(long)b * a.depth_bstride + d];
        dmin = fminf(dmin, dv); dmax = fmaxf(dmax, dv);
    }

    // ---- 2. texel bounding box per view from the 8 corner projections ----
    {
        const float cxl = (float)x0t + off, cxh = (float)min(x0t + WT_TW - 1, a.w - 1) + off;
        const float cyl = (float)y0t + off, cyh = (float)min(y0t + WT_TH - 1, a.h - 1) + off;
        for (int vb = 0; vb < a.n_src; vb += 32) {
            const int view = vb + (tid >> 3), corner = tid & 7;
            const int vc = min(view, a.n_src - 1);
            const float* cam = a.cams + ((long)vc * a.B + b) * PSCV_CAM_FLOATS;
            float u, v, z;
            sweep_uvz<GEOM>(cam, (corner & 1) ? cxh : cxl, (corner & 2) ? cyh : cyl, (corner & 4) ? dmax : dmin, a, u, v, z);
            bool ok = z > 1e-6f && fabsf(u) < 1e7f && fabsf(v) < 1e7f;   // also rejects NaN
            float umin = u, umax = u, vmin = v, vmax = v;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                umin = fminf(umin, __shfl_xor(umin, m, 64)); umax = fmaxf(umax, __shfl_xor(umax, m, 64));
                vmin = fminf(vmin, __shfl_xor(vmin, m, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
                ok = ok && (__shfl_xor((int)ok, m, 64) != 0);
            }
            if (corner == 0 && view < a.n_src) {
                const bool dpos = (GEOM == PSCV_GEOM_PROJ) || dmin > 1e-6f;   // HOMOG is projective in 1/d
                int X0 = 0, Y0 = 0, X1 = -1, Y1 = -1;
                if (ok && dpos) {
                    X0 = max((int)floorf(umin) - 1, 0); X1 = min((int)floorf(umax) + 2, a.ws - 1);
                    Y0 = max((int)floorf(vmin) - 1, 0); Y1 = min((int)floorf(vmax) + 2, a.hs - 1);
                }
                int* o = info + view * 8;
                o[0] = X0; o[1] = Y0; o[2] = X1 - X0 + 1; o[3] = Y1 - Y0 + 1; o[4] = (ok && dpos) ? 1 : 0;
            }
        }
    }
    __syncthreads();

    // ---- 3. stage the boxes that fit (greedy in view order); every thread derives the same allocation ----
    int used = 0;
    for (int v = 0; v < a.n_src; ++v) {
        const int X0 = info[v * 8 + 0], Y0 = info[v * 8 + 1], bw = info[v * 8 + 2], bh = info[v * 8 + 3];
        const int size = bw * bh;
        const bool staged = info[v * 8 + 4] && bw > 0 && bh > 0 && used + size <= WT_TEXELS;
        if (staged) {
            // one wave per patch row (rows are contiguous runs of bw x 64 B in the source): no integer division
            const TIn* img = reinterpret_cast<const TIn*>(a.src[v]);
            const int lane = tid & 63, wave = tid >> 6;
            for (int ty = wave; ty < bh; ty += 4) {
                const TIn* rowp = img + (((long)b * a.hs + Y0 + ty) * a.ws + X0) * C;
                const int t0 = used + ty * bw;
                for (int id = lane; id < bw * 4; id += 64) {
                    const uint4 val = *reinterpret_cast<const uint4*>(rowp + id * 8);
                    *reinterpret_cast<uint4*>(tex + wt_chunk_off(t0 + (id >> 2), id & 3)) = val;
                }
            }
        }
        __syncthreads();   // (also orders the info reads above before the rewrite below)
        if (tid == 0) { info[v * 8 + 5] = staged ? used : -1; }
        if (staged) used += size;
    }
    __syncthreads();

    // per-view patch descriptors -> scalar registers (the view loop below is fully unrolled: static indices)
    int vbase[NSRC], pX0[NSRC], pY0[NSRC], pBW[NSRC], pBH[NSRC];
#pragma unroll
    for (int v = 0; v < NSRC; ++v) {
        pX0[v] = __builtin_amdgcn_readfirstlane(info[v * 8 + 0]);
        pY0[v] = __builtin_amdgcn_readfirstlane(info[v * 8 + 1]);
        pBW[v] = __builtin_amdgcn_readfirstlane(info[v * 8 + 2]);
        pBH[v] = __builtin_amdgcn_readfirstlane(info[v * 8 + 3]);
        vbase[v] = __builtin_amdgcn_readfirstlane(info[v * 8 + 5]);
    }

    // ---- 4. sweep the planes ----
    const TIn* ref = reinterpret_cast<const TIn*>(a.ref);
    TOut* out = reinterpret_cast<TOut*>(a.out);
    VecF<16> rf;   // channels [8h, 8h+8) and [16+8h, 24+8h)
    {
        const f32x8 lo = Elem<TIn>::load8(ref + pix * C + hh * 8);
        const f32x8 hi = Elem<TIn>::load8(ref + pix * C + 16 + hh * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rf.v[j] = lo.v[j]; rf.v[8 + j] = hi.v[j]; }
    }
    const float invN = 1.0f / (float)(a.n_src + 1);
    const float invN2 = invN * invN;

    for (int d = d0; d < d1; ++d) {
        const float dval = a.depth[(long)b * a.depth_bstride + d];
        const long vox = ((long)b * a.D + d) * hw + pflat;
        VecF<16> acc0, acc1;
        float sum_e = 0.0f;
        if (COST == PSCV_COST_SOFTMIN) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc0.v[j] = 0.0f; acc1.v[j] = 0.0f; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc0.v[j] = rf.v[j]; acc1.v[j] = rf.v[j] * rf.v[j]; }
        }

#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const float* cam = a.cams + ((long)v * a.B + b) * PSCV_CAM_FLOATS;
            float ix, iy;
            sweep_index<GEOM>(cam, px, py, dval, a, ix, iy);
            // bilinear taps, identical arithmetic to gather_bilinear (warp_common.h)
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float fx = ix - x0f, fy = iy - y0f;
            const int xi0 = (int)x0f, yi0 = (int)y0f, xi1 = xi0 + 1, yi1 = yi0 + 1;
            const bool vx0 = (unsigned)xi0 < (unsigned)a.ws, vx1 = (unsigned)xi1 < (unsigned)a.ws;
            const bool vy0 = (unsigned)yi0 < (unsigned)a.hs, vy1 = (unsigned)yi1 < (unsigned)a.hs;
            const float gx = 1.0f - fx, gy = 1.0f - fy;
            const float w00 = (vx0 && vy0) ? gx * gy : 0.0f, w01 = (vx1 && vy0) ? fx * gy : 0.0f;
            const float w10 = (vx0 && vy1) ? gx * fy : 0.0f, w11 = (vx1 && vy1) ? fx * fy : 0.0f;
            const int xc0 = min(max(xi0, 0), a.ws - 1), xc1 = min(max(xi1, 0), a.ws - 1);
            const int yc0 = min(max(yi0, 0), a.hs - 1), yc1 = min(max(yi1, 0), a.hs - 1);

            const int base = vbase[v];   // wave-uniform
            f32x8 t00a, t00b, t01a, t01b, t10a, t10b, t11a, t11b;
            if (base >= 0) {
                const int X0 = pX0[v], Y0 = pY0[v], bw = pBW[v], bh = pBH[v];
                // a zero-weight (out-of-image) tap may fall outside the box: clamp it into the box
                const int bx0 = min(max(xc0 - X0, 0), bw - 1), bx1 = min(max(xc1 - X0, 0), bw - 1);
                const int by0 = min(max(yc0 - Y0, 0), bh - 1), by1 = min(max(yc1 - Y0, 0), bh - 1);
                const int r0 = base + by0 * bw, r1 = base + by1 * bw;
                // byte offset of chunk hh of texel e: e*64 + ((hh*16) ^ swz), swz = bit 2 of e moved to bit 5;
                // chunk 2+hh of the same texel is that offset with bit 5 flipped
                auto toff = [&](int e) { return (e << 6) + (hsel ^ ((e << 3) & 32)); };
                const int o00 = toff(r0 + bx0), o01 = toff(r0 + bx1), o10 = toff(r1 + bx0), o11 = toff(r1 + bx1);
                auto ld = [&](int o) { return Elem<TIn>::load8(reinterpret_cast<const TIn*>(tex + o)); };
                t00a = ld(o00); t00b = ld(o00 ^ 32);
                t01a = ld(o01); t01b = ld(o01 ^ 32);
                t10a = ld(o10); t10b = ld(o10 ^ 32);
                t11a = ld(o11); t11b = ld(o11 ^ 32);
            } else {
                const TIn* img = reinterpret_cast<const TIn*>(a.src[v]);
                const long row0 = ((long)b * a.hs + yc0) * a.ws, row1 = ((long)b * a.hs + yc1) * a.ws;
                t00a = Elem<TIn>::load8(img + (row0 + xc0) * C + hh * 8); t00b = Elem<TIn>::load8(img + (row0 + xc0) * C + 16 + hh * 8);
                t01a = Elem<TIn>::load8(img + (row0 + xc1) * C + hh * 8); t01b = Elem<TIn>::load8(img + (row0 + xc1) * C + 16 + hh * 8);
                t10a = Elem<TIn>::load8(img + (row1 + xc0) * C + hh * 8); t10b = Elem<TIn>::load8(img + (row1 + xc0) * C + 16 + hh * 8);
                t11a = Elem<TIn>::load8(img + (row1 + xc1) * C + hh * 8); t11b = Elem<TIn>::load8(img + (row1 + xc1) * C + 16 + hh * 8);
            }
            VecF<16> wv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wv.v[j] = fmaf(t11a.v[j], w11, fmaf(t10a.v[j], w10, fmaf(t01a.v[j], w01, t00a.v[j] * w00)));
                wv.v[8 + j] = fmaf(t11b.v[j], w11, fmaf(t10b.v[j], w10, fmaf(t01b.v[j], w01, t00b.v[j] * w00)));
            }

            if (COST == PSCV_COST_SOFTMIN) {
                VecF<16> diff;
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float t = rf.v[j] - wv.v[j];
                    diff.v[j] = t * t;
                    part += diff.v[j];
                }
                part += __shfl_xor(part, 1, 64);
                const float e = __expf(-a.temp * part);
                sum_e += e;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc0.v[j] = fmaf(e, diff.v[j], acc0.v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc0.v[j] += wv.v[j];
                    acc1.v[j] = fmaf(wv.v[j], wv.v[j], acc1.v[j]);
                }
            }
        }

        f32x8 oa, ob;
        if (COST == PSCV_COST_VARIANCE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                oa.v[j] = acc1.v[j] * invN - (acc0.v[j] * acc0.v[j]) * invN2;
                ob.v[j] = acc1.v[8 + j] * invN - (acc0.v[8 + j] * acc0.v[8 + j]) * invN2;
            }
        } else if (COST == PSCV_COST_VARIANCE_CVP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ma = acc0.v[j] * invN, mb = acc0.v[8 + j] * invN;
                oa.v[j] = acc1.v[j] * invN - ma * ma;
                ob.v[j] = acc1.v[8 + j] * invN - mb * mb;
            }
        } else {
            const float inv = 1.0f / (sum_e + 1e-6f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { oa.v[j] = acc0.v[j] * inv; ob.v[j] = acc0.v[8 + j] * inv; }
        }
        if (active) {
            Elem<TOut>::store8(out + vox * C + hh * 8, oa);
            Elem<TOut>::store8(out + vox * C + 16 + hh * 8, ob);
        }
    }
}

template <typename TIn, typename TOut, int GEOM, int COST, int NSRC>
static int wt_launch_n(const WarpArgs& a, int nblk, hipStream_t st) {
    auto kern = warp_cost_tiled_kernel<TIn, TOut, GEOM, COST, NSRC>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WT_LDS);
        if (e != hipSuccess) { set_error("pscv_warp_cost(tiled): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), WT_LDS, st, a);
    return 0;
}

// the source-view loop is unrolled at compile time; other view counts use the direct kernel
template <typename TIn, typename TOut, int GEOM, int COST>
static int wt_launch(const WarpArgs& a, int nblk, hipStream_t st) {
    switch (a.n_src) {
        case 2: return wt_launch_n<TIn, TOut, GEOM, COST, 2>(a, nblk, st);
        case 3: return wt_launch_n<TIn, TOut, GEOM, COST, 3>(a, nblk, st);
        case 4: return wt_launch_n<TIn, TOut, GEOM, COST, 4>(a, nblk, st);
    }
    return 1;
}

template <typename T>
static int wt_dispatch(const WarpArgs& a, int geom, int cost, int nblk, hipStream_t st) {
    if (geom == PSCV_GEOM_PROJ) {
        if (cost == PSCV_COST_VARIANCE) return wt_launch<T, T, PSCV_GEOM_PROJ, PSCV_COST_VARIANCE>(a, nblk, st);
        if (cost == PSCV_COST_VARIANCE_CVP) return wt_launch<T, T, PSCV_GEOM_PROJ, PSCV_COST_VARIANCE_CVP>(a, nblk, st);
        if (cost == PSCV_COST_SOFTMIN) return wt_launch<T, T, PSCV_GEOM_PROJ, PSCV_COST_SOFTMIN>(a, nblk, st);
    }
    return 1;   // not handled here
}

// Returns 0 if launched, 1 if this configuration is not covered by the tiled kernel (caller uses the direct
// kernel), negative on error.
int warp_cost_tiled_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || a.depth_per_pixel || in_dtype != out_dtype || (in_dtype != PSCV_F16 && in_dtype != PSCV_BF16)) return 1;
    if (a.n_src < 2 || a.n_src > 4) return 1;   // unrolled view counts (more views spill registers: direct kernel)
    const long tiles = (long)a.B * ((a.h + WT_TH - 1) / WT_TH) * ((a.w + WT_TW - 1) / WT_TW);
    int ppd = ppd_override > 0 ? ppd_override : 16;   // planes per block: amortises the patch staging
    while (ppd > 2 && tiles * ((a.D + ppd - 1) / ppd) < 1024) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    const long nblk = tiles * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(tiled): bad grid %ld", nblk); return -1; }
    return in_dtype == PSCV_F16 ? wt_dispatch<f16_t>(a, geom, cost, (int)nblk, st) : wt_dispatch<bf16_t>(a, geom, cost, (int)nblk, st);
}

}  // namespace pscv
