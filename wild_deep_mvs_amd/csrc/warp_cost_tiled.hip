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

typedef float wt_f2 __attribute__((ext_vector_type(2)));

// 8 channels of one chunk column (ck = 0: chunk hh, 1: chunk 2 + hh) from the four taps
template <typename TIn>
__device__ __forceinline__ void wt_mix8(const uint4 (&t)[4][2], int ck, const float (&w)[4], float* o) {
    const uint32_t aw[4] = {t[0][ck].x, t[0][ck].y, t[0][ck].z, t[0][ck].w}, bw[4] = {t[1][ck].x, t[1][ck].y, t[1][ck].z, t[1][ck].w};
    const uint32_t cw[4] = {t[2][ck].x, t[2][ck].y, t[2][ck].z, t[2][ck].w}, dw[4] = {t[3][ck].x, t[3][ck].y, t[3][ck].z, t[3][ck].w};
    if constexpr (Half16<TIn>::dtype == PSCV_F16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] = mul_mix_lo(aw[q], w[0]); o[2 * q + 1] = mul_mix_hi(aw[q], w[0]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(bw[q], w[1], o[2 * q]); o[2 * q + 1] = fma_mix_hi(bw[q], w[1], o[2 * q + 1]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(cw[q], w[2], o[2 * q]); o[2 * q + 1] = fma_mix_hi(cw[q], w[2], o[2 * q + 1]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] = fma_mix_lo(dw[q], w[3], o[2 * q]); o[2 * q + 1] = fma_mix_hi(dw[q], w[3], o[2 * q + 1]); }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[2 * q] = fmaf(Half16<TIn>::lo(dw[q]), w[3], fmaf(Half16<TIn>::lo(cw[q]), w[2], fmaf(Half16<TIn>::lo(bw[q]), w[1], Half16<TIn>::lo(aw[q]) * w[0])));
            o[2 * q + 1] = fmaf(Half16<TIn>::hi(dw[q]), w[3], fmaf(Half16<TIn>::hi(cw[q]), w[2], fmaf(Half16<TIn>::hi(bw[q]), w[1], Half16<TIn>::hi(aw[q]) * w[0])));
        }
    }
}

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
    // Per-view constants of this lane's pixel stay in registers for the whole sweep (the view loop is unrolled): the
    // depth-independent ray terms rot (x, y, 1) and the translation.  PROJ only (wt_dispatch).
    float rayx[NSRC], rayy[NSRC], rayz[NSRC], trx[NSRC], try_[NSRC], trz[NSRC];
#pragma unroll
    for (int v = 0; v < NSRC; ++v) {
        const float* cam = a.cams + ((long)v * a.B + b) * PSCV_CAM_FLOATS;
        rayx[v] = fmaf(cam[1], py, cam[0] * px) + cam[2];
        rayy[v] = fmaf(cam[4], py, cam[3] * px) + cam[5];
        rayz[v] = fmaf(cam[7], py, cam[6] * px) + cam[8];
        trx[v] = cam[9]; try_[v] = cam[10]; trz[v] = cam[11];
    }
    const TIn* ref = reinterpret_cast<const TIn*>(a.ref);
    char* const out = reinterpret_cast<char*>(a.out);
    wt_f2 rf2[8], rfsq[8];   // channels [8h, 8h+8) and [16+8h, 24+8h) as pairs
    float rf[16];
    {
        const f32x8 lo = Elem<TIn>::load8(ref + pix * C + hh * 8);
        const f32x8 hi = Elem<TIn>::load8(ref + pix * C + 16 + hh * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rf[j] = lo.v[j]; rf[8 + j] = hi.v[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { rf2[j] = wt_f2{rf[2 * j], rf[2 * j + 1]}; rfsq[j] = rf2[j] * rf2[j]; }
    }
    const float invN = 1.0f / (float)(a.n_src + 1);
    const float invN2 = invN * invN;
    const unsigned long plane_bytes = (unsigned long)hw * C * sizeof(TOut);
    const unsigned lane_out = (unsigned)pflat * (C * (unsigned)sizeof(TOut)) + (unsigned)hh * (8 * (unsigned)sizeof(TOut));
    const unsigned long img_bytes = (unsigned long)b * a.hs * a.ws * 64;

    for (int d = d0; d < d1; ++d) {
        const float dval = a.depth[(long)b * a.depth_bstride + d];
        wt_f2 s2[8], q2[8];
        float sum_e = 0.0f;
        if (COST == PSCV_COST_SOFTMIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s2[j] = wt_f2{0.f, 0.f};
        }

#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const float hx = fmaf(rayx[v], dval, trx[v]), hy = fmaf(rayy[v], dval, try_[v]), hz = fmaf(rayz[v], dval, trz[v]);
            const bool front = hz > 0.0f;
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            const float u = front ? hx * inv_z : -10.0f, w_ = front ? hy * inv_z : -10.0f;
            const float ix = __builtin_amdgcn_fmed3f(u, a.xlo, a.xhi), iy = __builtin_amdgcn_fmed3f(w_, a.ylo, a.yhi);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float fx = ix - x0f, fy = iy - y0f;
            const int xi0 = (int)x0f, yi0 = (int)y0f;
            const bool interior = (unsigned)xi0 < (unsigned)(a.ws - 1) && (unsigned)yi0 < (unsigned)(a.hs - 1);
            const int base = vbase[v];   // wave-uniform; < 0: this view's box is not staged
            uint4 t[4][2];               // [tap][chunk hh / 2 + hh]
            float w4[4];
            if (base >= 0 && __builtin_amdgcn_ballot_w64(!interior) == 0) {
                // every tap of the wave is inside the image, hence inside the staged box: no masks, no clamps
                const float gx = 1.0f - fx, gy = 1.0f - fy;
                w4[0] = gx * gy; w4[1] = fx * gy; w4[2] = gx * fy; w4[3] = fx * fy;
                const int e00 = base + (yi0 - pY0[v]) * pBW[v] + (xi0 - pX0[v]);
                const int e10 = e00 + pBW[v];
                // byte offset of chunk hh of texel e: e*64 + ((hh*16) ^ swz), swz = bit 2 of e moved to bit 5; chunk
                // 2+hh of the same texel is that offset with bit 5 flipped
                auto toff = [&](int e) { return (e << 6) + (hsel ^ ((e << 3) & 32)); };
                const int o00 = toff(e00), o01 = toff(e00 + 1), o10 = toff(e10), o11 = toff(e10 + 1);
                t[0][0] = *reinterpret_cast<const uint4*>(tex + o00); t[0][1] = *reinterpret_cast<const uint4*>(tex + (o00 ^ 32));
                t[1][0] = *reinterpret_cast<const uint4*>(tex + o01); t[1][1] = *reinterpret_cast<const uint4*>(tex + (o01 ^ 32));
                t[2][0] = *reinterpret_cast<const uint4*>(tex + o10); t[2][1] = *reinterpret_cast<const uint4*>(tex + (o10 ^ 32));
                t[3][0] = *reinterpret_cast<const uint4*>(tex + o11); t[3][1] = *reinterpret_cast<const uint4*>(tex + (o11 ^ 32));
            } else {
                Taps tp;
                make_taps<false, 64>(fx, fy, xi0, yi0, a.hs, a.ws, 0u, tp);
                w4[0] = tp.w00; w4[1] = tp.w01; w4[2] = tp.w10; w4[3] = tp.w11;
                if (base >= 0) {
                    // a zero-weight (out-of-image) tap may fall outside the box: clamp it into the box
                    const int X0 = pX0[v], Y0 = pY0[v], bw = pBW[v], bh = pBH[v];
                    const int xc0 = med3_i32(xi0, 0, a.ws - 1), xc1 = med3_i32(xi0 + 1, 0, a.ws - 1);
                    const int yc0 = med3_i32(yi0, 0, a.hs - 1), yc1 = med3_i32(yi0 + 1, 0, a.hs - 1);
                    const int bx0 = med3_i32(xc0 - X0, 0, bw - 1), bx1 = med3_i32(xc1 - X0, 0, bw - 1);
                    const int r0 = base + med3_i32(yc0 - Y0, 0, bh - 1) * bw, r1 = base + med3_i32(yc1 - Y0, 0, bh - 1) * bw;
                    auto toff = [&](int e) { return (e << 6) + (hsel ^ ((e << 3) & 32)); };
                    const int o00 = toff(r0 + bx0), o01 = toff(r0 + bx1), o10 = toff(r1 + bx0), o11 = toff(r1 + bx1);
                    t[0][0] = *reinterpret_cast<const uint4*>(tex + o00); t[0][1] = *reinterpret_cast<const uint4*>(tex + (o00 ^ 32));
                    t[1][0] = *reinterpret_cast<const uint4*>(tex + o01); t[1][1] = *reinterpret_cast<const uint4*>(tex + (o01 ^ 32));
                    t[2][0] = *reinterpret_cast<const uint4*>(tex + o10); t[2][1] = *reinterpret_cast<const uint4*>(tex + (o10 ^ 32));
                    t[3][0] = *reinterpret_cast<const uint4*>(tex + o11); t[3][1] = *reinterpret_cast<const uint4*>(tex + (o11 ^ 32));
                } else {
                    const char* img = reinterpret_cast<const char*>(a.src[v]) + img_bytes + hh * 16;
                    t[0][0] = *reinterpret_cast<const uint4*>(img + tp.o00); t[0][1] = *reinterpret_cast<const uint4*>(img + tp.o00 + 32);
                    t[1][0] = *reinterpret_cast<const uint4*>(img + tp.o01); t[1][1] = *reinterpret_cast<const uint4*>(img + tp.o01 + 32);
                    t[2][0] = *reinterpret_cast<const uint4*>(img + tp.o10); t[2][1] = *reinterpret_cast<const uint4*>(img + tp.o10 + 32);
                    t[3][0] = *reinterpret_cast<const uint4*>(img + tp.o11); t[3][1] = *reinterpret_cast<const uint4*>(img + tp.o11 + 32);
                }
            }
            float wv[16];
            wt_mix8<TIn>(t, 0, w4, wv);
            wt_mix8<TIn>(t, 1, w4, wv + 8);

            if (COST == PSCV_COST_SOFTMIN) {
                float diff[16], part = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float tt = rf[j] - wv[j];
                    diff[j] = tt * tt;
                    part += diff[j];
                }
                part += __shfl_xor(part, 1, 64);
                const float e = __expf(-a.temp * part);
                sum_e += e;
#pragma unroll
                for (int j = 0; j < 8; ++j) s2[j] = __builtin_elementwise_fma(wt_f2{e, e}, wt_f2{diff[2 * j], diff[2 * j + 1]}, s2[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const wt_f2 x = wt_f2{wv[2 * j], wv[2 * j + 1]};
                    if (v == 0) { s2[j] = rf2[j] + x; q2[j] = __builtin_elementwise_fma(x, x, rfsq[j]); }
                    else { s2[j] += x; q2[j] = __builtin_elementwise_fma(x, x, q2[j]); }
                }
            }
        }

        f32x8 oa, ob;
        if (COST == PSCV_COST_VARIANCE) {
            const wt_f2 n1 = wt_f2{invN, invN}, n2 = wt_f2{invN2, invN2};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wt_f2 ra = q2[j] * n1 - (s2[j] * s2[j]) * n2, rb = q2[4 + j] * n1 - (s2[4 + j] * s2[4 + j]) * n2;
                oa.v[2 * j] = ra[0]; oa.v[2 * j + 1] = ra[1]; ob.v[2 * j] = rb[0]; ob.v[2 * j + 1] = rb[1];
            }
        } else if (COST == PSCV_COST_VARIANCE_CVP) {
            const wt_f2 n1 = wt_f2{invN, invN};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wt_f2 ma = s2[j] * n1, mb = s2[4 + j] * n1;
                const wt_f2 ra = q2[j] * n1 - ma * ma, rb = q2[4 + j] * n1 - mb * mb;
                oa.v[2 * j] = ra[0]; oa.v[2 * j + 1] = ra[1]; ob.v[2 * j] = rb[0]; ob.v[2 * j + 1] = rb[1];
            }
        } else {
            const float inv = 1.0f / (sum_e + 1e-6f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                oa.v[2 * j] = s2[j][0] * inv; oa.v[2 * j + 1] = s2[j][1] * inv;
                ob.v[2 * j] = s2[4 + j][0] * inv; ob.v[2 * j + 1] = s2[4 + j][1] * inv;
            }
        }
        if (active) {
            TOut* op = reinterpret_cast<TOut*>(out + ((unsigned long)b * a.D + d) * plane_bytes + lane_out);
            Elem<TOut>::store8(op, oa);
            Elem<TOut>::store8(op + 16, ob);
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
