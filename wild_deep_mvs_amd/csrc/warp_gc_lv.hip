// LDS-staged group-wise correlation volumes (Vis-MVSNet pair branch), lane-owns-voxel design (round 4).
//
// Until round 4 the pair volumes came from the quad kernel (warp_cost_quad.hip: global taps, four lanes per 64-byte texel): 0.17-0.23
// of the HBM roofline, bound by the per-CU L1 tap rate.  The source boxes a tile of reference pixels can touch on a chunk of depth
// planes fit the LDS here as they do for MVSNet's variance sweep (scripts/dev/gc_feasibility.py: 361 us for the two four-view launches
// of the variance kernels against 617 us for the group-correlation launch on the stage-1 shape of BASELINE configuration 5), so this
// kernel reuses the machinery of warp_cost_lv.hip -- channel-chunk planar boxes staged with their zero padding, a lane owns a voxel,
// taps by `ds_read_b128` with immediate offsets, the next chunk's taps requested before the current one is blended -- with
//
//   * HOMOG geometry: hom = A p - (Bm p) / (d + 1e-9), index = hom.xy / hom.z * (W-1)/W            homography.py:63-69, 92-96, 113-117
//   * per-batch planes [B,D] (stage 1) AND per-pixel planes [B,D,h,w] (the refinement stages: the depth range of a (tile, chunk) is
//     reduced over the tile's pixels; the projective map of the box tile x [1/dmax, 1/dmin] keeps its corners' hull);
//   * one source view after the other (no sums across views): position once per (voxel, view), eight chunks = eight groups of four
//     channels, c_g = sum_i ref[4g+i] * warped[4g+i]  (nn_utils.py:473-490), packed to 16 bytes, one store per (voxel, view) -- 8
//     x-adjacent voxels of a lane group write 128 contiguous bytes;
//   * views whose box lies outside the image store zeros; views whose box does not fit take per-lane 8-byte global taps (slow, rare);
//   * up to four source views per launch (one staging wave each); the host side loops over groups of four.
//
// The reference's grid clamp (+-1.1 normalised = [-0.05 (W-1), 1.05 (W-1)]) moves samples only between positions whose four taps all
// lie outside the image, as long as the map is at least 21 texels wide and high (smaller maps are left to the quad kernel).
// Same fp32 operation chain as the quad kernel for the warp; the group sums are spelled as the fma chain its compiler forms.  Stored
// values agree with the quad kernel's to one 16-bit ulp (tests/test_gpu_vis.py).  No packed fp32 instructions (Makefile).
#include <type_traits>

#include "warp_common.h"
#include "warp_lds.h"
#include "warp_lv.h"

namespace pscv {

template <typename TIn, typename TOut>
__global__ __launch_bounds__(LV_THREADS) __attribute__((amdgpu_waves_per_eu(LV_OCC, LV_OCC))) void warp_gc_lv_kernel(const WarpArgs a) {
#pragma clang fp contract(off)      // (every fma of this kernel is written as one, see warp_cost_lv.hip)
    constexpr int C = 32, PIXB = 64, OVB = 16;      // 8 groups x 16 bits per voxel and view
    static_assert(sizeof(TOut) == 2 && sizeof(TIn) == 2, "16-bit storage");
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];

    const int dc = blockIdx.y;
    const int tpx = gridDim.x >> 3;
    const int ntx = (a.w + LV_T - 1) / LV_T, nty = (a.h + LV_TH - 1) / LV_TH;
    const int tile = ((int)blockIdx.x & 7) * tpx + ((int)blockIdx.x >> 3);
    if (tile >= a.B * nty * ntx) return;
    const int trow = (int)(((float)tile + 0.5f) * (1.0f / (float)ntx));     // exact: tile < 2^22
    const int txi = tile - trow * ntx;
    const int b = (int)(((float)trow + 0.5f) * (1.0f / (float)nty));
    const int tyi = trow - b * nty;

    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1);       // MODE.FP16_OVFL: saturating f32 -> f16 stores
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int x0t = txi * LV_T, y0t = tyi * LV_TH;
    const int d0 = dc * a.ppd, d1 = min(a.D, d0 + a.ppd);
    const int nd = d1 - d0;
    const int hw = a.h * a.w;
    const float* const depth_b = a.depth + (long)b * a.depth_bstride;
    const int n_src = a.n_src;
    int* const table = reinterpret_cast<int*>(lsm + LV_TABLE);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lsm;
    const bool pp_planes = a.depth_per_pixel != 0;

    int prow, pcol, pp;
    lv_voxel_of(lane, prow, pcol, pp);
    int x = x0t + pcol, y = y0t + prow;
    const bool active_px = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int pflat = y * a.w + x;
    const float px = (float)x + 0.5f, py = (float)(y + a.ref_y0) + 0.5f;      // half-pixel centres  homography.py:78-79

    // per-batch planes: lane i holds plane d0 + i (<= 64 planes per chunk); per-pixel planes: the range over this lane's share of the
    // (tile, chunk): pixel lane & 31 of the tile, planes (lane >> 5) * 16 .. + 15 of the chunk (<= 32 planes per chunk)
    float dlane = 0.0f, dlo, dhi;
    if (!pp_planes) {
        dlane = depth_b[min(d0 + lane, d1 - 1)];
        dlo = dlane; dhi = dlane;
    } else {
        const int tp = lane & 31;
        const int qx = min(x0t + (tp & 7), a.w - 1), qy = min(y0t + (tp >> 3), a.h - 1);
        const float* dp = depth_b + (long)qy * a.w + qx;
        dlo = INFINITY; dhi = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = dp[(long)(d0 + min((lane >> 5) * 16 + i, nd - 1)) * hw];
            dlo = fminf(dlo, v); dhi = fmaxf(dhi, v);
        }
    }

    float rf[C];
    {
        const TIn* rp = reinterpret_cast<const TIn*>(a.ref) + ((long)b * hw + pflat) * C;
#pragma unroll
        for (int k = 0; k < C / 8; ++k) {
            const f32x8 t = Elem<TIn>::load8(rp + 8 * k);
#pragma unroll
            for (int i = 0; i < 8; ++i) rf[8 * k + i] = t.v[i];
        }
    }

    // ---- 1. wave k: texel box of source view k from the 8 corner projections (tile corners x depth extremes) ----
    if (wave < WL_MAX_SRC) {
        const int k = wave;
        const float dmin = wl_wave_reduce<false>(dlo), dmax = wl_wave_reduce<true>(dhi);
        const int corner = lane & 7;
        const float cx = ((corner & 1) ? (float)min(x0t + LV_T - 1, a.w - 1) : (float)x0t) + 0.5f;
        const float cy = (float)(((corner & 2) ? min(y0t + LV_TH - 1, a.h - 1) : y0t) + a.ref_y0) + 0.5f;
        const float d = (corner & 4) ? dmax : dmin;
        int cX0 = 0, cY0 = 0, cX1 = 1, cY1 = 1, pitch = 2, mode = WL_ZERO;
        if (k < n_src) {
            lv_cf cam = (lv_cf)(a.cams + ((long)k * a.B + b) * PSCV_CAM_FLOATS);
            const float ax = fmaf(cam[1], cy, cam[0] * cx) + cam[2];
            const float ay = fmaf(cam[4], cy, cam[3] * cx) + cam[5];
            const float az = fmaf(cam[7], cy, cam[6] * cx) + cam[8];
            const float bx = fmaf(cam[10], cy, cam[9] * cx) + cam[11];
            const float by = fmaf(cam[13], cy, cam[12] * cx) + cam[14];
            const float bz = fmaf(cam[16], cy, cam[15] * cx) + cam[17];
            const float inv_d = __builtin_amdgcn_rcpf(d + 1e-9f);
            const float hx = fmaf(-bx, inv_d, ax), hy = fmaf(-by, inv_d, ay), hz = fmaf(-bz, inv_d, az);
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            const float u = hx * inv_z * a.sx, v = hy * inv_z * a.sy;
            const float okf = (d > 1e-6f && hz > 1e-6f && fabsf(u) < 1e6f && fabsf(v) < 1e6f) ? 1.0f : 0.0f;   // also rejects NaN
            const float umin = wl_reduce8<false>(u), umax = wl_reduce8<true>(u);
            const float vmin = wl_reduce8<false>(v), vmax = wl_reduce8<true>(v);
            const bool ok = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wl_reduce8<false>(okf))) != 0;
            const float sl = 1.0f / 32.0f;
            const int X0 = __builtin_amdgcn_readfirstlane((int)floorf(umin - sl)), X1 = __builtin_amdgcn_readfirstlane((int)floorf(umax + sl)) + 1;
            const int Y0 = __builtin_amdgcn_readfirstlane((int)floorf(vmin - sl)), Y1 = __builtin_amdgcn_readfirstlane((int)floorf(vmax + sl)) + 1;
            mode = WL_DIRECT;
            if (ok) {
                const bool outside = X1 < 0 || Y1 < 0 || X0 > a.ws - 1 || Y0 > a.hs - 1;
                const bool inside = X0 >= 0 && Y0 >= 0 && X1 <= a.ws - 1 && Y1 <= a.hs - 1;
                cX0 = max(X0, -2); cX1 = min(X1, a.ws + 1); cY0 = max(Y0, -2); cY1 = min(Y1, a.hs + 1);
                const int bw = cX1 - cX0 + 1, bh = cY1 - cY0 + 1;
                pitch = bw;
                if (outside) mode = WL_ZERO;
                else if (bw <= LV_BOX_W && bh <= LV_BOX_H) mode = inside ? WL_FAST : WL_GEN;
            }
        }
        if (lane == 0) {
            int4* row = reinterpret_cast<int4*>(table + k * 8);
            row[0] = make_int4(cX0, cY0, cX1, cY1);
            row[1] = make_int4(0, pitch, mode, 0);
        }
    }
    __syncthreads();

    // ---- 2. every wave: box records -> scalar registers, arena allocation greedy in view order ----
    int bMode[WL_MAX_SRC], bP[WL_MAX_SRC], bE0[WL_MAX_SRC], bBase[WL_MAX_SRC];
    {
        int used = 0;
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            const int4 r0 = *reinterpret_cast<const int4*>(table + k * 8), r1 = *reinterpret_cast<const int4*>(table + k * 8 + 4);
            const int X0 = __builtin_amdgcn_readfirstlane(r0.x), Y0 = __builtin_amdgcn_readfirstlane(r0.y);
            const int Y1 = __builtin_amdgcn_readfirstlane(r0.w);
            bP[k] = __builtin_amdgcn_readfirstlane(r1.y);
            int mode = k < n_src ? __builtin_amdgcn_readfirstlane(r1.z) : WL_ZERO;
            const int need = bP[k] * (Y1 - Y0 + 1);
            if ((mode == WL_FAST || mode == WL_GEN) && used + need > LV_ARENA) mode = WL_DIRECT;
            if (a.variant == 7 && (mode == WL_FAST || mode == WL_GEN)) mode = WL_DIRECT;     // ("warp_tile" = 7: nothing staged, a test aid)
            bBase[k] = used;
            if (mode == WL_FAST || mode == WL_GEN) used += need;
            bMode[k] = mode;
            if (a.mode_hist && k < n_src && tid == 0) atomicAdd(a.mode_hist + k * 4 + mode, 1);
            bE0[k] = ((bBase[k] - Y0 * bP[k] - X0) << 4) + (int)lds0;
        }
    }

    // ---- 3. wave k stages view k ----
    {
        const int k = wave;
        const int4 f0 = *reinterpret_cast<const int4*>(table + k * 8);
        const int sX0 = __builtin_amdgcn_readfirstlane(f0.x), sY0 = __builtin_amdgcn_readfirstlane(f0.y);
        const int sX1 = __builtin_amdgcn_readfirstlane(f0.z), sY1 = __builtin_amdgcn_readfirstlane(f0.w);
        int sP16 = bP[0] << 4, sMode = bMode[0], sBase = bBase[0];
        const void* srcp = a.src[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) { sP16 = bP[t] << 4; sMode = bMode[t]; sBase = bBase[t]; srcp = a.src[t]; }
        sBase = (sBase << 4) - sY0 * sP16 - (sX0 << 4);
        if (k < n_src && (sMode == WL_FAST || sMode == WL_GEN)) lv_stage_box<TIn>(lsm, srcp, b, a.hs, a.ws, lane, sX0, sY0, sX1, sY1, sP16, sBase);
    }

    // ray terms A p and Bm p of this pixel, per view                                        homography.py:63-69
    float rax[WL_MAX_SRC], ray[WL_MAX_SRC], raz[WL_MAX_SRC], rbx[WL_MAX_SRC], rby[WL_MAX_SRC], rbz[WL_MAX_SRC];
#pragma unroll
    for (int k = 0; k < WL_MAX_SRC; ++k) {
        lv_cf cam = (lv_cf)(a.cams + ((long)min(k, n_src - 1) * a.B + b) * PSCV_CAM_FLOATS);
        rax[k] = fmaf(cam[1], py, cam[0] * px) + cam[2];
        ray[k] = fmaf(cam[4], py, cam[3] * px) + cam[5];
        raz[k] = fmaf(cam[7], py, cam[6] * px) + cam[8];
        rbx[k] = fmaf(cam[10], py, cam[9] * px) + cam[11];
        rby[k] = fmaf(cam[13], py, cam[12] * px) + cam[14];
        rbz[k] = fmaf(cam[16], py, cam[15] * px) + cam[17];
    }

    char* const out = reinterpret_cast<char*>(a.out);
    const unsigned long view_bytes = (unsigned long)a.out_view_stride * 2;
    const unsigned long img_bytes = (unsigned long)b * a.hs * a.ws * PIXB;
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);

    // group sums of one chunk -> 16 bits                                                    nn_utils.py:473-490
    auto group = [&](int c, const float (&wv)[4]) -> float {
        float g = fmaf(rf[4 * c + 3], wv[3], fmaf(rf[4 * c + 2], wv[2], fmaf(rf[4 * c + 1], wv[1], rf[4 * c] * wv[0])));
        asm volatile("" : "+v"(g));      // (no v_fma_mixlo_f16 fusion of the last fma and the conversion: two roundings, like the quad kernel)
        return g;
    };

    // ---- 4. sweep: a trip = the wave's 32 pixels on two adjacent planes; one source view after the other ----
    for (int t = wave; 2 * t < nd; t += LV_THREADS / 64) {
        const bool active = active_px && 2 * t + pp < nd && a.variant != 8;
        const int d = d0 + min(2 * t + pp, nd - 1);
        float dval;
        if (!pp_planes) {
            const float dv0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), 2 * t));
            const float dv1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), min(2 * t + 1, nd - 1)));
            dval = pp ? dv1 : dv0;
        } else {
            dval = depth_b[(long)d * hw + pflat];
        }
        const float inv_d = __builtin_amdgcn_rcpf(dval + 1e-9f);
        char* const vox = out + (((unsigned long)b * a.D + d) * hw + pflat) * OVB;

#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            if (k < n_src) {
                uint32_t pk[4] = {0u, 0u, 0u, 0u};
                if (bMode[k] == WL_FAST || bMode[k] == WL_GEN) {
                    const float hx = fmaf(-rbx[k], inv_d, rax[k]), hy = fmaf(-rby[k], inv_d, ray[k]), hz = fmaf(-rbz[k], inv_d, raz[k]);
                    const float inv_z = __builtin_amdgcn_rcpf(hz);       // (a staged box has every corner in front of the camera)
                    const float ix = hx * inv_z * a.sx, iy = hy * inv_z * a.sy;
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    const float fx = ix - x0f, fy = iy - y0f;
                    const float gx = 1.0f - fx, gy = 1.0f - fy;
                    const float w[4] = {gx * gy, fx * gy, gx * fy, fx * fy};
                    int x0 = (int)x0f, y0 = (int)y0f;
                    if (bMode[k] == WL_GEN) {      // top-left tap into the box (with its zero padding)
                        const int4 r0 = *reinterpret_cast<const int4*>(table + k * 8);
                        x0 = med3_i32(x0, r0.x, r0.z - 1); y0 = med3_i32(y0, r0.y, r0.w - 1);
                    }
                    const unsigned aT = (unsigned)((__mul24(y0, bP[k]) + x0) * 16 + bE0[k]);
                    const unsigned aB = aT + (unsigned)(bP[k] << 4);
                    wl_f4 T[2][4];
                    auto request = [&](int buf, int c) {
                        T[buf][0] = lv_tap(aT, c * LV_PLANE); T[buf][1] = lv_tap(aT, c * LV_PLANE + 16);
                        T[buf][2] = lv_tap(aB, c * LV_PLANE); T[buf][3] = lv_tap(aB, c * LV_PLANE + 16);
                    };
                    request(0, 0);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (c + 1 < 8) request((c + 1) & 1, c + 1);
                        float wv[4];
                        lv_blend(T[c & 1][0], T[c & 1][1], T[c & 1][2], T[c & 1][3], w, wv);
                        const float g = group(c, wv);
                        if (c & 1) pk[c >> 1] = wl_pack2<TOut>(__builtin_bit_cast(float, pk[c >> 1]), g);
                        else pk[c >> 1] = __builtin_bit_cast(uint32_t, g);
                    }
                } else if (bMode[k] == WL_DIRECT) {
                    // per-lane global taps: behind-camera test, grid clamp, zero padding              homography.py:92-117
                    const float* cam = a.cams + ((long)k * a.B + b) * PSCV_CAM_FLOATS;
                    float ix, iy;
                    sweep_index<PSCV_GEOM_HOMOG>(cam, px, py, dval, a, ix, iy);
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    Taps tp;
                    make_taps<false, PIXB>(ix - x0f, iy - y0f, (int)x0f, (int)y0f, a.hs, a.ws, 0u, tp);
                    const float w[4] = {tp.w00, tp.w01, tp.w10, tp.w11};
                    const char* img = reinterpret_cast<const char*>(a.src[k]) + img_bytes;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint2 g00 = *reinterpret_cast<const uint2*>(img + tp.o00 + c * 8), g01 = *reinterpret_cast<const uint2*>(img + tp.o01 + c * 8);
                        const uint2 g10 = *reinterpret_cast<const uint2*>(img + tp.o10 + c * 8), g11 = *reinterpret_cast<const uint2*>(img + tp.o11 + c * 8);
                        const wl_f4 t00 = wl_f4{Half16<TIn>::lo(g00.x), Half16<TIn>::hi(g00.x), Half16<TIn>::lo(g00.y), Half16<TIn>::hi(g00.y)};
                        const wl_f4 t01 = wl_f4{Half16<TIn>::lo(g01.x), Half16<TIn>::hi(g01.x), Half16<TIn>::lo(g01.y), Half16<TIn>::hi(g01.y)};
                        const wl_f4 t10 = wl_f4{Half16<TIn>::lo(g10.x), Half16<TIn>::hi(g10.x), Half16<TIn>::lo(g10.y), Half16<TIn>::hi(g10.y)};
                        const wl_f4 t11 = wl_f4{Half16<TIn>::lo(g11.x), Half16<TIn>::hi(g11.x), Half16<TIn>::lo(g11.y), Half16<TIn>::hi(g11.y)};
                        float wv[4];
                        lv_blend(t00, t01, t10, t11, w, wv);
                        const float g = group(c, wv);
                        if (c & 1) pk[c >> 1] = wl_pack2<TOut>(__builtin_bit_cast(float, pk[c >> 1]), g);
                        else pk[c >> 1] = __builtin_bit_cast(uint32_t, g);
                    }
                }
                // (WL_ZERO: every tap of the view is zero padding -> the correlation is zero)
                if (active) *reinterpret_cast<uint4*>(vox + (unsigned long)k * view_bytes) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
}

template <typename TIn, typename TOut>
static int gc_launch(const WarpArgs& a, hipStream_t st) {
    auto kern = warp_gc_lv_kernel<TIn, TOut>;
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LV_LDS);
        if (e != hipSuccess) { set_error("pscv_warp_cost(gc): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    const int tiles = a.B * ((a.h + LV_TH - 1) / LV_TH) * ((a.w + LV_T - 1) / LV_T);
    hipLaunchKernelGGL(kern, dim3(8 * ((tiles + 7) / 8), a.n_dchunks), dim3(LV_THREADS), LV_LDS, st, a);
    return 0;
}

extern int* g_wl_mode_hist;   // warp_cost_tiled.hip (pscv_debug_wl_mode_hist)
extern Knob g_warp_tile;       // warp_cost.hip

// Returns 0 if launched (one launch per group of four source views), 1 if this configuration is not covered (the caller uses the quad
// kernel), negative on error.
int warp_gc_lv_try(WarpArgs& a0, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || geom != PSCV_GEOM_HOMOG || cost != PSCV_COST_GROUPCORR) return 1;
    if ((in_dtype != PSCV_F16 && in_dtype != PSCV_BF16) || out_dtype != in_dtype) return 1;
    if (a0.n_src < 1 || a0.n_src > PSCV_MAX_SRC) return 1;
    if (a0.ws > 16384 || a0.hs > 16384 || a0.ws < 21 || a0.hs < 21) return 1;     // (below 21 texels the grid clamp reaches inside the image's tap range)
    const long tiles = (long)a0.B * ((a0.h + LV_TH - 1) / LV_TH) * ((a0.w + LV_T - 1) / LV_T);
    if (tiles >= (1L << 22)) return 1;
    const int ppd_max = a0.depth_per_pixel ? 32 : 64;
    int ppd = ppd_override > 0 ? min((ppd_override + 1) & ~1, ppd_max) : 32;
    while (ppd > 4 && tiles * ((a0.D + ppd - 1) / ppd) < 1024) ppd >>= 1;
    const long nblk = tiles * ((a0.D + ppd - 1) / ppd);
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(gc): bad grid %ld", nblk); return -1; }
    for (int v0 = 0; v0 < a0.n_src; v0 += WL_MAX_SRC) {
        WarpArgs a = a0;
        a.n_src = min(WL_MAX_SRC, a0.n_src - v0);
        for (int i = 0; i < PSCV_MAX_SRC; ++i) a.src[i] = (i < a.n_src) ? a0.src[v0 + i] : nullptr;
        a.cams = a0.cams + (long)v0 * a0.B * PSCV_CAM_FLOATS;
        a.out = reinterpret_cast<char*>(a0.out) + (unsigned long)v0 * a0.out_view_stride * 2;
        a.ppd = ppd;
        a.n_dchunks = (a0.D + ppd - 1) / ppd;
        a.mode_hist = g_wl_mode_hist ? g_wl_mode_hist + 0 : nullptr;
        a.variant = g_warp_tile;
        const int rc = in_dtype == PSCV_F16 ? gc_launch<f16_t, f16_t>(a, st) : gc_launch<bf16_t, bf16_t>(a, st);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace pscv
