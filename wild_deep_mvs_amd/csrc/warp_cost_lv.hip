// LDS-staged plane-sweep warp + variance cost, third design (round 4): a LANE owns a voxel.  Selected by
// pscv_set_tuning("warp_tiled", 4); NOT the default (see "Measured" below).
//
// What the second design (warp_cost_tiled.hip: a quad of lanes owns a voxel, lane l holds channels 8l..8l+7 and computes the sample
// position in view l) pays per 16 voxels x 4 views = 295 vector-ALU instructions (profiles/r04_warp_isa_breakdown.txt): 216 are the
// reference's arithmetic, 79 are overhead, and 51 of those exist only because four lanes share a voxel: the coordinate arithmetic
// (23, done in every lane of the quad for one view each), the DPP broadcasts that hand weights and tap addresses round the quad (20,
// at 4.3 cycles each) and the tap address sums (8).  Here a lane owns ALL 32 channels of its voxel:
//
//   * the coordinate arithmetic runs once per (voxel, view) in the lane that uses it: ~21 instructions per voxel-view, no broadcasts;
//   * the staged boxes are CHANNEL-CHUNK PLANAR in LDS: plane c (c = 0..7) holds channels 4c..4c+3 of every staged texel as one
//     float4, texels in box order.  The four taps of chunk c are `ds_read_b128` at byte address E + c * PLANE + {0, 16} and
//     E + pitch*16 + c * PLANE + {0, 16}: two address registers per (voxel, view), everything else immediate offsets;
//   * the sweep runs chunk-outer / view-inner: sum and sum of squares of ONE chunk (8 registers) are live at a time; the (chunk, view)
//     steps are software-pipelined (the next step's four taps are requested before the current step is blended);
//   * the staged views of a block, in view order, fill slots 0 .. nv-1 (a view whose box lies outside the image contributes zero and
//     gets no slot); the sweep is straight-line code per nv, no per-view mode tests inside it;
//   * lanes of one LDS pass (16 lanes: the quads of even / odd index parity of a half wave, see warp_cost_tiled.hip) are 8 x-adjacent
//     pixels x 2 adjacent depth planes: their samples fall on ~9 consecutive texels of a box row (consecutive 16-byte slots) or on the
//     same texel (same address), whatever the box pitch -- so rows are not padded (pitch = box width) and boxes up to 32 x 16 texels
//     are staged (bank conflicts: 15 % of the LDS cycles, 11 % in the quad-owner kernel);
//   * a box clipped at the image border is staged WITH its zero padding (two texels beyond the border, written as zeros), so the
//     clipped mode needs no validity masks: the top-left tap index is clamped into the box and everything else is the inside case;
//   * 16-bit stores leave TRANSPOSED across the four 16-lane rows of the wave (v_permlane32_swap / v_permlane16_swap, 16 per trip):
//     64 lanes each writing 16 bytes at the 64-byte voxel stride run at half the rate of a store whose lanes cover whole voxels
//     (scripts/ubench/store_patterns.hip: 72.6 against 36.5 us for the 251 MB volume, whichever lanes hold the pieces);
//   * blocks with a view whose box does not fit, or has a corner behind the camera (WL_DIRECT), run a rolled general path that
//     recomputes positions per chunk and takes per-lane 8-byte global taps for that view: correct, slow (4x a fast block).
//
// Same fp32 operation chain as the direct kernels and the quad-owner kernel (blend t00*w00 -> fma t01 -> fma t10 -> fma t11; sums start
// at the reference feature; variance as fma(1/N, q, -(1/N^2 * (s*s)))), so the stored bits are identical to theirs
// (tests/test_gpu_warp_cost.py::test_lane_owner_kernel_equals_direct_kernel).  Contraction is off for the whole kernel: the compiler
// fused `ix - floor(ix)` with ix = hx * inv_z into one fma on the general path only.  Compiled without packed fp32 instructions like
// warp_cost_tiled.o (Makefile).
//
// Measured (MI355X, config 2, probe rig; scripts/dev/lv_probe.py, profiles/r04_warp_lane_owner.txt): 61.0 M vector-ALU instructions
// against 71.1 M (-14 %), 6.7 M LDS instructions against 6.4 M; interleaved A/B of stand-alone launches 111-115 us against 113-118 us
// (-2 %); in the step 110.9 against 113.7 us, step 0.980 against 0.986 ms.  The sweep runs at ~1.3 ns per vector-ALU instruction
// and SIMD whichever kernel issues it (clock ~1.8 GHz under this load), box + staging + stores alone take 55 us of the launch, and
// neither a second prefetch step, four blocks per CU (LV_OCC = 4: 122 us, its smaller arena sends 5 % of the blocks to the general
// path) nor three-address v_fma_f32 moved it.  On the DTU-like rig 60-90 % of the blocks have a view that does not fit and the
// launch takes 327 us (quad-owner 175, direct gather 160): hence not the default.  Ablations ("warp_tile" = 8 / 9 / 10): without
// stores 103 us; without taps and blend 56 us; box phase + staging (+ the reference-only variance) alone 50 us, i.e. the per-block
// fixed phases are NOT hidden behind the other blocks' sweeps (the quad-owner kernel's box + staging alone: 20 us at four blocks
// per CU).  Starting the first generation of blocks staggered changed nothing (114 us either way).
//
// Semantics and citations are those of warp_cost.hip (reference: models/MVSNet/module.py:130-166, model.py:109-139).
#include <type_traits>

#include "warp_common.h"
#include "warp_lds.h"
#include "warp_lv.h"

namespace pscv {

template <typename TIn, typename TOut, int COST>
__global__ __launch_bounds__(LV_THREADS) __attribute__((amdgpu_waves_per_eu(LV_OCC, LV_OCC))) void warp_cost_lv_kernel(const WarpArgs a) {
    // Every fused multiply-add of this kernel is written as one (fmaf): the compiler must not form others -- it turned `ix - floor(ix)`
    // with ix = hx * inv_z into fma(hx, inv_z, -floor) on the general path only (one rounding less than the other warp kernels:
    // 0.5 % of the stored values moved by one ulp).
#pragma clang fp contract(off)
    constexpr int C = 32, PIXB = 64;
    constexpr int OB = (int)sizeof(TOut);
    static_assert(COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP, "variance costs");
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];

    // ---- work decode (as warp_cost_tiled.hip): XCD k gets a contiguous run of tiles ----
    const int dc = blockIdx.y;
    const int tpx = gridDim.x >> 3;
    const int ntx = (a.w + LV_T - 1) / LV_T, nty = (a.h + LV_TH - 1) / LV_TH;
    const int tile = ((int)blockIdx.x & 7) * tpx + ((int)blockIdx.x >> 3);
    if (tile >= a.B * nty * ntx) return;
    const int trow = (int)(((float)tile + 0.5f) * (1.0f / (float)ntx));     // exact: tile < 2^22
    const int txi = tile - trow * ntx;
    const int b = (int)(((float)trow + 0.5f) * (1.0f / (float)nty));
    const int tyi = trow - b * nty;

    __builtin_amdgcn_s_setprio(3);                                   // box / staging phase ahead of the other blocks' sweeps
    __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1);       // MODE.FP16_OVFL: saturating f32 -> f16 stores
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int x0t = txi * LV_T, y0t = tyi * LV_TH;
    const int d0 = dc * a.ppd, d1 = min(a.D, d0 + a.ppd);
    const float* const depth_b = a.depth + (long)b * a.depth_bstride;
    const int n_src = a.n_src;
    int* const table = reinterpret_cast<int*>(lsm + LV_TABLE);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lsm;   // LDS byte address of the arena

    const float dlane = depth_b[min(d0 + lane, d1 - 1)];             // lane i holds plane d0 + i (<= 64 planes per chunk)

    int prow, pcol, pp;
    lv_voxel_of(lane, prow, pcol, pp);
    int x = x0t + pcol, y = y0t + prow;
    const bool active_px = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int hw = a.h * a.w;
    const int pflat = y * a.w + x;
    const float px = (float)x, py = (float)(y + a.ref_y0);
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

    // ---- 1. wave k: texel boxes of source view k from the 8 corner projections (see warp_cost_tiled.hip for the argument): the box of
    //         the whole chunk (lanes 0-7), of its first half (lanes 8-15) and of its second half (lanes 16-23) -- round 5: a block in
    //         which some view's whole-chunk box does not fit sweeps the two halves one after the other instead of its general path ----
    const int nplanes = d1 - d0;
    const int hsz = nplanes >= 4 ? ((nplanes / 2 + 1) & ~1) : nplanes;        // planes of the first half (even); no split below 4 planes
    if (wave < WL_MAX_SRC) {
        const int k = wave;
        const float inf = __builtin_inff();
        const bool in1 = lane < hsz;
        const float dmin1 = wl_wave_reduce<false>(in1 ? dlane : inf), dmax1 = wl_wave_reduce<true>(in1 ? dlane : -inf);
        const float dmin2 = hsz < nplanes ? wl_wave_reduce<false>(in1 ? inf : dlane) : dmin1;
        const float dmax2 = hsz < nplanes ? wl_wave_reduce<true>(in1 ? -inf : dlane) : dmax1;
        const float dmin0 = fminf(dmin1, dmin2), dmax0 = fmaxf(dmax1, dmax2);
        const int set = min(lane >> 3, 2);
        const int corner = lane & 7;
        const float cx = (corner & 1) ? (float)min(x0t + LV_T - 1, a.w - 1) : (float)x0t;
        const float cy = (float)(((corner & 2) ? min(y0t + LV_TH - 1, a.h - 1) : y0t) + a.ref_y0);
        const float dlo = set == 0 ? dmin0 : set == 1 ? dmin1 : dmin2, dhi = set == 0 ? dmax0 : set == 1 ? dmax1 : dmax2;
        const float d = (corner & 4) ? dhi : dlo;
        int cX0 = 0, cY0 = 0, cX1 = 1, cY1 = 1, pitch = 2, mode = WL_ZERO;      // mode: WL_FAST / WL_GEN here = "if the arena has room"
        if (k < n_src) {
            lv_cf cam = (lv_cf)(a.cams + ((long)k * a.B + b) * PSCV_CAM_FLOATS);
            const float ax = fmaf(cam[1], cy, cam[0] * cx) + cam[2];
            const float ay = fmaf(cam[4], cy, cam[3] * cx) + cam[5];
            const float az = fmaf(cam[7], cy, cam[6] * cx) + cam[8];
            const float hx = fmaf(ax, d, cam[9]), hy = fmaf(ay, d, cam[10]), hz = fmaf(az, d, cam[11]);
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            const float u = hx * inv_z, v = hy * inv_z;
            const float okf = (hz > 1e-6f && fabsf(u) < 1e6f && fabsf(v) < 1e6f) ? 1.0f : 0.0f;   // also rejects NaN
            const float umin = wl_reduce8<false>(u), umax = wl_reduce8<true>(u);               // (per 8-lane group = per plane range)
            const float vmin = wl_reduce8<false>(v), vmax = wl_reduce8<true>(v);
            const bool ok = wl_reduce8<false>(okf) != 0.0f;
            const float sl = 1.0f / 32.0f;     // slack for the per-pixel evaluation's different rounding (maps <= 16384 texels)
            const int X0 = (int)floorf(umin - sl), X1 = (int)floorf(umax + sl) + 1;
            const int Y0 = (int)floorf(vmin - sl), Y1 = (int)floorf(vmax + sl) + 1;
            mode = WL_DIRECT;
            if (ok) {
                const bool outside = X1 < 0 || Y1 < 0 || X0 > a.ws - 1 || Y0 > a.hs - 1;
                const bool inside = X0 >= 0 && Y0 >= 0 && X1 <= a.ws - 1 && Y1 <= a.hs - 1;
                // the staged box carries two texels of zero padding beyond each clipped border: a sample whose top-left tap
                // lies further out is clamped onto the padding, where both its taps of that axis are zero (module.py:160-166)
                cX0 = max(X0, -2); cX1 = min(X1, a.ws + 1); cY0 = max(Y0, -2); cY1 = min(Y1, a.hs + 1);
                const int bw = cX1 - cX0 + 1, bh = cY1 - cY0 + 1;
                pitch = bw;                    // (no padding of the rows: the lanes of an LDS pass read along ONE box row)
                if (outside) mode = WL_ZERO;
                else if (bw <= LV_BOX_W && bh <= LV_BOX_H) mode = inside ? WL_FAST : WL_GEN;
            }
        }
        if (corner == 0 && lane < 24) {
            int4* row = reinterpret_cast<int4*>(table + (set * WL_MAX_SRC + k) * 8);
            row[0] = make_int4(cX0, cY0, cX1, cY1);
            row[1] = make_int4(0, pitch, mode, 0);
        }
    }
    __syncthreads();

    // whole chunk first; with "warp_tile" = 2: if a view would not be staged there (its box is too large, or the arena is full), the two
    // halves one after the other, each with its own boxes and staging.  Round 5, interleaved A/B: DTU-like rig 332 -> 187 us (the
    // quad-owner kernel with the same split: 152 us), probe rig 114 -> 118 us (its 12 % of blocks with one unstaged view are cheaper
    // on the general path than swept twice) -- so the split is OFF by default in this (optional) kernel.
    int nsub = 1;
    if (hsz < nplanes && a.variant == 2) {
        int used = 0;
        bool direct = false;
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            const int4 r0 = *reinterpret_cast<const int4*>(table + k * 8), r1 = *reinterpret_cast<const int4*>(table + k * 8 + 4);
            int mode = k < n_src ? __builtin_amdgcn_readfirstlane(r1.z) : WL_ZERO;
            const int need = __builtin_amdgcn_readfirstlane(r1.y) * (__builtin_amdgcn_readfirstlane(r0.w) - __builtin_amdgcn_readfirstlane(r0.y) + 1);
            if ((mode == WL_FAST || mode == WL_GEN) && used + need > LV_ARENA) mode = WL_DIRECT;
            if (mode == WL_FAST || mode == WL_GEN) used += need;
            direct = direct || mode == WL_DIRECT;
        }
        if (direct) nsub = 2;
    }
    for (int sub = 0; sub < nsub; ++sub) {
    const int tset = (nsub == 2 ? 1 + sub : 0) * WL_MAX_SRC * 8;          // this plane range's records in the table (ints)
    const int s0 = d0 + (nsub == 2 && sub ? hsz : 0), s1 = nsub == 2 && !sub ? d0 + hsz : d1;
    if (sub) {
        __syncthreads();                                 // the first half's sweep is done with the arena
        __builtin_amdgcn_s_setprio(3);
    }

    // ---- 2. every wave: the box records -> scalar registers; arena allocation greedy in view order (a view whose box does not fit next
    //         to the earlier ones is not staged), the same in every wave; the staged views, in view order, fill slots 0 .. nv-1 ----
    int bMode[WL_MAX_SRC], bP[WL_MAX_SRC], bE0[WL_MAX_SRC], bBase[WL_MAX_SRC];
    int nv = 0, sView[WL_MAX_SRC] = {0, 0, 0, 0}, sP[WL_MAX_SRC] = {0, 0, 0, 0}, sE0[WL_MAX_SRC] = {0, 0, 0, 0};
    bool sGen[WL_MAX_SRC] = {false, false, false, false}, any_direct = a.variant == 7;     // ("warp_tile" = 7: every block on the general path, a test aid)
    {
        int used = 0;
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            const int4 r0 = *reinterpret_cast<const int4*>(table + tset + k * 8), r1 = *reinterpret_cast<const int4*>(table + tset + k * 8 + 4);
            const int X0 = __builtin_amdgcn_readfirstlane(r0.x), Y0 = __builtin_amdgcn_readfirstlane(r0.y);
            const int Y1 = __builtin_amdgcn_readfirstlane(r0.w);
            bP[k] = __builtin_amdgcn_readfirstlane(r1.y);
            int mode = k < n_src ? __builtin_amdgcn_readfirstlane(r1.z) : WL_ZERO;
            const int need = bP[k] * (Y1 - Y0 + 1);
            if ((mode == WL_FAST || mode == WL_GEN) && used + need > LV_ARENA) mode = WL_DIRECT;
            bBase[k] = used;
            if (mode == WL_FAST || mode == WL_GEN) used += need;
            bMode[k] = mode;
            if (a.mode_hist && k < n_src && tid == 0) atomicAdd(a.mode_hist + k * 4 + mode, 1);
            bE0[k] = ((bBase[k] - Y0 * bP[k] - X0) << 4) + (int)lds0;      // LDS byte address of texel (x, y), chunk 0 = (y * pitch + x) * 16 + bE0
            any_direct = any_direct || mode == WL_DIRECT;
            if (mode == WL_FAST || mode == WL_GEN) {
#pragma unroll
                for (int jj = 0; jj < WL_MAX_SRC; ++jj)
                    if (jj == nv) { sView[jj] = k; sP[jj] = bP[k]; sE0[jj] = bE0[k]; sGen[jj] = mode == WL_GEN; }
                ++nv;
            }
        }
    }
    if (a.variant == 9 || a.variant == 10) nv = 0;     // ("warp_tile" = 9: no taps, no blend -- box phase, staging, reference variance and stores only; 10: nor stores; ablations)

    // ---- 3. stage the boxes, 16-bit -> fp32, channel-chunk planar: wave k stages view k ----
    {
        const int k = wave;
        const int4 f0 = *reinterpret_cast<const int4*>(table + tset + k * 8);
        const int sX0 = __builtin_amdgcn_readfirstlane(f0.x), sY0 = __builtin_amdgcn_readfirstlane(f0.y);
        const int sX1 = __builtin_amdgcn_readfirstlane(f0.z), sY1 = __builtin_amdgcn_readfirstlane(f0.w);
        int sP16 = bP[0] << 4, sMode = bMode[0], sBase = bBase[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) { sP16 = bP[t] << 4; sMode = bMode[t]; sBase = bBase[t]; }
        sBase = (sBase << 4) - sY0 * sP16 - (sX0 << 4);
        const void* srcp = a.src[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) srcp = a.src[t];
        if (k < n_src && (sMode == WL_FAST || sMode == WL_GEN)) {
            const int bw = sX1 - sX0 + 1, bh = sY1 - sY0 + 1;              // bw <= 32, bh <= 16
            const long rstride = (long)a.ws * C;
            // a batch = 8 rows x 16 texels (a lane = one 16-byte piece of a row: texel lane >> 2, channels 8 (lane & 3) ..), loads first
            for (int yh = 0; yh < bh; yh += 8) {
                for (int xh = 0; xh < bw; xh += 16) {
                    const int cw = min(bw - xh, 16), ch = min(bh - yh, 8);
                    const int cl = min(lane, cw * 4 - 1);
                    const bool mine = lane < cw * 4;
                    const int gx = sX0 + xh + (cl >> 2);
                    const bool vx = (unsigned)gx < (unsigned)a.ws;
                    const TIn* col = reinterpret_cast<const TIn*>(srcp) + ((long)b * a.hs * a.ws + min(max(gx, 0), a.ws - 1)) * C + (cl & 3) * 8;
                    const int dst0 = gx * 16 + sBase + (cl & 3) * 2 * LV_PLANE;
                    uint4 val[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int gy = min(max(sY0 + yh + min(i, ch - 1), 0), a.hs - 1);
                        val[i] = *reinterpret_cast<const uint4*>(col + gy * rstride);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (mine && i < ch) {
                            const bool v = vx && (unsigned)(sY0 + yh + i) < (unsigned)a.hs;
                            const uint4 u = val[i];
                            float4 lo = make_float4(Half16<TIn>::lo(u.x), Half16<TIn>::hi(u.x), Half16<TIn>::lo(u.y), Half16<TIn>::hi(u.y));
                            float4 hi = make_float4(Half16<TIn>::lo(u.z), Half16<TIn>::hi(u.z), Half16<TIn>::lo(u.w), Half16<TIn>::hi(u.w));
                            if (!v) { lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f); hi = lo; }
                            const int dst = dst0 + (sY0 + yh + i) * sP16;
                            *reinterpret_cast<float4*>(lsm + dst) = lo;
                            *reinterpret_cast<float4*>(lsm + dst + LV_PLANE) = hi;
                        }
                    }
                }
            }
        }
    }

    // depth-independent ray terms rot (x, y, 1) of this pixel: per slot (fast path) -- module.py:138-144
    float rx[WL_MAX_SRC], ry[WL_MAX_SRC], rz[WL_MAX_SRC];
    lv_cf camp[WL_MAX_SRC];
#pragma unroll
    for (int jj = 0; jj < WL_MAX_SRC; ++jj) {
        camp[jj] = (lv_cf)(a.cams + ((long)sView[jj] * a.B + b) * PSCV_CAM_FLOATS);
        lv_cf cam = camp[jj];
        rx[jj] = fmaf(cam[1], py, cam[0] * px) + cam[2];
        ry[jj] = fmaf(cam[4], py, cam[3] * px) + cam[5];
        rz[jj] = fmaf(cam[7], py, cam[6] * px) + cam[8];
    }

    // 16-bit stores go out transposed (lv_row_transpose): store v of a trip writes, from lane (row r, column c), piece r (16 bytes) of
    // the voxel of lane 16 v + c -- the four pieces of a voxel leave in ONE instruction.  (64 lanes writing 16 bytes each at the
    // 64-byte voxel stride run at HALF the rate: scripts/ubench/store_patterns.hip, 72.6 against 36.5 us for the volume.)
    unsigned st_off[4];       // byte offset of that voxel's piece in a depth plane
    int st_pp = 0, st_act = 0;   // bit v: plane parity / pixel inside the image
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        int vr, vc, vp;
        lv_voxel_of(16 * v + (lane & 15), vr, vc, vp);
        const int vx = x0t + vc, vy = y0t + vr;
        st_act |= (vx < a.w && vy < a.h) ? 1 << v : 0;
        st_pp |= vp << v;
        st_off[v] = (unsigned)(min(vy, a.h - 1) * a.w + min(vx, a.w - 1)) * (C * 2) + (unsigned)(lane >> 4) * 16u;
    }
    if (a.variant == 8 || a.variant == 10) st_act = 0;     // ("warp_tile" = 8: no stores, an ablation)
    const float invN = 1.0f / (float)(n_src + 1);
    const float invN2 = 1.0f / ((float)(n_src + 1) * (float)(n_src + 1));
    char* const out = reinterpret_cast<char*>(a.out);
    const unsigned long plane_bytes = (unsigned long)hw * C * OB;
    const unsigned lane_out = (unsigned)pflat * (C * OB);
    const unsigned long img_bytes = (unsigned long)b * a.hs * a.ws * PIXB;
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);

    // variance of one chunk from its sums, rounded to fp32 and then to the stored format, 16 bytes per two chunks
    uint32_t piece[4][4];      // [dword][piece]: the trip's 64 output bytes of this lane's voxel (16-bit formats)
    auto finish = [&](int c, const float (&s)[4], const float (&q)[4], char* vox, bool active) {
        float o[4];
        if (COST == PSCV_COST_VARIANCE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaf(invN, q[i], -__fmul_rn(invN2, __fmul_rn(s[i], s[i])));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float m = __fmul_rn(invN, s[i]); o[i] = fmaf(invN, q[i], -__fmul_rn(m, m)); }
        }
        // (no v_fma_mixlo_f16 fusion of the last fma and the conversion: one rounding instead of two; warp_cost_tiled.hip)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(o[i]));
        if constexpr (OB == 4) {
            if (active) *reinterpret_cast<float4*>(vox + c * 16) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            piece[2 * (c & 1)][c >> 1] = wl_pack2<TOut>(o[0], o[1]);
            piece[2 * (c & 1) + 1][c >> 1] = wl_pack2<TOut>(o[2], o[3]);
        }
    };
    // the trip's four transposed stores (16-bit formats); plane0 / plane1: the two depth planes of the trip
    auto store_trip = [&](char* plane0, char* plane1, bool has1) {
        if constexpr (OB == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lv_row_transpose(piece[k]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool odd = (st_pp >> v) & 1;
                if (((st_act >> v) & 1) && (has1 || !odd))
                    *reinterpret_cast<uint4*>((odd ? plane1 : plane0) + st_off[v]) = make_uint4(piece[0][v], piece[1][v], piece[2][v], piece[3][v]);
            }
        }
    };

    // ---- 4. sweep: a trip = the wave's 32 pixels on two adjacent planes ----
    const int nd = s1 - s0, sl0 = s0 - d0;
    for (int t = wave; 2 * t < nd; t += LV_THREADS / 64) {
        const float dv0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), sl0 + 2 * t));
        const float dv1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), sl0 + min(2 * t + 1, nd - 1)));
        const float dval = pp ? dv1 : dv0;
        // (keeps r * r of the loop-invariant reference feature out of 32 more registers)
#pragma unroll
        for (int i = 0; i < C; ++i) asm volatile("" : "+v"(rf[i]));
        const bool active = active_px && 2 * t + pp < nd && a.variant != 8 && a.variant != 10;     // ("warp_tile" = 8: no stores, an ablation)
        const int d = s0 + min(2 * t + pp, nd - 1);
        char* const vox = out + ((unsigned long)b * a.D + d) * plane_bytes + lane_out;

        if (!any_direct) {
            // ---- fast path: every view of the block is staged or contributes zero; straight-line code per count of staged views ----
            auto chunks = [&](auto nv_c) {
                constexpr int NV = decltype(nv_c)::value;
                float w[NV > 0 ? NV : 1][4];
                unsigned aT[NV > 0 ? NV : 1], aB[NV > 0 ? NV : 1];      // LDS byte address of the top / bottom tap row in chunk plane 0
#pragma unroll
                for (int jj = 0; jj < NV; ++jj) {
                    lv_cf cam = camp[jj];
                    const float hx = fmaf(rx[jj], dval, cam[9]), hy = fmaf(ry[jj], dval, cam[10]), hz = fmaf(rz[jj], dval, cam[11]);
                    const float inv_z = __builtin_amdgcn_rcpf(hz);
                    const float ix = hx * inv_z, iy = hy * inv_z;
                    const float x0f = floorf(ix), y0f = floorf(iy);
                    const float fx = ix - x0f, fy = iy - y0f;
                    const float gx = 1.0f - fx, gy = 1.0f - fy;
                    w[jj][0] = gx * gy; w[jj][1] = fx * gy; w[jj][2] = gx * fy; w[jj][3] = fx * fy;
                    int x0 = (int)x0f, y0 = (int)y0f;
                    if (sGen[jj]) {      // top-left tap into the box (with its zero padding); the sample may lie anywhere
                        const int4 r0 = *reinterpret_cast<const int4*>(table + tset + sView[jj] * 8);
                        x0 = med3_i32(x0, r0.x, r0.z - 1); y0 = med3_i32(y0, r0.y, r0.w - 1);
                    }
                    aT[jj] = (unsigned)((__mul24(y0, sP[jj]) + x0) * 16 + sE0[jj]);
                    aB[jj] = aT[jj] + (unsigned)(sP[jj] << 4);
                }
                if constexpr (NV == 0) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float s[4], q[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const float r = rf[4 * c + i]; s[i] = r; q[i] = r * r; }
                        finish(c, s, q, vox, active);
                    }
                } else {
                    // software pipeline over the (chunk, view) steps: the four taps of the next step are requested before the
                    // current step is blended (two register buffers)
                    constexpr int PF = 1;        // steps requested ahead (PF + 1 register buffers; 2 ahead: 168 registers with spills, same time)
                    wl_f4 T[PF + 1][4];
                    auto request = [&](int buf, int c, int jj) {
                        T[buf][0] = lv_tap(aT[jj], c * LV_PLANE); T[buf][1] = lv_tap(aT[jj], c * LV_PLANE + 16);
                        T[buf][2] = lv_tap(aB[jj], c * LV_PLANE); T[buf][3] = lv_tap(aB[jj], c * LV_PLANE + 16);
                    };
                    float s[4], q[4], wv[4];
#pragma unroll
                    for (int p = 0; p < PF; ++p) request(p, p / NV, p % NV);
#pragma unroll
                    for (int step = 0; step < 8 * NV; ++step) {
                        const int c = step / NV, jj = step % NV;
                        if (step + PF < 8 * NV) request((step + PF) % (PF + 1), (step + PF) / NV, (step + PF) % NV);
                        if (jj == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) { const float r = rf[4 * c + i]; s[i] = r; q[i] = r * r; }      // the sums start at the reference feature  model.py:121-123
                        }
                        lv_blend(T[step % (PF + 1)][0], T[step % (PF + 1)][1], T[step % (PF + 1)][2], T[step % (PF + 1)][3], w[jj], wv);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { s[i] += wv[i]; q[i] = fmaf(wv[i], wv[i], q[i]); }
                        if (jj == NV - 1) finish(c, s, q, vox, active);
                    }
                }
            };
            if (nv == 4) chunks(std::integral_constant<int, 4>{});
            else if (nv == 3) chunks(std::integral_constant<int, 3>{});
            else if (nv == 2) chunks(std::integral_constant<int, 2>{});
            else if (nv == 1) chunks(std::integral_constant<int, 1>{});
            else chunks(std::integral_constant<int, 0>{});
            {
                char* const plane0 = out + ((unsigned long)b * a.D + s0 + 2 * t) * plane_bytes;
                store_trip(plane0, plane0 + plane_bytes, 2 * t + 1 < nd);
            }
        } else {
            // ---- general path (a view of the block takes global taps): per view and chunk, nothing kept across chunks ----
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                float s[4], q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float r = rf[0];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) r = c == cc ? rf[4 * cc + i] : r;
                    s[i] = r; q[i] = r * r;
                }
#pragma unroll 1
                for (int k = 0; k < n_src; ++k) {
                    const int mode = k == 0 ? bMode[0] : k == 1 ? bMode[1] : k == 2 ? bMode[2] : bMode[3];
                    if (mode == WL_ZERO) continue;
                    const float* cam = a.cams + ((long)k * a.B + b) * PSCV_CAM_FLOATS;
                    wl_f4 t00, t01, t10, t11;
                    float ww[4];
                    if (mode != WL_DIRECT) {
                        const int P = k == 0 ? bP[0] : k == 1 ? bP[1] : k == 2 ? bP[2] : bP[3];
                        const int E0 = k == 0 ? bE0[0] : k == 1 ? bE0[1] : k == 2 ? bE0[2] : bE0[3];
                        const float rxk = fmaf(cam[1], py, cam[0] * px) + cam[2];
                        const float ryk = fmaf(cam[4], py, cam[3] * px) + cam[5];
                        const float rzk = fmaf(cam[7], py, cam[6] * px) + cam[8];
                        const float hx = fmaf(rxk, dval, cam[9]), hy = fmaf(ryk, dval, cam[10]), hz = fmaf(rzk, dval, cam[11]);
                        const float inv_z = __builtin_amdgcn_rcpf(hz);
                        const float ix = hx * inv_z, iy = hy * inv_z;
                        const float x0f = floorf(ix), y0f = floorf(iy);
                        const float fx = ix - x0f, fy = iy - y0f;
                        const float gx = 1.0f - fx, gy = 1.0f - fy;
                        ww[0] = gx * gy; ww[1] = fx * gy; ww[2] = gx * fy; ww[3] = fx * fy;
                        const int4 r0 = *reinterpret_cast<const int4*>(table + tset + k * 8);
                        const int x0 = med3_i32((int)x0f, r0.x, r0.z - 1), y0 = med3_i32((int)y0f, r0.y, r0.w - 1);   // (no-op for boxes inside the image)
                        const unsigned at = (unsigned)((__mul24(y0, P) + x0) * 16 + E0 + c * LV_PLANE);
                        const unsigned ab = at + (unsigned)(P << 4);
                        t00 = lv_tap(at, 0); t01 = lv_tap(at, 16); t10 = lv_tap(ab, 0); t11 = lv_tap(ab, 16);
                    } else {
                        // behind-camera test, grid clamp, zero padding  module.py:146-166
                        float ix, iy;
                        sweep_index<PSCV_GEOM_PROJ>(cam, px, py, dval, a, ix, iy);
                        const float x0f = floorf(ix), y0f = floorf(iy);
                        Taps tp;
                        make_taps<false, PIXB>(ix - x0f, iy - y0f, (int)x0f, (int)y0f, a.hs, a.ws, (unsigned)c * 8u, tp);
                        ww[0] = tp.w00; ww[1] = tp.w01; ww[2] = tp.w10; ww[3] = tp.w11;
                        const char* img = reinterpret_cast<const char*>(k == 0 ? a.src[0] : k == 1 ? a.src[1] : k == 2 ? a.src[2] : a.src[3]) + img_bytes;
                        const uint2 g00 = *reinterpret_cast<const uint2*>(img + tp.o00), g01 = *reinterpret_cast<const uint2*>(img + tp.o01);
                        const uint2 g10 = *reinterpret_cast<const uint2*>(img + tp.o10), g11 = *reinterpret_cast<const uint2*>(img + tp.o11);
                        t00 = wl_f4{Half16<TIn>::lo(g00.x), Half16<TIn>::hi(g00.x), Half16<TIn>::lo(g00.y), Half16<TIn>::hi(g00.y)};
                        t01 = wl_f4{Half16<TIn>::lo(g01.x), Half16<TIn>::hi(g01.x), Half16<TIn>::lo(g01.y), Half16<TIn>::hi(g01.y)};
                        t10 = wl_f4{Half16<TIn>::lo(g10.x), Half16<TIn>::hi(g10.x), Half16<TIn>::lo(g10.y), Half16<TIn>::hi(g10.y)};
                        t11 = wl_f4{Half16<TIn>::lo(g11.x), Half16<TIn>::hi(g11.x), Half16<TIn>::lo(g11.y), Half16<TIn>::hi(g11.y)};
                    }
                    float wv[4];
                    lv_blend(t00, t01, t10, t11, ww, wv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { s[i] += wv[i]; q[i] = fmaf(wv[i], wv[i], q[i]); }
                }
                // (rolled loop: the chunk index of the store is a run-time value)
                float o[4];
                if (COST == PSCV_COST_VARIANCE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaf(invN, q[i], -__fmul_rn(invN2, __fmul_rn(s[i], s[i])));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float m = __fmul_rn(invN, s[i]); o[i] = fmaf(invN, q[i], -__fmul_rn(m, m)); }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(o[i]));
                if constexpr (OB == 4) {
                    if (active) *reinterpret_cast<float4*>(vox + c * 16) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    if (active) *reinterpret_cast<uint2*>(vox + c * 8) = make_uint2(wl_pack2<TOut>(o[0], o[1]), wl_pack2<TOut>(o[2], o[3]));
                }
            }
        }
    }
    }   // plane sub-range
}

template <typename TIn, typename TOut, int COST>
static int lv_launch(const WarpArgs& a, hipStream_t st) {
    auto kern = warp_cost_lv_kernel<TIn, TOut, COST>;
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LV_LDS);
        if (e != hipSuccess) { set_error("pscv_warp_cost(lv): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    const int tiles = a.B * ((a.h + LV_TH - 1) / LV_TH) * ((a.w + LV_T - 1) / LV_T);
    hipLaunchKernelGGL(kern, dim3(8 * ((tiles + 7) / 8), a.n_dchunks), dim3(LV_THREADS), LV_LDS, st, a);
    return 0;
}

template <typename TIn, typename TOut>
static int lv_dispatch(const WarpArgs& a, int cost, hipStream_t st) {
    if (cost == PSCV_COST_VARIANCE) return lv_launch<TIn, TOut, PSCV_COST_VARIANCE>(a, st);
    if (cost == PSCV_COST_VARIANCE_CVP) return lv_launch<TIn, TOut, PSCV_COST_VARIANCE_CVP>(a, st);
    return 1;
}

extern int* g_wl_mode_hist;   // warp_cost_tiled.hip (pscv_debug_wl_mode_hist)
extern Knob g_warp_tile;       // warp_cost.hip

// Returns 0 if launched, 1 if this configuration is not covered (the caller tries the quad-owner kernel next), negative on error.
int warp_cost_lv_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || a.depth_per_pixel || geom != PSCV_GEOM_PROJ || (in_dtype != PSCV_F16 && in_dtype != PSCV_BF16)) return 1;
    if (cost != PSCV_COST_VARIANCE && cost != PSCV_COST_VARIANCE_CVP) return 1;
    if (out_dtype != in_dtype && out_dtype != PSCV_F32) return 1;
    if (a.n_src < 1 || a.n_src > WL_MAX_SRC) return 1;
    if (a.ws > 16384 || a.hs > 16384) return 1;
    const long tiles = (long)a.B * ((a.h + LV_TH - 1) / LV_TH) * ((a.w + LV_T - 1) / LV_T);
    if (tiles >= (1L << 22)) return 1;   // tile index decode is exact below 2^22
    int ppd = ppd_override > 0 ? min((ppd_override + 1) & ~1, 64) : 32;   // planes per block: amortises the box staging
    while (ppd > 4 && tiles * ((a.D + ppd - 1) / ppd) < 1024) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    a.mode_hist = g_wl_mode_hist;
    a.variant = g_warp_tile;
    const long nblk = tiles * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(lv): bad grid %ld", nblk); return -1; }
    if (in_dtype == PSCV_F16) return out_dtype == PSCV_F32 ? lv_dispatch<f16_t, float>(a, cost, st) : lv_dispatch<f16_t, f16_t>(a, cost, st);
    return out_dtype == PSCV_F32 ? lv_dispatch<bf16_t, float>(a, cost, st) : lv_dispatch<bf16_t, bf16_t>(a, cost, st);
}

}  // namespace pscv
