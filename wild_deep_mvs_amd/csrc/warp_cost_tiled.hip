// LDS-staged plane-sweep warp + cost for 32-channel 16-bit feature maps (gfx950), second design (round 2).
//
// What bounds the sweep (profiles/README.md, scripts/ubench/valu_rate2.hip): the direct-gather kernels are limited by
// vector-ALU issue and by the per-CU L1 tap rate, not by HBM (their traffic is already the algorithmic minimum).
// Measured issue rates on MI355X: plain fp32 `v_fma_f32` / `v_add_f32` / `v_and_b32` are full rate; EVERY 16-bit form
// (`v_fma_mix_f32`, `v_pk_fma_f16`, `v_dot2_f32_f16|bf16`, `v_cvt_f32_f16`, `v_lshlrev_b32`, `v_perm_b32`) and every DPP
// move is half rate; `ds_read_b128` moves 256 B/clk/CU and overlaps with the vector ALU.  So:
//
//   * the source patches a tile of 8 x 8 reference pixels can touch on a chunk of depth planes are staged in LDS ONCE per
//     (tile, chunk, view) and CONVERTED TO FP32 while they are staged (conversion cost per staged texel, ~14x fewer than
//     taps; boxes from the 8 corner projections of the tile at the chunk's depth extremes);
//   * every bilinear tap is then two `ds_read_b128` (8 channels per lane) and the blend is four fp32 FMAs per channel --
//     the same fp32 operation chain as the direct kernels (`fmaf(float(h), w, acc)` == `v_fma_mix_f32`), so the results are
//     bit-identical to theirs (tests/test_gpu_warp_cost.py::test_tiled_kernel_equals_direct_kernel);
//   * a quad of lanes owns a voxel (lane l: channels 8l..8l+7) and lane l computes the sample position in source view l,
//     so the coordinate arithmetic runs once per four voxel-views; weights and the tap address travel through the quad
//     with DPP `quad_perm` broadcasts; the ray terms rot (x, y, 1) of (pixel, view l) are lane constants of the sweep;
//   * per (block, view) staging mode: box strictly inside the source image -> no validity masks, clamps or behind-camera
//     test at all; box clipped at the border -> general zero-padding weights, taps clamped into the staged box; box
//     entirely outside -> the view contributes f = 0 and is skipped (variance) -- 19 % of the voxel-views of the bench
//     scene; a corner at / behind the camera or a box that does not fit -> direct global taps for that view;
//   * adaptive split (round 5): the box phase computes, per view, the box of the whole 32-plane chunk AND of its two halves; a
//     block in which some view's whole-chunk box does not fit (wide baselines: on the DTU-like rig a sample travels 0.3 texels
//     per plane) sweeps the halves one after the other, each with its own staging phase, instead of taking global taps:
//     DTU-like rig 176 -> 150 us (global-tap share of the (block, view) pairs 43 % -> 15 %), probe rig unchanged (122 -> 119 us),
//     same stored bits (scripts/dev/warp_ab.py, interleaved);
//   * one wave computes the boxes (cameras through the scalar cache) and publishes them through a 128-byte LDS table;
//     the staging loads of a wave are one branch-free batch; this short phase runs at raised wave priority because the
//     older workgroup of the CU, in its vector-ALU-bound sweep, would otherwise win every issue slot.
//
// Occupancy: 256 threads = 8 x 4 pixels x 32 planes, all (<= 4) source views resident: 318 texels x 128 B fp32 = 40 KiB
// -> four workgroups (16 waves) per CU, so three sweep while one computes its boxes and stages (WL_TILE_H = 8: 512 threads,
// 8 x 8 pixels, 80 KiB, two per CU -- 3-6 % slower: fewer staged texels per voxel but the staging phase is covered worse).  LDS layout of a texel: "lo" plane holds channels {8l..8l+3 : l = 0..3} (64 B),
// "hi" plane {8l+4..8l+7}; a quad reads 64 contiguous bytes per instruction; the quads of one `ds_read_b128` lane group
// are four x-adjacent pixels and the box pitch is a multiple of 4, so a group hits four distinct 64-byte bank groups.
//
// Measured (profiles/README.md): 115 us against 166 us for the quad kernel inside the headline step (f16; bf16 alike),
// 48 M vector-ALU instructions against 81 M, L1 tap traffic 8.8 M accesses against 67.9 M.  The sweep is now bound by
// vector-ALU issue (~70 % busy; ~165 instructions per 16 voxels x 4 views, 77 of them the blend and the two sums).
// Tried and not kept: software-pipelining the views inside a wave (next view's LDS reads issued between the channel
// halves of the current blend: 0.71x the quad kernel's time against 0.63x), per-wave redundant box computation.
//
// Semantics and citations are those of warp_cost.hip.
#include <type_traits>

#include "warp_common.h"
#include "warp_lds.h"

// Built WITHOUT packed fp32 instructions (Makefile: -target-feature -packed-fp32-ops): v_pk_{mul,add,fma}_f32 with the op_sel bit of src1 set
// return wrong low results in lanes 48-63 beside MFMA waves of another stream on this MI355X pool; the SLP vectorizer
// emits that form for the blend below.  scripts/lint_isa.py keeps the form out of the library.
namespace pscv {

#ifndef WL_TILE_H
#define WL_TILE_H 4
#endif
constexpr int WL_T = 8;                      // tile width in reference pixels
constexpr int WL_TH = WL_TILE_H;             // tile height: 8 (512 threads, two workgroups per CU) or 4 (256 threads, four per CU)
constexpr int WL_THREADS = 64 * WL_TH;       // a wave = 2 tile rows x 8 pixels; (WL_TH / 2) pixel groups x 2 plane parities
constexpr int WL_PG = WL_TH / 2;             // pixel groups (waves per plane parity)
constexpr int WL_ARENA = WL_TH == 8 ? 636 : 316;   // staged texels per block (all views): 80 / 40 KiB of fp32
constexpr int WL_HI = WL_ARENA * 64;         // byte offset of the "hi" channel plane
constexpr int WL_STAGE_ROWS = 8;             // box rows one wave stages (one load batch)
constexpr int WL_BOX_H = WL_STAGE_ROWS * (WL_THREADS / 64 / WL_MAX_SRC);   // tallest box the staging phase covers: 8 (4-row tile) / 16
constexpr int WL_NSETS = 7;                  // plane ranges with their own boxes: the whole chunk, its two halves (round 5), its four quarters (round 6)
constexpr int WL_TABLE = 2 * WL_HI;          // box records (16 B: X0 | Y0, X1 | Y1, pitch | mode as 16-bit pairs), [7 plane ranges][4 views]:
                                             // range 0 = the whole chunk, 1 / 2 = its first / second half, 3..6 = the quarters (adaptive split)
constexpr int WL_LDS = WL_TABLE + WL_NSETS * WL_MAX_SRC * 16 + 32;
static_assert(WL_LDS <= (WL_TH == 8 ? 81920 : 40960), "two / four blocks per CU");


// eight fp32 channels x four taps -> eight blended channels; t = {00lo, 00hi, 01lo, 01hi, 10lo, 10hi, 11lo, 11hi}
__device__ __forceinline__ void wl_blend8(const wl_f4 (&t)[8], const float (&w)[4], float (&o)[8]) {
    o[0] = t[0].x * w[0]; o[1] = t[0].y * w[0]; o[2] = t[0].z * w[0]; o[3] = t[0].w * w[0];
    o[4] = t[1].x * w[0]; o[5] = t[1].y * w[0]; o[6] = t[1].z * w[0]; o[7] = t[1].w * w[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        o[0] = fmaf(t[2 * k].x, w[k], o[0]); o[1] = fmaf(t[2 * k].y, w[k], o[1]);
        o[2] = fmaf(t[2 * k].z, w[k], o[2]); o[3] = fmaf(t[2 * k].w, w[k], o[3]);
        o[4] = fmaf(t[2 * k + 1].x, w[k], o[4]); o[5] = fmaf(t[2 * k + 1].y, w[k], o[5]);
        o[6] = fmaf(t[2 * k + 1].z, w[k], o[6]); o[7] = fmaf(t[2 * k + 1].w, w[k], o[7]);
    }
}

__device__ __forceinline__ wl_f4 wl_tap(const unsigned char* lsm, unsigned off) { return *reinterpret_cast<const wl_f4*>(lsm + off); }

template <typename TOut> __device__ __forceinline__ void wl_store8(char* p, const float (&o)[8]) {
    if constexpr (sizeof(TOut) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p + 16) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(wl_pack2<TOut>(o[0], o[1]), wl_pack2<TOut>(o[2], o[3]),
                                                   wl_pack2<TOut>(o[4], o[5]), wl_pack2<TOut>(o[6], o[7]));
    }
}

// (The software-pipelined sweep of round 3 -- next view's taps requested out of a second register buffer before the current view is
//  blended; 164 VGPRs = three workgroups per CU; 2 % faster in the same binary, 5 % slower than the plain loop at 101 VGPRs -- was removed
//  from this file in round 4; commit cb456f1 and docs/DESIGN_notes_r1-r3.md hold the code and the measurement.)

template <typename TIn, typename TOut, int GEOM, int COST>
__global__ __launch_bounds__(WL_THREADS, 4) void warp_cost_lds_kernel(const WarpArgs a) {
    constexpr int C = 32, PIXB = 64;
    constexpr int OB = (int)sizeof(TOut);
    constexpr bool VAR = COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP;
    static_assert(GEOM == PSCV_GEOM_PROJ, "PROJ geometry (three depth-independent ray terms per (view, pixel))");
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];

    // ---- work decode: grid = (8 x tiles-per-XCD, depth chunks).  Hardware places consecutive workgroups on consecutive XCDs,
    //      so blockIdx.x & 7 is the XCD: give XCD k a contiguous run of tiles (its source footprint stays inside that XCD's
    //      4 MiB L2; the depth chunks of a tile re-read nearly the same texels).  Float reciprocals replace integer division. ----
    const int dc = blockIdx.y;
    const int tpx = gridDim.x >> 3;
    const int ntx = (a.w + WL_T - 1) / WL_T, nty = (a.h + WL_TH - 1) / WL_TH;
    const int tile = ((int)blockIdx.x & 7) * tpx + ((int)blockIdx.x >> 3);
    if (tile >= a.B * nty * ntx) return;
    const int trow = (int)(((float)tile + 0.5f) * (1.0f / (float)ntx));     // exact: tile < 2^22
    const int txi = tile - trow * ntx;
    const int b = (int)(((float)trow + 0.5f) * (1.0f / (float)nty));
    const int tyi = trow - b * nty;

    // the box / staging phase of a new workgroup runs at raised priority: the (older) waves of the CU's other workgroup are in
    // their vector-ALU-bound sweep and would otherwise win every issue slot (arbitration is priority, then age), stretching
    // this short phase -- and with it the time the CU runs on one workgroup's waves only -- to ~25 000 cycles
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1);     // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: saturating f32 -> f16 stores
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int x0t = txi * WL_T, y0t = tyi * WL_TH;
    const int d0 = dc * a.ppd, d1 = min(a.D, d0 + a.ppd);
    const float* const depth_b = a.depth + (long)b * a.depth_bstride;
    const int n_src = a.n_src;
    int* const table = reinterpret_cast<int*>(lsm + WL_TABLE);

    // depth planes of the chunk: lane i holds plane d0 + i (<= 64 planes per chunk)
    const float dlane = depth_b[min(d0 + lane, d1 - 1)];

    // per-lane constants of the sweep (loads issued before the box phase): lane l of a quad owns channels 8l..8l+7 of the
    // quad's voxel and computes the sample position in source view l
    const int quad = lane >> 2, l = lane & 3;
    // pixel of the tile.  A ds_read_b128 is served in groups of 16 lanes = the quads {0,3,5,6} / {1,2,4,7} (+8) of a wave:
    // give each group four x-ADJACENT pixels, whose samples fall on at most four consecutive (or identical) texels of a
    // row = four distinct 64-byte bank groups, for any source scale <= 4/3 (quads of a tile row in natural order collide
    // as soon as the scale is below 1: columns 0 3 5 6 -> texels 0 2 4 5)
    const int pl = (2 * (wave % WL_PG) + (quad >> 3)) * 8 + ((0x73261540u >> (4 * (quad & 7))) & 7);
    int x = x0t + (pl & 7), y = y0t + (pl >> 3);
    const bool active = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int hw = a.h * a.w;
    const int pflat = y * a.w + x;
    const float px = (float)x, py = (float)(y + a.ref_y0);
    const unsigned chb = (unsigned)l * 16u;
    float rx, ry, rz, tx, ty_, tz;     // view l: depth-independent ray terms rot (x, y, 1) and the translation
    {
        const float* cam = a.cams + ((long)min(l, n_src - 1) * a.B + b) * PSCV_CAM_FLOATS;
        rx = fmaf(cam[1], py, cam[0] * px) + cam[2];
        ry = fmaf(cam[4], py, cam[3] * px) + cam[5];
        rz = fmaf(cam[7], py, cam[6] * px) + cam[8];
        tx = cam[9]; ty_ = cam[10]; tz = cam[11];
    }
    float rf[8], rf2[8];     // reference feature and its square: what the two sums of the variance start from
    {
        const f32x8 t = Elem<TIn>::load8(reinterpret_cast<const TIn*>(a.ref) + ((long)b * hw + pflat) * C + l * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rf[j] = t.v[j]; rf2[j] = t.v[j] * t.v[j]; }
    }

    // ---- 1. wave 0: texel box per view from the 8 corner projections (tile corners x depth extremes of the chunk): for a
    //         fixed plane the warp is a homography (convex sets stay convex while z > 0), for a fixed pixel the sample moves
    //         monotonically along its epipolar line, so every sample of the (tile, chunk) lies in the bounding box of these
    //         8 points.  The camera blocks come through the SCALAR cache (constant address space: the vector memory path is
    //         busy with the cost-volume stores of the other waves and answers in thousands of cycles); lanes = corners;
    //         min / max by DPP; arena allocation greedy in view order, all in scalar registers. ----
    // Round 5: THREE boxes per view in the same phase -- the whole chunk (lanes 0-7 = its 8 corners), its first half (lanes 8-15) and
    // its second half (lanes 16-23).  If a view's whole-chunk box does not fit (wide baselines: the sample travels 0.3 texels per
    // plane on DTU-like rigs, 10 texels over 32 planes) the block sweeps the two halves one after the other, each with its own, half
    // as long boxes, instead of taking global taps for that view: no extra box phase, one more staging phase for such blocks only.
    // Round 6: SEVEN boxes per view -- the four quarters too (lanes 24-55).  On the DTU-like rig half of the 16-plane ranges of the
    // widest-baseline view still took global taps (boxes 15-18 texels wide, or no room left in the arena); 8-plane ranges fit in
    // 98 % of the cases (profiles/r06_warp_quarter_split.txt), so a half in which some view would take global taps is swept as its two quarters.
    const int nplanes = d1 - d0;
    auto split_size = [](int m) { return m >= 4 ? ((m / 2 + 1) & ~1) : m; };     // planes of the first part (even); no split below 4 planes
    const int hsz = split_size(nplanes);                         // first half: [0, hsz)
    const int q0 = split_size(hsz), q2 = split_size(nplanes - hsz);   // first quarter of each half: [0, q0), [hsz, hsz + q2)
    if (wave < WL_MAX_SRC) {      // wave k: the boxes of source view k
        const int k = wave;
        // (depth ranges: wave reductions of the per-lane planes over the four quarters; planes need not be monotone.  Lanes beyond the
        //  chunk repeat its last plane and count into the last quarter; an empty quarter reduces to +-inf and is never swept.)
        const float inf = __builtin_inff();
        const bool inq0 = lane < q0, inq1 = lane >= q0 && lane < hsz, inq2 = lane >= hsz && lane < hsz + q2, inq3 = lane >= hsz + q2;
        const float qmin0 = wl_wave_reduce<false>(inq0 ? dlane : inf), qmax0 = wl_wave_reduce<true>(inq0 ? dlane : -inf);
        const float qmin1 = wl_wave_reduce<false>(inq1 ? dlane : inf), qmax1 = wl_wave_reduce<true>(inq1 ? dlane : -inf);
        const float qmin2 = wl_wave_reduce<false>(inq2 ? dlane : inf), qmax2 = wl_wave_reduce<true>(inq2 ? dlane : -inf);
        const float qmin3 = wl_wave_reduce<false>(inq3 ? dlane : inf), qmax3 = wl_wave_reduce<true>(inq3 ? dlane : -inf);
        const float dmin1 = fminf(qmin0, qmin1), dmax1 = fmaxf(qmax0, qmax1);
        const float dmin2 = hsz < nplanes ? fminf(qmin2, qmin3) : dmin1, dmax2 = hsz < nplanes ? fmaxf(qmax2, qmax3) : dmax1;
        const float dmin0 = fminf(dmin1, dmin2), dmax0 = fmaxf(dmax1, dmax2);
        const int set = min(lane >> 3, WL_NSETS - 1);
        const int corner = lane & 7;
        const float cx = (corner & 1) ? (float)min(x0t + WL_T - 1, a.w - 1) : (float)x0t;
        const float cy = (float)(((corner & 2) ? min(y0t + WL_TH - 1, a.h - 1) : y0t) + a.ref_y0);
        const float dlo = set == 0 ? dmin0 : set == 1 ? dmin1 : set == 2 ? dmin2 : set == 3 ? qmin0 : set == 4 ? qmin1 : set == 5 ? qmin2 : qmin3;
        const float dhi = set == 0 ? dmax0 : set == 1 ? dmax1 : set == 2 ? dmax2 : set == 3 ? qmax0 : set == 4 ? qmax1 : set == 5 ? qmax2 : qmax3;
        const float d = (corner & 4) ? dhi : dlo;
        int cX0 = 0, cY0 = 0, cX1 = -1, cY1 = -1, pitch = 4, mode = WL_ZERO;      // mode: WL_FAST / WL_GEN here = "if the arena has room"
        if (k < n_src) {
            typedef const __attribute__((address_space(4))) float* wl_cf;
            wl_cf cam = (wl_cf)(a.cams + ((long)k * a.B + b) * PSCV_CAM_FLOATS);
            const float ax = fmaf(cam[1], cy, cam[0] * cx) + cam[2];
            const float ay = fmaf(cam[4], cy, cam[3] * cx) + cam[5];
            const float az = fmaf(cam[7], cy, cam[6] * cx) + cam[8];
            const float hx = fmaf(ax, d, cam[9]), hy = fmaf(ay, d, cam[10]), hz = fmaf(az, d, cam[11]);
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            const float u = hx * inv_z, v = hy * inv_z;
            const float okf = (hz > 1e-6f && fabsf(u) < 1e6f && fabsf(v) < 1e6f) ? 1.0f : 0.0f;   // also rejects NaN
            // (per 8-lane group = per plane range; every lane of a group holds the group's result)
            const float umin = wl_reduce8<false>(u), umax = wl_reduce8<true>(u);
            const float vmin = wl_reduce8<false>(v), vmax = wl_reduce8<true>(v);
            const bool ok = wl_reduce8<false>(okf) != 0.0f;
            // slack of 1/32 texel: the per-pixel fp32 evaluation (different rounding, 1-ulp rcp on both sides) differs
            // from the corners' by < 2e-6 relative, i.e. < 1/32 for maps up to 16384 texels wide (larger ones are refused)
            const float sl = 1.0f / 32.0f;
            const int X0 = (int)floorf(umin - sl), X1 = (int)floorf(umax + sl) + 1;
            const int Y0 = (int)floorf(vmin - sl), Y1 = (int)floorf(vmax + sl) + 1;
            mode = WL_DIRECT;
            if (ok) {
                const bool outside = X1 < 0 || Y1 < 0 || X0 > a.ws - 1 || Y0 > a.hs - 1;
                const bool inside = X0 >= 0 && Y0 >= 0 && X1 <= a.ws - 1 && Y1 <= a.hs - 1;
                cX0 = max(X0, 0); cX1 = min(X1, a.ws - 1); cY0 = max(Y0, 0); cY1 = min(Y1, a.hs - 1);
                const int bw = cX1 - cX0 + 1, bh = cY1 - cY0 + 1;
                pitch = (bw + 3) & ~3;   // a multiple of 4: the quads of a ds_read_b128 lane group stay conflict-free across rows
                if (outside) { mode = WL_ZERO; cX0 = 0; cY0 = 0; cX1 = -1; cY1 = -1; pitch = 4; }
                else if (bw <= 16 && bh <= WL_BOX_H) mode = inside ? WL_FAST : WL_GEN;
                else { cX0 = 0; cY0 = 0; cX1 = -1; cY1 = -1; pitch = 4; }      // (not staged: keep the record inside 16 bits)
            } else { cX0 = 0; cY0 = 0; cX1 = -1; cY1 = -1; pitch = 4; }
        }
        if (corner == 0 && lane < 8 * WL_NSETS)
            *reinterpret_cast<uint4*>(table + (set * WL_MAX_SRC + k) * 4) =
                make_uint4(((unsigned)cX0 & 0xffffu) | ((unsigned)cY0 << 16), ((unsigned)cX1 & 0xffffu) | ((unsigned)cY1 << 16),
                           (unsigned)pitch | ((unsigned)mode << 16), 0u);
    }
    __syncthreads();

    // ---- 2. every wave: the records of a plane range -> scalar registers; arena allocation greedy in view order (a view whose box
    //         does not fit next to the earlier ones takes global taps), the same in every wave; this lane's view -> vector registers ----
    int bPitch[WL_MAX_SRC], bMode[WL_MAX_SRC], bBase[WL_MAX_SRC], bX0[WL_MAX_SRC], bY0[WL_MAX_SRC], bX1[WL_MAX_SRC], bY1[WL_MAX_SRC];
    bool any_gen = false;
    auto read_records = [&](int set) -> bool {           // returns whether any view of the launch takes global taps
        int used = 0;
        bool direct = false;
        any_gen = false;
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            const uint4 rec = *reinterpret_cast<const uint4*>(table + (set * WL_MAX_SRC + k) * 4);
            const int rx = __builtin_amdgcn_readfirstlane((int)rec.x), ry = __builtin_amdgcn_readfirstlane((int)rec.y);
            const int rz = __builtin_amdgcn_readfirstlane((int)rec.z);
            bX0[k] = (short)(rx & 0xffff); bY0[k] = rx >> 16;
            bX1[k] = (short)(ry & 0xffff); bY1[k] = ry >> 16;
            bPitch[k] = rz & 0xffff;
            int mode = rz >> 16;
            const int need = bPitch[k] * (bY1[k] - bY0[k] + 1);
            if ((mode == WL_FAST || mode == WL_GEN) && used + need > WL_ARENA) mode = WL_DIRECT;
            bBase[k] = used;
            if (mode == WL_FAST || mode == WL_GEN) used += need;
            bMode[k] = mode;
            any_gen = any_gen || mode == WL_GEN;
            direct = direct || (k < n_src && mode == WL_DIRECT);
        }
        return direct;
    };
    // Sweep plan, the same in every wave: the whole chunk if every view's box fits; else each half on its own -- as two quarters if
    // some view would still take global taps in that half.  A segment = (record set, first plane, end plane) packed into 16 bits.
    // (pscv_set_tuning("warp_tile", 1) switches the split off, 3 stops at the halves: A/B runs, scripts/dev/warp_ab.py)
    unsigned long plan = 0;
    int nseg = 0;
    auto push = [&](int set, int pa, int pb) { plan |= (unsigned long)((unsigned)set | (unsigned)pa << 3 | (unsigned)pb << 9) << (16 * nseg); ++nseg; };
    if (!read_records(0) || !(hsz < nplanes) || a.variant == 1) {
        push(0, 0, nplanes);
    } else {
        const bool quarters = a.variant != 3;
        if (read_records(1) && q0 < hsz && quarters) { push(3, 0, q0); push(4, q0, hsz); } else push(1, 0, hsz);
        if (read_records(2) && q2 < nplanes - hsz && quarters) { push(5, hsz, hsz + q2); push(6, hsz + q2, nplanes); } else push(2, hsz, nplanes);
    }
    for (int sub = 0; sub < nseg; ++sub) {
    const unsigned seg = (unsigned)(plan >> (16 * sub)) & 0xffffu;
    if (sub) {
        __syncthreads();                                 // the previous segment's sweep is done with the arena
        __builtin_amdgcn_s_setprio(3);                   // (staging phase at raised priority, like the first one)
    }
    read_records((int)(seg & 7u));
    const int s0 = d0 + (int)((seg >> 3) & 63u), s1 = d0 + (int)(seg >> 9);
    if (a.mode_hist && tid == 0) {                       // (bench.py's mode histogram, per swept plane range; off in product launches)
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k)
            if (k < n_src) atomicAdd(a.mode_hist + k * 4 + bMode[k], 1);
    }
    const int sel = l;
    int mX0 = bX0[0], mY0 = bY0[0], mX1 = bX1[0], mY1 = bY1[0], mpitch = bPitch[0], mbase = bBase[0];
#pragma unroll
    for (int k = 1; k < WL_MAX_SRC; ++k)
        if (sel == k) { mX0 = bX0[k]; mY0 = bY0[k]; mX1 = bX1[k]; mY1 = bY1[k]; mpitch = bPitch[k]; mbase = bBase[k]; }
    const int meb = mbase - mY0 * mpitch - mX0;    // texel index = y * pitch + x + meb

    // ---- 3. stage the boxes, 16-bit -> fp32 on the way: waves 2k, 2k+1 take the even / odd rows of view k's box ----
    {
        constexpr int WPV = WL_THREADS / 64 / WL_MAX_SRC;   // waves per view: 2 (even / odd rows) or 1
        const int k = wave / WPV;
        int X0 = bX0[0], Y0 = bY0[0], X1 = bX1[0], Y1 = bY1[0], vbase = bBase[0], pitch = bPitch[0], mode = bMode[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) { X0 = bX0[t]; Y0 = bY0[t]; X1 = bX1[t]; Y1 = bY1[t]; vbase = bBase[t]; pitch = bPitch[t]; mode = bMode[t]; }
        const void* srcp = a.src[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) srcp = a.src[t];
        if (mode == WL_FAST || mode == WL_GEN) {
            const int bw = X1 - X0 + 1, bh = Y1 - Y0 + 1;      // bw <= 16, bh <= WL_BOX_H
            // eight row loads per wave, issued as one branch-free batch (clamped addresses; only the LDS writes are
            // predicated): a per-row `if` in front of each load costs a full memory round trip per row
            constexpr int RU = WL_STAGE_ROWS;
            const int cl = min(lane, bw * 4 - 1);          // 16-byte chunk of a row
            const bool mine = lane < bw * 4;
            const TIn* col = reinterpret_cast<const TIn*>(srcp) + (((long)b * a.hs + Y0) * a.ws + X0) * C + cl * 8;
            const int dst0 = (vbase + (cl >> 2)) * 64 + (cl & 3) * 16;
            const long rstride = (long)a.ws * C;
            uint4 val[RU];
#pragma unroll
            for (int i = 0; i < RU; ++i) {
                const int ty = min((wave % WPV) + WPV * i, bh - 1);
                val[i] = *reinterpret_cast<const uint4*>(col + ty * rstride);
            }
#pragma unroll
            for (int i = 0; i < RU; ++i) {
                const int ty = (wave % WPV) + WPV * i;
                if (mine && ty < bh) {
                    const uint4 u = val[i];
                    const float4 lo = make_float4(Half16<TIn>::lo(u.x), Half16<TIn>::hi(u.x), Half16<TIn>::lo(u.y), Half16<TIn>::hi(u.y));
                    const float4 hi = make_float4(Half16<TIn>::lo(u.z), Half16<TIn>::hi(u.z), Half16<TIn>::lo(u.w), Half16<TIn>::hi(u.w));
                    *reinterpret_cast<float4*>(lsm + dst0 + ty * pitch * 64) = lo;
                    *reinterpret_cast<float4*>(lsm + dst0 + ty * pitch * 64 + WL_HI) = hi;
                }
            }
        }
    }

    const float invN = 1.0f / (float)(n_src + 1);
    const float invN2 = 1.0f / ((float)(n_src + 1) * (float)(n_src + 1));
    char* const out = reinterpret_cast<char*>(a.out);
    const unsigned long plane_bytes = (unsigned long)hw * C * OB;
    const unsigned lane_out = (unsigned)pflat * (C * OB) + (unsigned)l * (8 * OB);
    const unsigned long img_bytes = (unsigned long)b * a.hs * a.ws * PIXB;
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);

    // ---- 5. sweep: one voxel per quad and step, all source views, per-view mode branches ----
    for (int d = s0 + wave / WL_PG; d < s1; d += 2) {
        const float dval = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), d - d0));
        float s[8], q[8];          // variance: sum, sum of squares; softmin: sum e*diff (s only)
        float sum_e = 0.0f;
        // The sums start at the reference feature (model.py:121-123).  Variance: view 0 WRITES them in three-address form
        // (s = rf + w, q = fma(w, w, rf^2): the same bits as s = rf; s += w) and only a block whose view 0 lies outside the image copies
        // rf / rf^2 -- initialising them up here cost 16 v_mov per trip, 5 % of the loop's vector instructions (round 6).
        if (!VAR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        }
        auto accumulate = [&](const float (&wv)[8], bool first) {
            if (VAR) {
                if (first) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[j] = rf[j] + wv[j]; q[j] = fmaf(wv[j], wv[j], rf2[j]); }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[j] += wv[j]; q[j] = fmaf(wv[j], wv[j], q[j]); }
                }
            } else {   // SOFTMIN  model.py:141-173
                float df[8], part = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = rf[j] - wv[j]; df[j] = t * t; part += df[j]; }
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
                const float e = __expf(-a.temp * part);
                sum_e += e;
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = fmaf(e, df[j], s[j]);
            }
        };

        // this lane's view: sample position -> bilinear weights, byte offset E of the top-left tap in the arena and the byte
        // steps DX / DY to the right / lower taps (only meaningful, and only used, when that view's box is staged)
        float w00, w01, w10, w11;
        int E, DX, DY;            // DX / DY: only blocks with a clipped (GEN) view compute and read them
        {
            const float hx = fmaf(rx, dval, tx), hy = fmaf(ry, dval, ty_), hz = fmaf(rz, dval, tz);
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            float ix = hx * inv_z, iy = hy * inv_z;
            if (any_gen) {   // (a staged box has every corner in front of the camera: no behind-camera test)
                ix = __builtin_amdgcn_fmed3f(ix, a.xlo, a.xhi);      // grid clamp  module.py:151-155
                iy = __builtin_amdgcn_fmed3f(iy, a.ylo, a.yhi);
                // (any_gen is wave-uniform: keep this a BRANCH -- as a select the two clamps, their two constant moves and two v_cndmask
                //  ran in every trip of every block, 2 % of the loop's vector instructions for the 15 % of the blocks that need them)
                asm volatile("" : "+v"(ix), "+v"(iy));
            }
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float fx = ix - x0f, fy = iy - y0f;
            const float gx = 1.0f - fx, gy = 1.0f - fy;
            w00 = gx * gy; w01 = fx * gy; w10 = gx * fy; w11 = fx * fy;
            const int x0 = (int)x0f, y0 = (int)y0f;
            if (any_gen) {
                // zero padding: a tap outside the image has weight 0 and is read from the nearest staged texel instead
                const int x1 = x0 + 1, y1 = y0 + 1;
                const bool vx0 = (unsigned)x0 < (unsigned)a.ws, vx1 = (unsigned)x1 < (unsigned)a.ws;
                const bool vy0 = (unsigned)y0 < (unsigned)a.hs, vy1 = (unsigned)y1 < (unsigned)a.hs;
                w00 = (vx0 && vy0) ? w00 : 0.0f; w01 = (vx1 && vy0) ? w01 : 0.0f;
                w10 = (vx0 && vy1) ? w10 : 0.0f; w11 = (vx1 && vy1) ? w11 : 0.0f;
                const int xc0 = med3_i32(x0, mX0, mX1), xc1 = med3_i32(x1, mX0, mX1);
                const int yc0 = med3_i32(y0, mY0, mY1), yc1 = med3_i32(y1, mY0, mY1);
                E = (__mul24(yc0, mpitch) + xc0 + meb) << 6;
                DX = (xc1 - xc0) << 6;
                DY = __mul24(yc1 - yc0, mpitch) << 6;
            } else {
                E = (__mul24(y0, mpitch) + x0 + meb) << 6;
                asm volatile("" : "=v"(DX), "=v"(DY));               // (defined, no instruction: never read without a GEN view)
            }
        }

#define WL_VIEW(K, CTRL)                                                                                                  \
        if (K < n_src && (bMode[K] != WL_ZERO || !VAR)) {                                                                  \
            float wv[8];                                                                                                   \
            if (bMode[K] == WL_ZERO) {                                                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) wv[j] = 0.0f;                                                \
            } else {                                                                                                       \
                wl_f4 t[8];                                                                                                \
                float w[4];                                                                                                \
                if (bMode[K] != WL_DIRECT) {                                                                               \
                    w[0] = wl_dpp_f<CTRL>(w00); w[1] = wl_dpp_f<CTRL>(w01); w[2] = wl_dpp_f<CTRL>(w10); w[3] = wl_dpp_f<CTRL>(w11); \
                    const unsigned a00 = (unsigned)wl_dpp_i<CTRL>(E) + chb;                                               \
                    if (bMode[K] == WL_FAST) {                                                                             \
                        const unsigned a10 = a00 + ((unsigned)bPitch[K] << 6);                                            \
                        t[0] = wl_tap(lsm, a00);                                               \
                        t[1] = wl_tap(lsm, a00 + WL_HI);                                       \
                        t[2] = wl_tap(lsm, a00 + 64);                                          \
                        t[3] = wl_tap(lsm, a00 + 64 + WL_HI);                                  \
                        t[4] = wl_tap(lsm, a10);                                               \
                        t[5] = wl_tap(lsm, a10 + WL_HI);                                       \
                        t[6] = wl_tap(lsm, a10 + 64);                                          \
                        t[7] = wl_tap(lsm, a10 + 64 + WL_HI);                                  \
                    } else {                                                                                               \
                        const unsigned a01 = a00 + (unsigned)wl_dpp_i<CTRL>(DX);                                          \
                        const unsigned a10 = a00 + (unsigned)wl_dpp_i<CTRL>(DY);                                          \
                        const unsigned a11 = a10 + (a01 - a00);                                                            \
                        t[0] = wl_tap(lsm, a00);                                               \
                        t[1] = wl_tap(lsm, a00 + WL_HI);                                       \
                        t[2] = wl_tap(lsm, a01);                                               \
                        t[3] = wl_tap(lsm, a01 + WL_HI);                                       \
                        t[4] = wl_tap(lsm, a10);                                               \
                        t[5] = wl_tap(lsm, a10 + WL_HI);                                       \
                        t[6] = wl_tap(lsm, a11);                                               \
                        t[7] = wl_tap(lsm, a11 + WL_HI);                                       \
                    }                                                                                                      \
                } else {                                                                                                   \
                    /* general path, direct global taps: behind-camera test, grid clamp, zero padding  module.py:146-166 */ \
                    const float* cam = a.cams + ((long)K * a.B + b) * PSCV_CAM_FLOATS;                                    \
                    const float gax = fmaf(cam[1], py, cam[0] * px) + cam[2];                                             \
                    const float gay = fmaf(cam[4], py, cam[3] * px) + cam[5];                                             \
                    const float gaz = fmaf(cam[7], py, cam[6] * px) + cam[8];                                             \
                    const float hx = fmaf(gax, dval, cam[9]), hy = fmaf(gay, dval, cam[10]), hz = fmaf(gaz, dval, cam[11]); \
                    const bool front = hz > 0.0f;                                                                          \
                    const float inv_z = __builtin_amdgcn_rcpf(hz);                                                        \
                    const float u = front ? hx * inv_z : -10.0f;                                                          \
                    const float v_ = front ? hy * inv_z : -10.0f;                                                         \
                    const float ix = __builtin_amdgcn_fmed3f(u, a.xlo, a.xhi);                                            \
                    const float iy = __builtin_amdgcn_fmed3f(v_, a.ylo, a.yhi);                                           \
                    const float x0f = floorf(ix), y0f = floorf(iy);                                                       \
                    Taps tp;                                                                                               \
                    make_taps<false, PIXB>(ix - x0f, iy - y0f, (int)x0f, (int)y0f, a.hs, a.ws, chb, tp);                 \
                    w[0] = tp.w00; w[1] = tp.w01; w[2] = tp.w10; w[3] = tp.w11;                                           \
                    const char* img = reinterpret_cast<const char*>(a.src[K]) + img_bytes;                                \
                    const uint4 g[4] = {*reinterpret_cast<const uint4*>(img + tp.o00), *reinterpret_cast<const uint4*>(img + tp.o01), \
                                        *reinterpret_cast<const uint4*>(img + tp.o10), *reinterpret_cast<const uint4*>(img + tp.o11)}; \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                        \
                        t[2 * k] = wl_f4{Half16<TIn>::lo(g[k].x), Half16<TIn>::hi(g[k].x), Half16<TIn>::lo(g[k].y), Half16<TIn>::hi(g[k].y)}; \
                        t[2 * k + 1] = wl_f4{Half16<TIn>::lo(g[k].z), Half16<TIn>::hi(g[k].z), Half16<TIn>::lo(g[k].w), Half16<TIn>::hi(g[k].w)}; \
                    }                                                                                                      \
                }                                                                                                          \
                wl_blend8(t, w, wv);                                                                                       \
            }                                                                                                              \
            accumulate(wv, K == 0);                                                                                        \
        } else if (K == 0 && VAR) {      /* view 0 contributes nothing to this block: the sums start as plain copies */    \
            /* (the empty asm pins these copies into THIS block: as plain assignments the compiler hoists them above the branch */ \
            /*  and every trip pays them again) */                                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
                s[j] = rf[j]; q[j] = rf2[j];                                                                               \
                asm volatile("" : "+v"(s[j]), "+v"(q[j]));                                                                 \
            }                                                                                                              \
        }
        WL_VIEW(0, 0x00)
        WL_VIEW(1, 0x55)
        WL_VIEW(2, 0xAA)
        WL_VIEW(3, 0xFF)
#undef WL_VIEW

        float o[8];
        // the packed build's rounding, spelled out (its compiler emits mul, mul, fma / mul, mul, fma for these two expressions)
        if (COST == PSCV_COST_VARIANCE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf(invN, q[j], -__fmul_rn(invN2, __fmul_rn(s[j], s[j])));
        } else if (COST == PSCV_COST_VARIANCE_CVP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float m = __fmul_rn(invN, s[j]); o[j] = fmaf(invN, q[j], -__fmul_rn(m, m)); }
        } else {
            const float inv = 1.0f / (sum_e + 1e-6f);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s[j] * inv;
        }
        // The stored value is the fp32 result ROUNDED to fp32, then to 16 bits -- what "compute in fp32, store 16-bit" means and what
        // the direct kernels do.  Without this the compiler folds the last fma and the conversion into v_fma_mixlo/hi_f16 (ONE rounding
        // of the exact fma): 0.01-0.4 % of the stored values then differ by one fp16 ulp from the direct kernels' (round 4: found when
        // the scalar build became the default and the bit-equality tests against the direct kernels failed).
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(o[j]));
        if (active) wl_store8<TOut>(out + ((unsigned long)b * a.D + d) * plane_bytes + lane_out, o);
    }
    }   // plane sub-range
}

template <typename TIn, typename TOut, int COST>
static int wl_launch(const WarpArgs& a, int nblk, hipStream_t st) {
    auto kern = warp_cost_lds_kernel<TIn, TOut, PSCV_GEOM_PROJ, COST>;
    // "warp_lds_pad" (KiB, measurement knob): ask for more LDS than the kernel needs = fewer workgroups per CU with the same code --
    // the occupancy experiment of scripts/dev/wl_residency.py and the stream-mode co-residency runs (room for another stream's conv0)
    extern Knob g_warp_lds_pad;
    const int lds = min(WL_LDS + 1024 * max(0, (int)g_warp_lds_pad), 160 * 1024);
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
        if (e != hipSuccess) { set_error("pscv_warp_cost(lds): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    const int tiles = a.B * ((a.h + WL_TH - 1) / WL_TH) * ((a.w + WL_T - 1) / WL_T);
    hipLaunchKernelGGL(kern, dim3(8 * ((tiles + 7) / 8), a.n_dchunks), dim3(WL_THREADS), lds, st, a);
    return 0;
}

template <typename TIn, typename TOut>
static int wl_dispatch(const WarpArgs& a, int cost, int nblk, hipStream_t st) {
    if (cost == PSCV_COST_VARIANCE) return wl_launch<TIn, TOut, PSCV_COST_VARIANCE>(a, nblk, st);
    if (cost == PSCV_COST_VARIANCE_CVP) return wl_launch<TIn, TOut, PSCV_COST_VARIANCE_CVP>(a, nblk, st);
    if (cost == PSCV_COST_SOFTMIN) return wl_launch<TIn, TOut, PSCV_COST_SOFTMIN>(a, nblk, st);
    return 1;
}

// Returns 0 if launched, 1 if this configuration is not covered by the LDS-staged kernel (the caller uses the quad /
// generic direct kernels), negative on error.
extern Knob g_warp_tile;   // warp_cost.hip
int* g_wl_mode_hist = nullptr;   // set by pscv_debug_wl_mode_hist (development aid, not thread-safe; shared with warp_cost_lv.hip)

int warp_cost_tiled_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || a.depth_per_pixel || geom != PSCV_GEOM_PROJ || (in_dtype != PSCV_F16 && in_dtype != PSCV_BF16)) return 1;
    if (out_dtype != in_dtype && out_dtype != PSCV_F32) return 1;
    if (a.n_src < 1 || a.n_src > WL_MAX_SRC) return 1;
    if (a.ws > 16384 || a.hs > 16384) return 1;
    if ((long)((a.h + WL_TH - 1) / WL_TH) * ((a.w + WL_T - 1) / WL_T) * a.B >= (1L << 22)) return 1;   // tile index decode is exact below 2^22
    const long tiles = (long)a.B * ((a.h + WL_TH - 1) / WL_TH) * ((a.w + WL_T - 1) / WL_T);
    // planes per block: amortises the box + staging phases (13 % of a 32-plane workgroup's lifetime).  Round 6: 48 planes where the
    // depth axis has room for two such chunks -- the adaptive split (halves, quarters) takes care of the blocks whose 48-plane boxes do
    // not fit: 3-view step 0.900 -> 0.884 / 0.912 -> 0.899 ms on the probe rig, 0.987 -> 0.977 / 0.970 -> 0.970 on the DTU-like rig
    // (free-running per-view graphs, alternating; 64 planes: no better, 24: worse)
    int ppd = ppd_override > 0 ? min((ppd_override + 1) & ~1, 64) : (a.D >= 96 ? 48 : 32);
    while (ppd > 4 && tiles * ((a.D + ppd - 1) / ppd) < 1024) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    a.variant = g_warp_tile;
    a.mode_hist = g_wl_mode_hist;
    const long nblk = tiles * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(lds): bad grid %ld", nblk); return -1; }
    if (in_dtype == PSCV_F16) return out_dtype == PSCV_F32 ? wl_dispatch<f16_t, float>(a, cost, (int)nblk, st)
                                                           : wl_dispatch<f16_t, f16_t>(a, cost, (int)nblk, st);
    return out_dtype == PSCV_F32 ? wl_dispatch<bf16_t, float>(a, cost, (int)nblk, st)
                                 : wl_dispatch<bf16_t, bf16_t>(a, cost, (int)nblk, st);
}

}  // namespace pscv

// Development aid (bench.py's `alt_geometry` / mode histogram): while `hist` (16 device ints, [view 0..3][DIRECT, GEN, FAST, ZERO]) is
// set, every launch of the LDS-staged kernel from this process adds its per-(workgroup, view) staging modes to it; null turns it off.
extern "C" void pscv_debug_wl_mode_hist(int* hist) { pscv::g_wl_mode_hist = hist; }
// occupancy the runtime computes for the f16 variance instantiation (development aid; scripts/dev/wl_occupancy.py)
extern "C" int pscv_debug_wl_occupancy(int* blocks_per_cu, int* lds_bytes, int* threads) {
    auto kern = pscv::warp_cost_lds_kernel<pscv::f16_t, pscv::f16_t, PSCV_GEOM_PROJ, PSCV_COST_VARIANCE>;
    (void)pscv::ensure_dyn_lds(reinterpret_cast<const void*>(kern), pscv::WL_LDS);
    *lds_bytes = pscv::WL_LDS;
    *threads = pscv::WL_THREADS;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kern, pscv::WL_THREADS, pscv::WL_LDS);
}

