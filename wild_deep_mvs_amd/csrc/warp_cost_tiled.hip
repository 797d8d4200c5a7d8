// LDS-staged plane-sweep warp + cost for 32-channel 16-bit feature maps (gfx950), second design.
//
// What bounds the sweep (profiles/README.md, scripts/ubench/valu_rate2.hip): the direct-gather kernels are limited by
// vector-ALU issue and by the per-CU L1 tap rate, not by HBM.  Measured issue rates on MI355X: plain fp32 `v_fma_f32` /
// `v_add_f32` / `v_and_b32` are full rate; EVERY 16-bit form (`v_fma_mix_f32`, `v_pk_fma_f16`, `v_dot2_f32_f16|bf16`,
// `v_cvt_f32_f16`, `v_lshlrev_b32`, `v_perm_b32`) and every DPP move is half rate; `ds_read_b128` moves 256 B/clk/CU and
// overlaps fully with the vector ALU.  So:
//
//   * the source patches a tile of reference pixels can touch are staged in LDS ONCE per (tile, depth chunk, view) and
//     are CONVERTED TO FP32 while they are staged (conversion cost is paid per staged texel, ~14x fewer than taps);
//   * every bilinear tap is then two `ds_read_b128` (8 channels per lane) and the blend is four full-rate fp32 FMAs
//     per channel -- the same fp32 operation chain as the direct kernels (`fmaf(float(h), w, acc)` == `v_fma_mix_f32`),
//     so the results are bit-identical to theirs;
//   * a quad of lanes owns a voxel (lane l: channels 8l..8l+7) and each lane of the quad computes the sample position
//     of a DIFFERENT (plane, view) combination -- two planes x two source views per step -- so the coordinate
//     arithmetic runs once per four voxel-views; weights and the texel index travel through the quad with DPP;
//   * a view whose texel box lies strictly inside the source image needs no validity masks, clamps or behind-camera
//     test at all (block-uniform decision from the 8 corner projections of the tile at the chunk's depth extremes);
//     other views use direct global taps with the general (zero-padding) arithmetic for that block only.
//
// Occupancy: 512 threads = 8 x 8 reference pixels x PD planes (8), all source views resident: 592 texels x 128 B fp32
// + ray terms = 79 KiB -> two blocks (16 waves) per CU.  LDS layout of a texel: "lo" plane holds channels
// {8l..8l+3 : l = 0..3} (64 B), "hi" plane {8l+4..8l+7}; a quad reads 64 contiguous bytes per instruction and the four
// quads of a `ds_read_b128` lane group hit four texels of one row, conflict-free when the box pitch is a multiple of 4.
//
// Semantics and citations are those of warp_cost.hip.
#include <type_traits>

#include "warp_common.h"

namespace pscv {

constexpr int WL_THREADS = 512;
constexpr int WL_T = 8;                      // tile = WL_T x WL_T reference pixels
constexpr int WL_ARENA = 640;                // staged texels per block (all views): 80 KiB of fp32 -> two blocks per CU
constexpr int WL_HI = WL_ARENA * 64;         // byte offset of the "hi" channel plane
constexpr int WL_MAX_SRC = 4;                // source views of this kernel = lanes of a quad (others: quad kernel)
constexpr int WL_LDS = 2 * WL_HI;
static_assert(WL_LDS <= 81920, "two blocks per CU");

typedef float wl_f2 __attribute__((ext_vector_type(2)));

// quad broadcast: every lane of a quad reads quad lane CTRL & 3.  (bound_ctrl with full row / bank masks: no lane keeps its
// old value, so the compiler needs no copy of the source in front of the move.)
template <int CTRL> __device__ __forceinline__ int wl_dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float wl_dpp_f(float x) {
    return __builtin_bit_cast(float, wl_dpp_i<CTRL>(__builtin_bit_cast(int, x)));
}

// eight fp32 channels x four taps -> eight blended channels; t = {00lo, 00hi, 01lo, 01hi, 10lo, 10hi, 11lo, 11hi}
__device__ __forceinline__ void wl_blend8(const float4 (&t)[8], const float (&w)[4], float (&o)[8]) {
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

template <typename TOut> __device__ __forceinline__ void wl_store8(char* p, const float (&o)[8]) {
    if constexpr (sizeof(TOut) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p + 16) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(Half16<TOut>::pack(o[0], o[1]), Half16<TOut>::pack(o[2], o[3]),
                                                   Half16<TOut>::pack(o[4], o[5]), Half16<TOut>::pack(o[6], o[7]));
    }
}

// per-(block, view) staging mode, wave-uniform
constexpr int WL_DIRECT = 0;   // not staged (a corner at / behind the source camera, or the box does not fit): global taps
constexpr int WL_GEN = 1;      // box clipped at the image border: LDS taps, general (zero-padding) weights
constexpr int WL_FAST = 2;     // box strictly inside the image: LDS taps, no masks / clamps
constexpr int WL_ZERO = 3;     // box entirely outside the image: every tap is zero padding, the view contributes f = 0

template <typename TIn, typename TOut, int GEOM, int COST>
__global__ __launch_bounds__(WL_THREADS, 4) void warp_cost_lds_kernel(const WarpArgs a) {
    constexpr int C = 32, PIXB = 64;
    constexpr int OB = (int)sizeof(TOut);
    constexpr bool VAR = COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP;
    static_assert(GEOM == PSCV_GEOM_PROJ, "PROJ geometry (three depth-independent ray terms per (view, pixel))");
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];

    // ---- work decode: XCD-banded, tile-major, depth-chunk minor (the chunks of a tile re-read nearly the same texels: L2) ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
    wg = __builtin_amdgcn_readfirstlane(wg);
    const int dc = wg % a.n_dchunks; wg /= a.n_dchunks;
    const int ntx = (a.w + WL_T - 1) / WL_T, nty = (a.h + WL_T - 1) / WL_T;
    const int txi = wg % ntx; wg /= ntx;
    const int tyi = wg % nty;
    const int b = wg / nty;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int x0t = txi * WL_T, y0t = tyi * WL_T;
    const int d0 = dc * a.ppd, d1 = min(a.D, d0 + a.ppd);
    const float* const depth_b = a.depth + (long)b * a.depth_bstride;
    const int n_src = a.n_src;

    // ---- 1. depth planes of the chunk: lane i holds plane d0 + i (<= 64 planes per chunk); extremes by a wave reduction
    //         (planes need not be monotone) ----
    const float dlane = depth_b[min(d0 + lane, d1 - 1)];
    float dmin = dlane, dmax = dlane;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        dmin = fminf(dmin, __shfl_xor(dmin, m, 64)); dmax = fmaxf(dmax, __shfl_xor(dmax, m, 64));
    }

    // ---- 2. texel box per view from the 8 corner projections (tile corners x depth extremes): for a fixed plane the warp
    //         is a homography (convex sets stay convex while z > 0), for a fixed pixel the sample moves monotonically along
    //         its epipolar line, so every sample of the (tile, chunk) lies in the bounding box of these 8 points.  Every
    //         wave computes this for itself (lanes 0..31 = 4 views x 8 corners) into scalar registers: no LDS, no barrier. ----
    int bX0[WL_MAX_SRC], bY0[WL_MAX_SRC], bX1[WL_MAX_SRC], bY1[WL_MAX_SRC], bBase[WL_MAX_SRC], bPitch[WL_MAX_SRC], bMode[WL_MAX_SRC];
    bool any_gen = false;
    {
        const int view = min((lane >> 3) & 3, n_src - 1), corner = lane & 7;
        const float cxl = (float)x0t, cxh = (float)min(x0t + WL_T - 1, a.w - 1);
        const float cyl = (float)y0t, cyh = (float)min(y0t + WL_T - 1, a.h - 1);
        const float* cam = a.cams + ((long)view * a.B + b) * PSCV_CAM_FLOATS;
        const float px = (corner & 1) ? cxh : cxl, py = (corner & 2) ? cyh : cyl, d = (corner & 4) ? dmax : dmin;
        const float ax = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ay = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float az = fmaf(cam[7], py, cam[6] * px) + cam[8];
        const float hx = fmaf(ax, d, cam[9]), hy = fmaf(ay, d, cam[10]), hz = fmaf(az, d, cam[11]);
        const float inv_z = 1.0f / hz;
        const float u = hx * inv_z, v = hy * inv_z;
        bool ok = hz > 1e-6f && fabsf(u) < 1e6f && fabsf(v) < 1e6f;   // also rejects NaN
        float umin = u, umax = u, vmin = v, vmax = v;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            umin = fminf(umin, __shfl_xor(umin, m, 64)); umax = fmaxf(umax, __shfl_xor(umax, m, 64));
            vmin = fminf(vmin, __shfl_xor(vmin, m, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
            ok = ok && (__shfl_xor((int)ok, m, 64) != 0);
        }
        // slack of 1/64 texel: the per-pixel fp32 evaluation (1-ulp rcp, different rounding) differs from the corners' by
        // < 1e-6 relative, i.e. < 1/64 for maps up to 16384 texels wide (warp_cost_tiled_try refuses larger ones)
        const float sl = 1.0f / 64.0f;
        const int X0 = ok ? (int)floorf(umin - sl) : 0, X1 = ok ? (int)floorf(umax + sl) + 1 : -1;
        const int Y0 = ok ? (int)floorf(vmin - sl) : 0, Y1 = ok ? (int)floorf(vmax + sl) + 1 : -1;
        int used = 0;
#pragma unroll
        for (int k = 0; k < WL_MAX_SRC; ++k) {
            const int rX0 = __builtin_amdgcn_readlane(X0, 8 * k), rX1 = __builtin_amdgcn_readlane(X1, 8 * k);
            const int rY0 = __builtin_amdgcn_readlane(Y0, 8 * k), rY1 = __builtin_amdgcn_readlane(Y1, 8 * k);
            const bool okk = __builtin_amdgcn_readlane((int)ok, 8 * k) != 0;
            const bool outside = rX1 < 0 || rY1 < 0 || rX0 > a.ws - 1 || rY0 > a.hs - 1;
            const bool inside = rX0 >= 0 && rY0 >= 0 && rX1 <= a.ws - 1 && rY1 <= a.hs - 1;
            bX0[k] = max(rX0, 0); bX1[k] = min(rX1, a.ws - 1);
            bY0[k] = max(rY0, 0); bY1[k] = min(rY1, a.hs - 1);
            const int bw = bX1[k] - bX0[k] + 1, bh = bY1[k] - bY0[k] + 1;
            bPitch[k] = (bw + 3) & ~3;   // a multiple of 4: the four quads of a ds_read_b128 lane group stay conflict-free across rows
            int mode = WL_DIRECT;
            if (k < n_src && okk) {
                if (outside) mode = WL_ZERO;
                else if (bw <= 16 && used + bPitch[k] * bh <= WL_ARENA) mode = inside ? WL_FAST : WL_GEN;
            }
            bMode[k] = mode;
            bBase[k] = used;
            if (mode == WL_FAST || mode == WL_GEN) used += bPitch[k] * bh;
            any_gen = any_gen || mode == WL_GEN;
        }
    }

    // ---- 3. stage the boxes, 16-bit -> fp32 on the way: waves 2k, 2k+1 take the even / odd rows of view k's box ----
    {
        const int k = wave >> 1;
        int X0 = bX0[0], Y0 = bY0[0], X1 = bX1[0], Y1 = bY1[0], vbase = bBase[0], pitch = bPitch[0], mode = bMode[0];
        const void* srcp = a.src[0];
#pragma unroll
        for (int t = 1; t < WL_MAX_SRC; ++t)
            if (k == t) { X0 = bX0[t]; Y0 = bY0[t]; X1 = bX1[t]; Y1 = bY1[t]; vbase = bBase[t]; pitch = bPitch[t]; mode = bMode[t]; srcp = a.src[t]; }
        if (mode == WL_FAST || mode == WL_GEN) {
            const int bw = X1 - X0 + 1, bh = Y1 - Y0 + 1;
            constexpr int RU = 8;     // rows in flight per wave (boxes are <= 16 rows in practice; the loop covers any height)
            const bool mine = lane < bw * 4;          // 16-byte chunk of a row (bw <= 16 texels)
            const TIn* col = reinterpret_cast<const TIn*>(srcp) + (((long)b * a.hs + Y0) * a.ws + X0) * C + lane * 8;
            const int dst0 = (vbase + (lane >> 2)) * 64 + (lane & 3) * 16;
            for (int r0 = wave & 1; r0 < bh; r0 += 2 * RU) {
                uint4 val[RU];
#pragma unroll
                for (int i = 0; i < RU; ++i) {
                    const int ty = r0 + 2 * i;
                    if (mine && ty < bh) val[i] = *reinterpret_cast<const uint4*>(col + (long)ty * a.ws * C);
                }
#pragma unroll
                for (int i = 0; i < RU; ++i) {
                    const int ty = r0 + 2 * i;
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
    }

    // ---- 4. per-lane constants of the sweep: lane l of a quad owns channels 8l..8l+7 of the quad's voxel and computes the
    //         sample position in source view l ----
    const int quad = lane >> 2, l = lane & 3;
    const int pl = (2 * (wave & 3) + (quad >> 3)) * 8 + (quad & 7);     // pixel of the tile
    int x = x0t + (pl & 7), y = y0t + (pl >> 3);
    const bool active = x < a.w && y < a.h;
    x = min(x, a.w - 1); y = min(y, a.h - 1);
    const int hw = a.h * a.w;
    const int pflat = y * a.w + x;
    const float px = (float)x, py = (float)y;
    const unsigned chb = (unsigned)l * 16u;

    // view l: depth-independent ray terms rot (x, y, 1), translation, and its (clipped) box
    float rx, ry, rz, tx, ty_, tz;
    {
        const float* cam = a.cams + ((long)min(l, n_src - 1) * a.B + b) * PSCV_CAM_FLOATS;
        rx = fmaf(cam[1], py, cam[0] * px) + cam[2];
        ry = fmaf(cam[4], py, cam[3] * px) + cam[5];
        rz = fmaf(cam[7], py, cam[6] * px) + cam[8];
        tx = cam[9]; ty_ = cam[10]; tz = cam[11];
    }
    int mX0 = bX0[0], mX1 = bX1[0], mY0 = bY0[0], mY1 = bY1[0], mpitch = bPitch[0], mbase = bBase[0];
#pragma unroll
    for (int k = 1; k < WL_MAX_SRC; ++k)
        if (l == k) { mX0 = bX0[k]; mX1 = bX1[k]; mY0 = bY0[k]; mY1 = bY1[k]; mpitch = bPitch[k]; mbase = bBase[k]; }
    const int meb = mbase - mY0 * mpitch - mX0;    // texel index = y * pitch + x + meb

    float rf[8];
    {
        const f32x8 t = Elem<TIn>::load8(reinterpret_cast<const TIn*>(a.ref) + ((long)b * hw + pflat) * C + l * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) rf[j] = t.v[j];
    }
    const float invN = 1.0f / (float)(n_src + 1);
    const float invN2 = 1.0f / ((float)(n_src + 1) * (float)(n_src + 1));
    char* const out = reinterpret_cast<char*>(a.out);
    const unsigned long plane_bytes = (unsigned long)hw * C * OB;
    const unsigned lane_out = (unsigned)pflat * (C * OB) + (unsigned)l * (8 * OB);
    const unsigned long img_bytes = (unsigned long)b * a.hs * a.ws * PIXB;
    __syncthreads();

    // ---- 5. sweep: one voxel per quad and step, all source views ----
    for (int d = d0 + (wave >> 2); d < d1; d += 2) {
        const float dval = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlane), d - d0));
        float s[8], q[8];          // variance: sum, sum of squares; softmin: sum e*diff (s only)
        float sum_e = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (VAR) { s[j] = rf[j]; q[j] = rf[j] * rf[j]; }   // the sums start at the reference feature  model.py:121-123
            else { s[j] = 0.0f; q[j] = 0.0f; }
        }
        auto accumulate = [&](const float (&wv)[8]) {
            if (VAR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += wv[j]; q[j] = fmaf(wv[j], wv[j], q[j]); }
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
        int E, DX = 64, DY = mpitch << 6;
        {
            const float hx = fmaf(rx, dval, tx), hy = fmaf(ry, dval, ty_), hz = fmaf(rz, dval, tz);
            const float inv_z = __builtin_amdgcn_rcpf(hz);
            float ix = hx * inv_z, iy = hy * inv_z;
            if (any_gen) {   // (a staged box has every corner in front of the camera: no behind-camera test)
                ix = __builtin_amdgcn_fmed3f(ix, a.xlo, a.xhi);      // grid clamp  module.py:151-155
                iy = __builtin_amdgcn_fmed3f(iy, a.ylo, a.yhi);
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
                E = (yc0 * mpitch + xc0 + meb) << 6;
                DX = (xc1 - xc0) << 6;
                DY = ((yc1 - yc0) * mpitch) << 6;
            } else {
                E = (y0 * mpitch + x0 + meb) << 6;
            }
        }

#define WL_VIEW(K, CTRL)                                                                                                  \
        if (K < n_src && (bMode[K] != WL_ZERO || !VAR)) {                                                                  \
            float wv[8];                                                                                                   \
            if (bMode[K] == WL_ZERO) {                                                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) wv[j] = 0.0f;                                                \
            } else {                                                                                                       \
                float4 t[8];                                                                                               \
                float w[4];                                                                                                \
                if (bMode[K] != WL_DIRECT) {                                                                               \
                    w[0] = wl_dpp_f<CTRL>(w00); w[1] = wl_dpp_f<CTRL>(w01); w[2] = wl_dpp_f<CTRL>(w10); w[3] = wl_dpp_f<CTRL>(w11); \
                    const unsigned a00 = (unsigned)wl_dpp_i<CTRL>(E) + chb;                                               \
                    if (bMode[K] == WL_FAST) {                                                                             \
                        const unsigned a10 = a00 + ((unsigned)bPitch[K] << 6);                                            \
                        t[0] = *reinterpret_cast<const float4*>(lsm + a00);                                               \
                        t[1] = *reinterpret_cast<const float4*>(lsm + a00 + WL_HI);                                       \
                        t[2] = *reinterpret_cast<const float4*>(lsm + a00 + 64);                                          \
                        t[3] = *reinterpret_cast<const float4*>(lsm + a00 + 64 + WL_HI);                                  \
                        t[4] = *reinterpret_cast<const float4*>(lsm + a10);                                               \
                        t[5] = *reinterpret_cast<const float4*>(lsm + a10 + WL_HI);                                       \
                        t[6] = *reinterpret_cast<const float4*>(lsm + a10 + 64);                                          \
                        t[7] = *reinterpret_cast<const float4*>(lsm + a10 + 64 + WL_HI);                                  \
                    } else {                                                                                               \
                        const unsigned a01 = a00 + (unsigned)wl_dpp_i<CTRL>(DX);                                          \
                        const unsigned a10 = a00 + (unsigned)wl_dpp_i<CTRL>(DY);                                          \
                        const unsigned a11 = a10 + (a01 - a00);                                                            \
                        t[0] = *reinterpret_cast<const float4*>(lsm + a00);                                               \
                        t[1] = *reinterpret_cast<const float4*>(lsm + a00 + WL_HI);                                       \
                        t[2] = *reinterpret_cast<const float4*>(lsm + a01);                                               \
                        t[3] = *reinterpret_cast<const float4*>(lsm + a01 + WL_HI);                                       \
                        t[4] = *reinterpret_cast<const float4*>(lsm + a10);                                               \
                        t[5] = *reinterpret_cast<const float4*>(lsm + a10 + WL_HI);                                       \
                        t[6] = *reinterpret_cast<const float4*>(lsm + a11);                                               \
                        t[7] = *reinterpret_cast<const float4*>(lsm + a11 + WL_HI);                                       \
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
                        t[2 * k] = make_float4(Half16<TIn>::lo(g[k].x), Half16<TIn>::hi(g[k].x), Half16<TIn>::lo(g[k].y), Half16<TIn>::hi(g[k].y)); \
                        t[2 * k + 1] = make_float4(Half16<TIn>::lo(g[k].z), Half16<TIn>::hi(g[k].z), Half16<TIn>::lo(g[k].w), Half16<TIn>::hi(g[k].w)); \
                    }                                                                                                      \
                }                                                                                                          \
                wl_blend8(t, w, wv);                                                                                       \
            }                                                                                                              \
            accumulate(wv);                                                                                                \
        }
        WL_VIEW(0, 0x00)
        WL_VIEW(1, 0x55)
        WL_VIEW(2, 0xAA)
        WL_VIEW(3, 0xFF)
#undef WL_VIEW

        float o[8];
        if (COST == PSCV_COST_VARIANCE) {
            const wl_f2 n1 = wl_f2{invN, invN}, n2 = wl_f2{invN2, invN2};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wl_f2 s2 = wl_f2{s[2 * j], s[2 * j + 1]}, q2 = wl_f2{q[2 * j], q[2 * j + 1]};
                const wl_f2 r = q2 * n1 - (s2 * s2) * n2;
                o[2 * j] = r[0]; o[2 * j + 1] = r[1];
            }
        } else if (COST == PSCV_COST_VARIANCE_CVP) {
            const wl_f2 n1 = wl_f2{invN, invN};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wl_f2 s2 = wl_f2{s[2 * j], s[2 * j + 1]}, q2 = wl_f2{q[2 * j], q[2 * j + 1]};
                const wl_f2 m = s2 * n1;
                const wl_f2 r = q2 * n1 - m * m;
                o[2 * j] = r[0]; o[2 * j + 1] = r[1];
            }
        } else {
            const float inv = 1.0f / (sum_e + 1e-6f);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = s[j] * inv;
        }
        if (active) wl_store8<TOut>(out + ((unsigned long)b * a.D + d) * plane_bytes + lane_out, o);
    }
}

template <typename TIn, typename TOut, int COST>
static int wl_launch(const WarpArgs& a, int nblk, hipStream_t st) {
    auto kern = warp_cost_lds_kernel<TIn, TOut, PSCV_GEOM_PROJ, COST>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS);
        if (e != hipSuccess) { set_error("pscv_warp_cost(lds): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(WL_THREADS), WL_LDS, st, a);
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
int warp_cost_tiled_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || a.depth_per_pixel || geom != PSCV_GEOM_PROJ || (in_dtype != PSCV_F16 && in_dtype != PSCV_BF16)) return 1;
    if (out_dtype != in_dtype && out_dtype != PSCV_F32) return 1;
    if (a.n_src < 1 || a.n_src > WL_MAX_SRC) return 1;
    if (a.ws > 16384 || a.hs > 16384) return 1;
    const long tiles = (long)a.B * ((a.h + WL_T - 1) / WL_T) * ((a.w + WL_T - 1) / WL_T);
    int ppd = ppd_override > 0 ? min((ppd_override + 1) & ~1, 64) : 16;   // planes per block: amortises the patch staging
    while (ppd > 4 && tiles * ((a.D + ppd - 1) / ppd) < 1024) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    const long nblk = tiles * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(lds): bad grid %ld", nblk); return -1; }
    if (in_dtype == PSCV_F16) return out_dtype == PSCV_F32 ? wl_dispatch<f16_t, float>(a, cost, (int)nblk, st)
                                                           : wl_dispatch<f16_t, f16_t>(a, cost, (int)nblk, st);
    return out_dtype == PSCV_F32 ? wl_dispatch<bf16_t, float>(a, cost, (int)nblk, st)
                                 : wl_dispatch<bf16_t, bf16_t>(a, cost, (int)nblk, st);
}

}  // namespace pscv
