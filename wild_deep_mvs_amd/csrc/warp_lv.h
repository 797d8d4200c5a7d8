// Pieces shared by the lane-owns-voxel LDS-staged kernels (warp_cost_lv.hip: variance costs; warp_gc_lv.hip: group-wise correlation):
// geometry of a workgroup, LDS layout, tap reads, the lane -> voxel map of a wave trip.
#pragma once
#include "warp_common.h"
#include "warp_lds.h"

namespace pscv {

constexpr int LV_T = 8, LV_TH = 4;           // tile of reference pixels
constexpr int LV_THREADS = 256;              // 4 waves; a wave trip = 32 pixels x 2 planes
#ifndef LV_OCC
#define LV_OCC 3                             // blocks per CU (= waves per SIMD): 3 -> 52 KiB arena, 168 registers; 4 -> 39.5 KiB, 128
#endif
constexpr int LV_ARENA = LV_OCC == 3 ? 416 : 316;   // staged texels per block (all views), fp32
constexpr int LV_PLANE = LV_ARENA * 16;      // bytes of one channel-chunk plane
constexpr int LV_TABLE = 8 * LV_PLANE;       // per-view box records written by wave 0
constexpr int LV_LDS = LV_TABLE + 3 * WL_MAX_SRC * 32 + 32;   // box records: [3 plane ranges (whole chunk, first half, second half)][4 views] x 32 B
constexpr int LV_BOX_W = 32, LV_BOX_H = 16;  // largest box the staging phase covers (one wave per view, batches of 8 rows x 16 texels)
static_assert(LV_OCC * LV_LDS <= 160 * 1024, "LV_OCC blocks per CU");
static_assert(7 * LV_PLANE + 16 < 65536, "chunk planes within the immediate offset of ds_read");

typedef const __attribute__((address_space(4))) float* lv_cf;   // camera blocks through the scalar cache
typedef const __attribute__((address_space(3))) wl_f4* lv_lp;   // a tap in LDS, by absolute byte address

// (address + constant in one expression: the constant lands in the instruction's offset field)
__device__ __forceinline__ wl_f4 lv_tap(unsigned addr, int off) { return *reinterpret_cast<lv_lp>(addr + (unsigned)off); }

__device__ __forceinline__ void lv_blend(const wl_f4& t00, const wl_f4& t01, const wl_f4& t10, const wl_f4& t11, const float (&w)[4], float (&wv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = fmaf(t11[i], w[3], fmaf(t10[i], w[2], fmaf(t01[i], w[1], t00[i] * w[0])));
}

// The voxel of lane L in a wave trip: LDS pass g (= half wave x index parity of the quad of lanes) is pixel row g of the tile; its 16
// lanes are 8 pixels x 2 planes.
__device__ __forceinline__ void lv_voxel_of(int L, int& prow, int& pcol, int& pp) {
    const int q3 = (L >> 2) & 7;
    const int j16 = ((q3 >> 1) << 2) | (L & 3);
    prow = ((L >> 5) << 1) | (__builtin_popcount(q3) & 1); pcol = j16 & 7; pp = j16 >> 3;
}

// 4 x 4 transpose across the four 16-lane rows of a wave (gfx950 row swaps): in: x[p] = piece p of the voxel each lane owns;
// out: x[v] in lane (row r, column c) = piece r of the voxel of lane (row v, column c).
//   v_permlane32_swap a, b : a = [a0 a1 b0 b1], b = [a2 a3 b2 b3]      v_permlane16_swap a, b : a = [a0 b0 a2 b2], b = [a1 b1 a3 b3]
__device__ __forceinline__ void lv_row_transpose(uint32_t (&x)[4]) {
    auto s02 = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false);
    auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    x[0] = t01[0]; x[1] = t01[1]; x[2] = t23[0]; x[3] = t23[1];
}

// Stage the box of one source view, 16-bit -> fp32, channel-chunk planar, with zero padding outside the image (one wave; a batch =
// 8 rows x 16 texels, a lane = one 16-byte piece of a row: texel lane >> 2, channels 8 (lane & 3) ..; the loads of a batch first).
// (sX0..sY1): the box (may reach two texels beyond the image); sP16: bytes per box row in a chunk plane; sBase: byte address of
// texel (0, 0) of the image in chunk plane 0 (relative to lsm).
template <typename TIn>
__device__ __forceinline__ void lv_stage_box(unsigned char* lsm, const void* srcp, int b, int hs, int ws, int lane, int sX0, int sY0, int sX1,
                                             int sY1, int sP16, int sBase) {
    constexpr int C = 32;
    const int bw = sX1 - sX0 + 1, bh = sY1 - sY0 + 1;              // bw <= 32, bh <= 16
    const long rstride = (long)ws * C;
    for (int yh = 0; yh < bh; yh += 8) {
        for (int xh = 0; xh < bw; xh += 16) {
            const int cw = min(bw - xh, 16), ch = min(bh - yh, 8);
            const int cl = min(lane, cw * 4 - 1);
            const bool mine = lane < cw * 4;
            const int gx = sX0 + xh + (cl >> 2);
            const bool vx = (unsigned)gx < (unsigned)ws;
            const TIn* col = reinterpret_cast<const TIn*>(srcp) + ((long)b * hs * ws + min(max(gx, 0), ws - 1)) * C + (cl & 3) * 8;
            const int dst0 = gx * 16 + sBase + (cl & 3) * 2 * LV_PLANE;
            uint4 val[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gy = min(max(sY0 + yh + min(i, ch - 1), 0), hs - 1);
                val[i] = *reinterpret_cast<const uint4*>(col + gy * rstride);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (mine && i < ch) {
                    const bool v = vx && (unsigned)(sY0 + yh + i) < (unsigned)hs;
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

}  // namespace pscv
