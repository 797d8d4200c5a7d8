// Pieces shared by the two LDS-staged plane-sweep kernels (warp_cost_tiled.hip: a quad of lanes owns a voxel; warp_cost_lv.hip: a lane
// owns a voxel): DPP broadcasts / reductions, the saturating 16-bit pack, the per-(block, view) staging modes.
#pragma once
#include "warp_common.h"

namespace pscv {

constexpr int WL_MAX_SRC = 4;                // source views of the LDS-staged kernels (more: quad kernel)
typedef float wl_f4 __attribute__((ext_vector_type(4)));

// quad broadcast: every lane of a quad reads quad lane CTRL & 3.  (bound_ctrl with full row / bank masks: no lane keeps its
// old value, so the compiler needs no copy of the source in front of the move.)
template <int CTRL> __device__ __forceinline__ int wl_dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float wl_dpp_f(float x) {
    return __builtin_bit_cast(float, wl_dpp_i<CTRL>(__builtin_bit_cast(int, x)));
}


// min / max over groups of 8 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror) and over the whole wave (+ row_mirror,
// row_bcast15, row_bcast31; the result is read from lane 63): vector-ALU DPP modifiers instead of LDS-crossbar shuffles
template <bool MAX> __device__ __forceinline__ float wl_mm(float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); }
template <bool MAX, int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float wl_red_step(float x) {
    const int xi = __builtin_bit_cast(int, x);
    const float y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, ROWMASK, 0xf, false));
    return wl_mm<MAX>(x, y);
}
template <bool MAX> __device__ __forceinline__ float wl_reduce8(float x) {
    x = wl_red_step<MAX, 0xB1>(x);     // quad_perm [1,0,3,2]
    x = wl_red_step<MAX, 0x4E>(x);     // quad_perm [2,3,0,1]
    return wl_red_step<MAX, 0x141>(x); // row_half_mirror
}
template <bool MAX> __device__ __forceinline__ float wl_wave_reduce(float x) {
    x = wl_reduce8<MAX>(x);
    x = wl_red_step<MAX, 0x140>(x);          // row_mirror: all 16 lanes of a row
    x = wl_red_step<MAX, 0x142, 0xa>(x);     // row_bcast15 into rows 1 and 3
    x = wl_red_step<MAX, 0x143, 0xc>(x);     // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// fp16 stores saturate at +-65504 like every other kernel of the engine (pscv_common.h), but through the MODE.FP16_OVFL bit the
// kernel sets at its start ("an overflowed FP16 result is clamped to +-MAX_FP16 ... preserving true INF"): the per-element
// v_med3_f32 clamp of pack_f16x2 costs 8 vector-ALU instructions per voxel here, 5 % of the sweep
template <typename TOut> __device__ __forceinline__ uint32_t wl_pack2(float lo, float hi) {
    if constexpr (Half16<TOut>::dtype == PSCV_F16) {
        h2_t v;
        v[0] = (_Float16)lo;
        v[1] = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    } else {
        return Half16<TOut>::pack(lo, hi);
    }
}
// per-(block, view) staging mode, wave-uniform
constexpr int WL_DIRECT = 0;   // not staged (a corner at / behind the source camera, or the box does not fit): global taps
constexpr int WL_GEN = 1;      // box clipped at the image border: LDS taps, general (zero-padding) weights
constexpr int WL_FAST = 2;     // box strictly inside the image: LDS taps, no masks / clamps
constexpr int WL_ZERO = 3;     // box entirely outside the image: every tap is zero padding, the view contributes f = 0

}  // namespace pscv
