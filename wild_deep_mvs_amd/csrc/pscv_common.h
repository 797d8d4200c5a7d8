// Shared device/host helpers for the pscv kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/pscv.h"

namespace pscv {

// ---- error channel ---------------------------------------------------------
void set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel, device, size), from
// any host thread (pscv_host.cpp).  A process that launches on a second GPU, or a larger request later, sets it again.
hipError_t ensure_dyn_lds(const void* kernel, int bytes);
int device_cu_count();      // compute units of the current device (cached per device); <= 0 on error

// Tuning knob (pscv_set_tuning): ONE process-wide value, seen by every host thread that launches -- PyTorch runs autograd's backward
// on its own thread and DataParallel runs replicas on worker threads, so a knob set from the main thread must reach them --
// plus an optional override for the calling thread only (pscv_set_tuning_thread: concurrent A/B runs on different streams).
// Reads like an int at the use sites.
struct Knob {
    int process;                    // written by pscv_set_tuning (relaxed atomic through the builtins below)
    int id;                         // slot of the thread-local override table (pscv_host.cpp)
    operator int() const;
    void set(int v) { __atomic_store_n(&process, v, __ATOMIC_RELAXED); }
};
enum { KNOB_WARP_LPV, KNOB_WARP_PPD, KNOB_WARP_TILED, KNOB_WARP_Q2, KNOB_CONV_SMALL_TILES, KNOB_SWEEP_TH16, KNOB_SWEEP_DC,
       KNOB_SWEEPC_SLOTS, KNOB_SWEEPC_PD, KNOB_C1_NB, KNOB_C1_SWEEP, KNOB_WARP_BWD_DIRECT, KNOB_CONV_S2_SWEEP, KNOB_S2S_SLOTS,
       KNOB_WARP_TILE, KNOB_FUSE_C0, KNOB_SPARE1, KNOB_SPARE2, KNOB_SPARE3, KNOB_SPARE4, KNOB_SWEEP_KDM, KNOB_SWEEP_KDM_PD, KNOB_WARP_LDS_PAD, KNOB_WARP_GC_LDS, KNOB_TAIL_NBK, KNOB_CONV_WIDE, KNOB_CONV_SMALL_NT, KNOB_COUNT };
bool knob_thread_value(int id, int* v);             // this thread's override of slot id, if one is set
void knob_thread_set(int id, int v, bool enable);

#define PSCV_CHECK_ARG(cond, ...)          \
    do {                                   \
        if (!(cond)) {                     \
            pscv::set_error(__VA_ARGS__);  \
            return -1;                     \
        }                                  \
    } while (0)

#define PSCV_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            pscv::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return -2;                                                                 \
        }                                                                              \
    } while (0)

// ---- phase timing (development builds only: -DPSCV_PROFILE) ---------------------------------------------------------
// Cycle stamps are kept in registers and written once per workgroup (wave 0) at the end: slots 0..5 = cycles of phases 0..5,
// 6 / 7 = absolute begin / end (low 32 bits), 8 = HW_ID, 9 = XCC_ID.  Each translation unit owns its buffer and exports a getter
// (PSCV_PROF_EXPORT); scripts/dev/phase_prof.py reads them.
#ifdef PSCV_PROFILE
constexpr int PSCV_PROF_BLOCKS = 16384;
#define PSCV_PROF_BUFFER(tag) __device__ unsigned int pscv_prof_##tag[pscv::PSCV_PROF_BLOCKS * 16];
#define PSCV_PROF_BEGIN unsigned long long pt_prev = __builtin_readcyclecounter(); const unsigned long long pt_begin = pt_prev; \
    unsigned pt_acc[6] = {0, 0, 0, 0, 0, 0};
#define PSCV_STAMP(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pt_acc[i] += (unsigned)(t_ - pt_prev); pt_prev = t_; }
#define PSCV_STAMP_WAIT(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PSCV_STAMP(i) }
#define PSCV_PROF_END(tag, blk) if (threadIdx.x == 0 && (blk) < pscv::PSCV_PROF_BLOCKS) { \
        unsigned hwid_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_)); \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
        unsigned int* o_ = pscv_prof_##tag + (blk) * 16; \
        for (int i_ = 0; i_ < 6; ++i_) o_[i_] = pt_acc[i_]; \
        o_[6] = (unsigned)pt_begin; o_[7] = (unsigned)pt_prev; o_[8] = hwid_; o_[9] = xcc_; }
#define PSCV_PROF_EXPORT(tag) extern "C" int pscv_debug_prof_##tag(unsigned int* out, int n_blocks) { \
        if (n_blocks > pscv::PSCV_PROF_BLOCKS) n_blocks = pscv::PSCV_PROF_BLOCKS; \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pscv::pscv_prof_##tag), (size_t)n_blocks * 16 * sizeof(unsigned int)); }
#else
#define PSCV_PROF_BUFFER(tag)
#define PSCV_PROF_BEGIN
#define PSCV_STAMP(i)
#define PSCV_STAMP_WAIT(i)
#define PSCV_PROF_END(tag, blk)
#define PSCV_PROF_EXPORT(tag)
#endif

// ---- workgroup-id decode ----------------------------------------------------------------------------------------------
// x % d and x /= d for a launch constant d through a host-made reciprocal (mg = floor(2^32 / d) + 1: the quotient estimate is
// exact or one too large for any 32-bit x; one correction step).  The compiler's own sequence for a run-time divisor is ~25
// dependent scalar / transcendental instructions per division, three of them at the top of every conv workgroup.
__host__ __device__ __forceinline__ unsigned fast_div_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; }
__device__ __forceinline__ int fast_divmod(int& x, int d, unsigned mg) {
    unsigned q = d <= 1 ? (unsigned)x : __umulhi((unsigned)x, mg);
    int r = x - (int)(q * (unsigned)d);
    if (r < 0) { --q; r += d; }
    x = (int)q;
    return r;
}

// ---- bf16 <-> fp32 (round to nearest even, same as torch's .to(bfloat16)) ----
__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
__host__ __device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// two fp32 -> packed bf16 in ONE instruction (gfx950 v_cvt_pk_bf16_f32, round to nearest even): the software form above costs
// ~6 vector-ALU operations per element, which made every bf16 epilogue visibly slower than its fp16 twin (warp sweep 196 vs 158 us)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_hw_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t x) { return __uint_as_float(x << 16); }
__device__ __forceinline__ float bf16hi(uint32_t x) { return __uint_as_float(x & 0xffff0000u); }

// ---- fp16 <-> fp32: round to nearest even, SATURATING at +-65504 (a stored activation never becomes inf) ----
__host__ __device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    const uint32_t sign = (c.u >> 16) & 0x8000u;
    uint32_t x = c.u & 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);            // NaN
    if (x > 0x477fe000u) return (uint16_t)(sign | 0x7bffu);            // |f| > 65504 (incl. inf): saturate
    if (x < 0x38800000u) {                                             // |f| < 2^-14: subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;                    // < 2^-25
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - (int)(x >> 23);
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
}
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16lo(uint32_t x) { return (float)__builtin_bit_cast(h2_t, x)[0]; }
__device__ __forceinline__ float f16hi(uint32_t x) { return (float)__builtin_bit_cast(h2_t, x)[1]; }
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    h2_t v;
    v[0] = (_Float16)__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f);
    v[1] = (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f);
    return __builtin_bit_cast(uint32_t, v);
}

// Saturating fp32 -> fp16 stores without the two v_med3 clamps per pair: with MODE.FP16_OVFL set (once per wave, at kernel entry)
// the hardware conversion clamps an overflowing result to +-65504 itself.  Kernels that call fp16_ovfl_mode() may pack with
// pack_f16x2_ovfl(); everything else keeps pack_f16x2().
// Exception: the mode clamps OVERFLOWING FINITE results only -- a true +-inf input stays +-inf (pack_f16x2's med3 clamp stores
// +-65504 for it).  On the engine's own path no stored tensor can hold inf (every 16-bit store saturates), so the difference is
// reachable only through caller-supplied feature maps that already contain inf; tests/test_gpu_warp_cost.py pins that behaviour.
// The conversions carry no dependency on MODE, so nothing may be scheduled across the s_setreg: sched_barrier right behind it.
// Measured on gfx950 (scripts/dev/nan_probe.py): with the bit set, v_mfma_f32_16x16x32_f16 no longer propagates a NaN operand
// (the stride-2 depth sweep, the only MFMA kernel that kept the bit set for its whole lifetime, returned finite values for a NaN
// input).  MFMA kernels therefore raise the bit only around their store conversions: fp16_ovfl_mode(true) ... (false).
__device__ __forceinline__ void fp16_ovfl_mode(bool on = true) {
    __builtin_amdgcn_sched_barrier(0);
    if (on) __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1);   // hwreg(HW_REG_MODE, 23, 1)
    else __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 0);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ uint32_t pack_f16x2_ovfl(float lo, float hi) {
    h2_t v;
    v[0] = (_Float16)lo;
    v[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
// ReLU floor max(x, lo) that PROPAGATES NaN like torch.relu / the reference's modules do (v_max / v_med3 / fmaxf return the
// non-NaN operand, which turned a diverged activation into -inf or 0 and hid it from isfinite checks): v_cmp + v_cndmask.
// lo = -inf is the "no ReLU" constant of the conv epilogues (x < -inf is never true).
__device__ __forceinline__ float relu_floor(float x, float lo) { return x < lo ? lo : x; }
__device__ __forceinline__ float clamp_lo(float x, float lo) { return relu_floor(x, lo); }

// 16-bit storage type tags (both are raw uint16 in memory)
struct bf16_t { uint16_t bits; };
struct f16_t { uint16_t bits; };
static_assert(sizeof(bf16_t) == 2 && sizeof(f16_t) == 2, "16-bit storage");

// 8 consecutive channels of one texel / voxel
struct f32x8 { float v[8]; };

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = PSCV_F32;
    __device__ static __forceinline__ f32x8 load8(const float* p) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        f32x8 r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
    __device__ static __forceinline__ void store8(float* p, const f32x8& r) {
        *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
    }
    __device__ static __forceinline__ void store2(float* p, float a, float b) {
        *reinterpret_cast<float2*>(p) = make_float2(a, b);
    }
};
template <typename H> struct Half16;   // per-format pack / unpack of a 32-bit word holding two elements
template <> struct Half16<bf16_t> {
    static constexpr int dtype = PSCV_BF16;
    __device__ static __forceinline__ float lo(uint32_t x) { return bf16lo(x); }
    __device__ static __forceinline__ float hi(uint32_t x) { return bf16hi(x); }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
    __device__ static __forceinline__ uint32_t pack_ovfl(float a, float b) { return pack_bf16x2(a, b); }
    __device__ static __forceinline__ float one(uint16_t v) { return bf16_to_f32(v); }
    __host__ __device__ static __forceinline__ uint16_t bits(float f) { return f32_to_bf16(f); }
};
template <> struct Half16<f16_t> {
    static constexpr int dtype = PSCV_F16;
    __device__ static __forceinline__ float lo(uint32_t x) { return f16lo(x); }
    __device__ static __forceinline__ float hi(uint32_t x) { return f16hi(x); }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_f16x2(a, b); }
    __device__ static __forceinline__ uint32_t pack_ovfl(float a, float b) { return pack_f16x2_ovfl(a, b); }   // needs fp16_ovfl_mode()
    __device__ static __forceinline__ float one(uint16_t v) { return f16lo((uint32_t)v); }
    __host__ __device__ static __forceinline__ uint16_t bits(float f) { return f32_to_f16_bits(f); }
};

template <typename H> struct Elem16 {
    static constexpr int dtype = Half16<H>::dtype;
    __device__ static __forceinline__ f32x8 load8(const H* p) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        f32x8 r;
        r.v[0] = Half16<H>::lo(a.x); r.v[1] = Half16<H>::hi(a.x);
        r.v[2] = Half16<H>::lo(a.y); r.v[3] = Half16<H>::hi(a.y);
        r.v[4] = Half16<H>::lo(a.z); r.v[5] = Half16<H>::hi(a.z);
        r.v[6] = Half16<H>::lo(a.w); r.v[7] = Half16<H>::hi(a.w);
        return r;
    }
    __device__ static __forceinline__ void store8(H* p, const f32x8& r) {
        uint4 a;
        a.x = Half16<H>::pack(r.v[0], r.v[1]);
        a.y = Half16<H>::pack(r.v[2], r.v[3]);
        a.z = Half16<H>::pack(r.v[4], r.v[5]);
        a.w = Half16<H>::pack(r.v[6], r.v[7]);
        *reinterpret_cast<uint4*>(p) = a;
    }
    __device__ static __forceinline__ void store2(H* p, float a, float b) {
        *reinterpret_cast<uint32_t*>(p) = Half16<H>::pack(a, b);
    }
};
template <> struct Elem<bf16_t> : Elem16<bf16_t> {};
template <> struct Elem<f16_t> : Elem16<f16_t> {};

}  // namespace pscv
