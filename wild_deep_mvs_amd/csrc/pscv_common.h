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
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float bf16lo(uint32_t x) { return __uint_as_float(x << 16); }
__device__ __forceinline__ float bf16hi(uint32_t x) { return __uint_as_float(x & 0xffff0000u); }

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
template <> struct Elem<uint16_t> {
    static constexpr int dtype = PSCV_BF16;
    __device__ static __forceinline__ f32x8 load8(const uint16_t* p) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        f32x8 r;
        r.v[0] = bf16lo(a.x); r.v[1] = bf16hi(a.x);
        r.v[2] = bf16lo(a.y); r.v[3] = bf16hi(a.y);
        r.v[4] = bf16lo(a.z); r.v[5] = bf16hi(a.z);
        r.v[6] = bf16lo(a.w); r.v[7] = bf16hi(a.w);
        return r;
    }
    __device__ static __forceinline__ void store8(uint16_t* p, const f32x8& r) {
        uint4 a;
        a.x = pack_bf16x2(r.v[0], r.v[1]);
        a.y = pack_bf16x2(r.v[2], r.v[3]);
        a.z = pack_bf16x2(r.v[4], r.v[5]);
        a.w = pack_bf16x2(r.v[6], r.v[7]);
        *reinterpret_cast<uint4*>(p) = a;
    }
    __device__ static __forceinline__ void store2(uint16_t* p, float a, float b) {
        *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b);
    }
};

}  // namespace pscv
