// Fused plane-sweep warp + cost for 32-channel 16-bit feature maps, quad mapping (gfx950).
//
// Why a second mapping: with 2 lanes per voxel (warp_cost.hip) the sweep saturates the per-CU vector L1 -- rocprofv3
// shows ~1 cache access per clock per CU (TCP_TOTAL_CACHE_ACCESSES / cycles / CU = 0.94, profiles/) because a quad of
// lanes (the unit the L1 serves per clock, up to 64 contiguous bytes) straddles two texels and uses only 32 bytes of
// each access.  Here the four lanes of a quad own the four 16-byte channel chunks of ONE 64-byte texel, so every tap
// is exactly one full-width L1 access: 4 accesses per (voxel, view), the minimum for a 2x2 gather (-37 %).
//
// To keep the coordinate arithmetic at "once per two lanes" (the sweep is also close to VALU-issue bound) a quad works
// on a PAIR of voxels: the same reference pixel on two consecutive depth planes.  Lanes 0-1 compute the sample
// position of plane A, lanes 2-3 that of plane B; offsets and bilinear weights are then broadcast inside the quad
// with DPP quad_perm moves (full-rate VALU, no LDS), and every lane blends its 8 channels of both voxels.
//
//   block = 256 threads = 64 quads = 64 reference pixels x PPD depth planes (two per iteration)
//   wave  = 16 x-adjacent pixels: each tap instruction reads 16 x 64 B, each store writes 1 KiB contiguous
//
// (Four voxels per quad with the records exchanged through LDS was tried: 19 % fewer vector-ALU instructions but
// 175 us against 156 us -- at three waves per SIMD the LDS round trip in front of the taps is no longer hidden.)
//
// Semantics and citations are those of warp_cost.hip (the two kernels produce identical bits; tests/).
#include <type_traits>

#include "warp_common.h"

namespace pscv {

typedef float q2_f2 __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    // (old = src: every lane is written, and the compiler need not materialise a separate "old" register)
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
constexpr int Q2_FROM_A = 0x00;   // quad_perm [0,0,0,0]: every lane of the quad reads quad lane 0
constexpr int Q2_FROM_B = 0xAA;   // quad_perm [2,2,2,2]

// Keeps a 16-byte tap load where it was written: without it the compiler merges the interior / border branches of the
// bf16 kernels (whose blend is plain C) into a shared tail, sinks the loads there and splits them into dword loads
// with 64-bit vector addresses (measured: 469 us instead of 160).
__device__ __forceinline__ void q2_pin(uint4& t) { asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w)); }

// four taps (16 B each) of one voxel -> 8 blended channels
template <typename TIn>
__device__ __forceinline__ void q2_mix8(const uint4 (&t)[4], const float (&w)[4], float (&o)[8]) {
    const uint32_t aw[4] = {t[0].x, t[0].y, t[0].z, t[0].w}, bw[4] = {t[1].x, t[1].y, t[1].z, t[1].w};
    const uint32_t cw[4] = {t[2].x, t[2].y, t[2].z, t[2].w}, dw[4] = {t[3].x, t[3].y, t[3].z, t[3].w};
    if constexpr (Half16<TIn>::dtype == PSCV_F16) {
        // v_fma_mix_f32: half -> float conversion fused into the fp32 FMA; stage-wise so that neighbouring
        // instructions are independent
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

template <typename TOut> __device__ __forceinline__ void q2_store8(char* p, const float (&o)[8]) {
    if constexpr (sizeof(TOut) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p + 16) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(Half16<TOut>::pack(o[0], o[1]), Half16<TOut>::pack(o[2], o[3]),
                                                   Half16<TOut>::pack(o[4], o[5]), Half16<TOut>::pack(o[6], o[7]));
    }
}

template <typename TIn, typename TOut, int GEOM, int COST>
__global__ __launch_bounds__(256, 4) void warp_cost_q2_kernel(const WarpArgs a) {
    constexpr int C = 32, PPB = 64, PIXB = 64;
    constexpr int RQ = (GEOM == PSCV_GEOM_HOMOG) ? 2 : 1;
    constexpr int OB = (int)sizeof(TOut);
    constexpr bool VAR = COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP;

    // XCD-aware bijective remap (see warp_cost.hip): XCD k owns a contiguous band of reference pixels
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    const int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
    const int pb = __builtin_amdgcn_readfirstlane(wg / a.n_dchunks);
    const int dc = __builtin_amdgcn_readfirstlane(wg - pb * a.n_dchunks);
    const int b = __builtin_amdgcn_readfirstlane(pb / a.npb_batch);
    const int pbb = __builtin_amdgcn_readfirstlane(pb - b * a.npb_batch);

    const int tid = threadIdx.x;
    __shared__ float cam_lds[PSCV_MAX_SRC * PSCV_CAM_FLOATS];
    extern __shared__ __attribute__((aligned(16))) float4 ray_lds[];   // [n_src][PPB][RQ]
    for (int i = tid; i < a.n_src * PSCV_CAM_FLOATS; i += 256) {
        const int v = i / PSCV_CAM_FLOATS, k = i - v * PSCV_CAM_FLOATS;
        cam_lds[i] = a.cams[((long)v * a.B + b) * PSCV_CAM_FLOATS + k];
    }
    __syncthreads();

    const int hw = a.h * a.w;
    const int pl = tid >> 2, l = tid & 3;
    const bool mineB = (l & 2) != 0;          // this lane computes the sample position of plane B of the pair
    const unsigned chb = (unsigned)l * 16u;   // byte offset of this lane's 8-channel chunk inside a texel
    int pflat = pbb * PPB + pl;
    const bool active = pflat < hw;
    pflat = active ? pflat : hw - 1;
    const int y = pflat / a.w;
    const int x = pflat - y * a.w;
    const float off = (GEOM == PSCV_GEOM_HOMOG) ? 0.5f : 0.0f;   // homography.py:78-79 half-pixel centres
    const float px = (float)x + off, py = (float)(y + a.ref_y0) + off;

    // depth-independent ray terms once per (pixel, view); the four lanes of a quad write identical values and only
    // their own wave reads the slot back (program order suffices)
    for (int v = 0; v < a.n_src; ++v) {
        const float* cam = cam_lds + v * PSCV_CAM_FLOATS;
        float4* sp = ray_lds + (v * PPB + pl) * RQ;
        const float ax = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ay = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float az = fmaf(cam[7], py, cam[6] * px) + cam[8];
        if (GEOM == PSCV_GEOM_PROJ) {
            sp[0] = make_float4(ax, ay, az, 0.0f);
        } else {
            const float bx = fmaf(cam[10], py, cam[9] * px) + cam[11];
            const float by = fmaf(cam[13], py, cam[12] * px) + cam[14];
            const float bz = fmaf(cam[16], py, cam[15] * px) + cam[17];
            sp[0] = make_float4(ax, ay, az, bx);
            sp[1] = make_float4(by, bz, 0.0f, 0.0f);
        }
    }

    float rf[8];
    q2_f2 rf2[4], rfsq[4];
    if (COST != PSCV_COST_WARP_ONLY) {
        const f32x8 t = Elem<TIn>::load8(reinterpret_cast<const TIn*>(a.ref) + ((long)b * hw + pflat) * C + l * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) rf[j] = t.v[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) { rf2[j] = q2_f2{rf[2 * j], rf[2 * j + 1]}; rfsq[j] = rf2[j] * rf2[j]; }
    }

    const int d0 = dc * a.ppd;
    const int d1 = min(a.D, d0 + a.ppd);
    const float invN = 1.0f / (float)(a.n_src + 1);
    const float invN2 = 1.0f / ((float)(a.n_src + 1) * (float)(a.n_src + 1));
    const unsigned long img_bytes_v = (unsigned long)b * a.hs * a.ws * PIXB;
    const unsigned long img_bytes = ((unsigned long)__builtin_amdgcn_readfirstlane((unsigned)(img_bytes_v >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)img_bytes_v);
    const unsigned row_bytes = (unsigned)a.ws * PIXB;
    char* const out = reinterpret_cast<char*>(a.out);
    // bytes of one (b, d) plane of the output and this lane's offset inside it
    constexpr int OCH = (COST == PSCV_COST_GROUPCORR) ? C / 4 : C;             // output channels per voxel
    const unsigned long plane_bytes = (unsigned long)hw * OCH * OB;
    const unsigned lane_out = (unsigned)pflat * (OCH * OB) + (unsigned)l * (OCH / 4 * OB);
    const unsigned long view_bytes = (unsigned long)a.out_view_stride * OB;

    for (int d = d0; d < d1; d += 2) {
        const int dB = min(d + 1, d1 - 1);          // odd tail: plane B repeats plane A and is not stored
        const bool storeB = d + 1 < d1;
        const int dm = mineB ? dB : d;
        const float dval = a.depth_per_pixel ? a.depth[(long)b * a.depth_bstride + (long)dm * hw + pflat]
                                             : a.depth[(long)b * a.depth_bstride + dm];
        const float inv_d = (GEOM == PSCV_GEOM_HOMOG) ? __builtin_amdgcn_rcpf(dval + 1e-9f) : 0.0f;
        char* const outA = out + ((unsigned long)b * a.D + d) * plane_bytes;     // wave-uniform
        char* const outB = out + ((unsigned long)b * a.D + dB) * plane_bytes;

        q2_f2 sA[4], qA[4], sB[4], qB[4];   // variance: sum, sum of squares; softmin: sum e*diff (s only)
        float sum_eA = 0.0f, sum_eB = 0.0f;
        if (!VAR) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sA[j] = q2_f2{0.f, 0.f}; sB[j] = q2_f2{0.f, 0.f}; }
        }

        auto one_view = [&](const int v, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const float* cam = cam_lds + v * PSCV_CAM_FLOATS;
            const float4* ray = ray_lds + (v * PPB + pl) * RQ;
            float hx, hy, hz;
            if (GEOM == PSCV_GEOM_PROJ) {
                const float4 r0 = ray[0];
                hx = fmaf(r0.x, dval, cam[9]);
                hy = fmaf(r0.y, dval, cam[10]);
                hz = fmaf(r0.z, dval, cam[11]);
            } else {
                const float4 r0 = ray[0], r1 = ray[1];
                hx = fmaf(-r0.w, inv_d, r0.x);
                hy = fmaf(-r1.x, inv_d, r0.y);
                hz = fmaf(-r1.y, inv_d, r0.z);
            }
            const bool front = hz > 0.0f;
            const float inv_z = __builtin_amdgcn_rcpf(GEOM == PSCV_GEOM_HOMOG ? fmaxf(hz, 1e-9f) : hz);
            float u = front ? hx * inv_z : -10.0f;
            float w_ = front ? hy * inv_z : -10.0f;
            if (GEOM == PSCV_GEOM_HOMOG) { u *= a.sx; w_ *= a.sy; }
            const float ix = __builtin_amdgcn_fmed3f(u, a.xlo, a.xhi);
            const float iy = __builtin_amdgcn_fmed3f(w_, a.ylo, a.yhi);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float fx = ix - x0f, fy = iy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;

            const char* img = reinterpret_cast<const char*>(a.src[v]) + img_bytes;
            const bool interior = (unsigned)x0 < (unsigned)(a.ws - 1) && (unsigned)y0 < (unsigned)(a.hs - 1);
            // (loads AND blend inside each branch: with a shared tail the compiler merges the two address forms and
            //  splits the taps into dword loads)
            uint4 tA[4], tB[4];
            float wA[4], wB[4], wvA[8], wvB[8];
            if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
                Taps t;
                make_taps<true, PIXB>(fx, fy, x0, y0, a.hs, a.ws, 0u, t);
                wA[0] = dpp_f<Q2_FROM_A>(t.w00); wA[1] = dpp_f<Q2_FROM_A>(t.w01); wA[2] = dpp_f<Q2_FROM_A>(t.w10); wA[3] = dpp_f<Q2_FROM_A>(t.w11);
                wB[0] = dpp_f<Q2_FROM_B>(t.w00); wB[1] = dpp_f<Q2_FROM_B>(t.w01); wB[2] = dpp_f<Q2_FROM_B>(t.w10); wB[3] = dpp_f<Q2_FROM_B>(t.w11);
                const unsigned oA = dpp_u<Q2_FROM_A>(t.o00) | chb, oB = dpp_u<Q2_FROM_B>(t.o00) | chb;
                const char* pA0 = img + oA;
                const char* pA1 = img + (oA + row_bytes);
                const char* pB0 = img + oB;
                const char* pB1 = img + (oB + row_bytes);
                tA[0] = *reinterpret_cast<const uint4*>(pA0); tA[1] = *reinterpret_cast<const uint4*>(pA0 + PIXB);
                tA[2] = *reinterpret_cast<const uint4*>(pA1); tA[3] = *reinterpret_cast<const uint4*>(pA1 + PIXB);
                tB[0] = *reinterpret_cast<const uint4*>(pB0); tB[1] = *reinterpret_cast<const uint4*>(pB0 + PIXB);
                tB[2] = *reinterpret_cast<const uint4*>(pB1); tB[3] = *reinterpret_cast<const uint4*>(pB1 + PIXB);
#pragma unroll
                for (int k = 0; k < 4; ++k) { q2_pin(tA[k]); q2_pin(tB[k]); }
                q2_mix8<TIn>(tA, wA, wvA);
                q2_mix8<TIn>(tB, wB, wvB);
            } else {
                Taps t;
                make_taps<false, PIXB>(fx, fy, x0, y0, a.hs, a.ws, 0u, t);
                wA[0] = dpp_f<Q2_FROM_A>(t.w00); wA[1] = dpp_f<Q2_FROM_A>(t.w01); wA[2] = dpp_f<Q2_FROM_A>(t.w10); wA[3] = dpp_f<Q2_FROM_A>(t.w11);
                wB[0] = dpp_f<Q2_FROM_B>(t.w00); wB[1] = dpp_f<Q2_FROM_B>(t.w01); wB[2] = dpp_f<Q2_FROM_B>(t.w10); wB[3] = dpp_f<Q2_FROM_B>(t.w11);
                tA[0] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_A>(t.o00) | chb));
                tA[1] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_A>(t.o01) | chb));
                tA[2] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_A>(t.o10) | chb));
                tA[3] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_A>(t.o11) | chb));
                tB[0] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_B>(t.o00) | chb));
                tB[1] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_B>(t.o01) | chb));
                tB[2] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_B>(t.o10) | chb));
                tB[3] = *reinterpret_cast<const uint4*>(img + (dpp_u<Q2_FROM_B>(t.o11) | chb));
#pragma unroll
                for (int k = 0; k < 4; ++k) { q2_pin(tA[k]); q2_pin(tB[k]); }
                q2_mix8<TIn>(tA, wA, wvA);
                q2_mix8<TIn>(tB, wB, wvB);
            }

            if (VAR) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const q2_f2 va = q2_f2{wvA[2 * j], wvA[2 * j + 1]}, vb = q2_f2{wvB[2 * j], wvB[2 * j + 1]};
                    if (FIRST) {   // sum starts at the reference feature: (ref + src_0) + src_1 ...  model.py:121-131
                        sA[j] = rf2[j] + va; sB[j] = rf2[j] + vb;
                        qA[j] = __builtin_elementwise_fma(va, va, rfsq[j]);
                        qB[j] = __builtin_elementwise_fma(vb, vb, rfsq[j]);
                    } else {
                        sA[j] += va; sB[j] += vb;
                        qA[j] = __builtin_elementwise_fma(va, va, qA[j]);
                        qB[j] = __builtin_elementwise_fma(vb, vb, qB[j]);
                    }
                }
            } else if (COST == PSCV_COST_SOFTMIN) {
                float dfA[8], dfB[8], partA = 0.0f, partB = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ta = rf[j] - wvA[j], tb = rf[j] - wvB[j];
                    dfA[j] = ta * ta; dfB[j] = tb * tb;
                    partA += dfA[j]; partB += dfB[j];
                }
                // sum over all 32 channels = over the four lanes of the quad (same pairing order as the 2-lane kernel
                // is not required: the reference sums in fp32 over channels, any order is within rounding)
                partA += __shfl_xor(partA, 1, 64); partA += __shfl_xor(partA, 2, 64);
                partB += __shfl_xor(partB, 1, 64); partB += __shfl_xor(partB, 2, 64);
                const float eA = __expf(-a.temp * partA), eB = __expf(-a.temp * partB);
                sum_eA += eA; sum_eB += eB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sA[j] = __builtin_elementwise_fma(q2_f2{eA, eA}, q2_f2{dfA[2 * j], dfA[2 * j + 1]}, sA[j]);
                    sB[j] = __builtin_elementwise_fma(q2_f2{eB, eB}, q2_f2{dfB[2 * j], dfB[2 * j + 1]}, sB[j]);
                }
            } else if (COST == PSCV_COST_GROUPCORR) {
                // 8 groups of 4 channels per voxel; this lane holds groups 2l and 2l+1      nn_utils.py:473-490
                const float a0 = rf[0] * wvA[0] + rf[1] * wvA[1] + rf[2] * wvA[2] + rf[3] * wvA[3];
                const float a1 = rf[4] * wvA[4] + rf[5] * wvA[5] + rf[6] * wvA[6] + rf[7] * wvA[7];
                const float b0 = rf[0] * wvB[0] + rf[1] * wvB[1] + rf[2] * wvB[2] + rf[3] * wvB[3];
                const float b1 = rf[4] * wvB[4] + rf[5] * wvB[5] + rf[6] * wvB[6] + rf[7] * wvB[7];
                if (active) {
                    Elem<TOut>::store2(reinterpret_cast<TOut*>(outA + (unsigned long)v * view_bytes + lane_out), a0, a1);
                    if (storeB) Elem<TOut>::store2(reinterpret_cast<TOut*>(outB + (unsigned long)v * view_bytes + lane_out), b0, b1);
                }
            } else {  // WARP_ONLY
                if (active) {
                    q2_store8<TOut>(outA + (unsigned long)v * view_bytes + lane_out, wvA);
                    if (storeB) q2_store8<TOut>(outB + (unsigned long)v * view_bytes + lane_out, wvB);
                }
            }
        };

        one_view(0, std::true_type{});
        for (int v = 1; v < a.n_src; ++v) one_view(v, std::false_type{});

        if (VAR || COST == PSCV_COST_SOFTMIN) {
            float oA[8], oB[8];
            if (COST == PSCV_COST_VARIANCE) {
                const q2_f2 n1 = q2_f2{invN, invN}, n2 = q2_f2{invN2, invN2};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const q2_f2 ra = qA[j] * n1 - (sA[j] * sA[j]) * n2, rb = qB[j] * n1 - (sB[j] * sB[j]) * n2;
                    oA[2 * j] = ra[0]; oA[2 * j + 1] = ra[1]; oB[2 * j] = rb[0]; oB[2 * j + 1] = rb[1];
                }
            } else if (COST == PSCV_COST_VARIANCE_CVP) {
                const q2_f2 n1 = q2_f2{invN, invN};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const q2_f2 ma = sA[j] * n1, mb = sB[j] * n1;
                    const q2_f2 ra = qA[j] * n1 - ma * ma, rb = qB[j] * n1 - mb * mb;
                    oA[2 * j] = ra[0]; oA[2 * j + 1] = ra[1]; oB[2 * j] = rb[0]; oB[2 * j + 1] = rb[1];
                }
            } else {
                const float ia = 1.0f / (sum_eA + 1e-6f), ib = 1.0f / (sum_eB + 1e-6f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    oA[2 * j] = sA[j][0] * ia; oA[2 * j + 1] = sA[j][1] * ia;
                    oB[2 * j] = sB[j][0] * ib; oB[2 * j + 1] = sB[j][1] * ib;
                }
            }
            if (active) {
                q2_store8<TOut>(outA + lane_out, oA);
                if (storeB) q2_store8<TOut>(outB + lane_out, oB);
            }
        }
    }
}

template <typename TIn, typename TOut, int GEOM, int COST>
static int q2_launch(const WarpArgs& a, int nblk, hipStream_t st) {
    auto kern = warp_cost_q2_kernel<TIn, TOut, GEOM, COST>;
    const size_t ray_bytes = (size_t)a.n_src * 64 * (GEOM == PSCV_GEOM_HOMOG ? 32 : 16);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ray_bytes, st, a);   // <= 32 KiB: inside the default limit
    return 0;
}

template <typename TIn, typename TOut>
static int q2_dispatch(const WarpArgs& a, int geom, int cost, int nblk, hipStream_t st) {
    if (geom == PSCV_GEOM_PROJ) {
        switch (cost) {
            case PSCV_COST_VARIANCE: return q2_launch<TIn, TOut, PSCV_GEOM_PROJ, PSCV_COST_VARIANCE>(a, nblk, st);
            case PSCV_COST_VARIANCE_CVP: return q2_launch<TIn, TOut, PSCV_GEOM_PROJ, PSCV_COST_VARIANCE_CVP>(a, nblk, st);
            case PSCV_COST_SOFTMIN: return q2_launch<TIn, TOut, PSCV_GEOM_PROJ, PSCV_COST_SOFTMIN>(a, nblk, st);
            case PSCV_COST_WARP_ONLY: return q2_launch<TIn, TOut, PSCV_GEOM_PROJ, PSCV_COST_WARP_ONLY>(a, nblk, st);
        }
    } else if (geom == PSCV_GEOM_HOMOG) {
        switch (cost) {
            case PSCV_COST_GROUPCORR: return q2_launch<TIn, TOut, PSCV_GEOM_HOMOG, PSCV_COST_GROUPCORR>(a, nblk, st);
            case PSCV_COST_WARP_ONLY: return q2_launch<TIn, TOut, PSCV_GEOM_HOMOG, PSCV_COST_WARP_ONLY>(a, nblk, st);
        }
    }
    return 1;
}

// Returns 0 if launched, 1 if this configuration is not covered (the caller uses the generic kernel), < 0 on error.
int warp_cost_q2_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st) {
    if (C != 32 || (in_dtype != PSCV_F16 && in_dtype != PSCV_BF16)) return 1;
    if (out_dtype != in_dtype && out_dtype != PSCV_F32) return 1;
    a.npb_batch = (a.h * a.w + 63) / 64;
    const long n_pixblocks = (long)a.npb_batch * a.B;
    int ppd = ppd_override > 0 ? ((ppd_override + 1) & ~1) : 8;
    while (ppd > 2 && n_pixblocks * ((a.D + ppd - 1) / ppd) < 4096) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    const long nblk = n_pixblocks * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_warp_cost(q2): bad grid %ld", nblk); return -1; }
    if (in_dtype == PSCV_F16) return out_dtype == PSCV_F32 ? q2_dispatch<f16_t, float>(a, geom, cost, (int)nblk, st)
                                                           : q2_dispatch<f16_t, f16_t>(a, geom, cost, (int)nblk, st);
    return out_dtype == PSCV_F32 ? q2_dispatch<bf16_t, float>(a, geom, cost, (int)nblk, st)
                                 : q2_dispatch<bf16_t, bf16_t>(a, geom, cost, (int)nblk, st);
}

}  // namespace pscv
