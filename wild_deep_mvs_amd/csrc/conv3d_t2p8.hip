// Transposed 3x3x3 convolution, stride 2, 16 -> 8 channels (MVSNet CostRegNet.conv11 -- the only decoder layer that
// writes the full-resolution volume -- and the Vis U-Net decoder), with parity-pair packed MFMA rows.  gfx950.
//
// A stride-2 transposed conv splits into 8 output-parity classes (conv3d.hip).  With C_out = 8 only half of the 16
// MFMA rows carry channels, so the two W-parities of a (pd, ph) class share one MFMA tile: rows 0-7 = output
// x = 2*ix, rows 8-15 = output x = 2*ix + 1.  With C_in = 16 one k-step (K = 32) holds the two input taps
// ix and ix + 1: rows 0-7 use [W(kw=1) | 0], rows 8-15 use [W(kw=2) | W(kw=0)].  9 MFMAs per 16 input voxels cover
// all 8 classes (14 before), all 9 A fragments stay in 36 VGPRs, and the 4 lanes that share an input voxel store
// 32 contiguous bytes (two neighbouring output voxels x 8 channels) instead of 8-byte pieces at a 32-byte stride.
//
// Replaces (fdarmon/wild_deep_mvs): Sequential(ConvTranspose3d(16, 8, k3, p1, op1, s2), BatchNorm3d, ReLU) + skip add
// models/MVSNet/model.py:67-70,81; ConvTranspose3d(16, 8) of the Vis U-Net models/VisMVSNet/nn_utils.py:232.
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 tp_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 tp_f16x8;
typedef __attribute__((ext_vector_type(4))) float tp_f32x4;

template <typename H> struct TpMfma;
template <> struct TpMfma<bf16_t> {
    __device__ static __forceinline__ tp_f32x4 run(const uint4& a, const uint4& b, const tp_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tp_bf16x8, a), __builtin_bit_cast(tp_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct TpMfma<f16_t> {
    __device__ static __forceinline__ tp_f32x4 run(const uint4& a, const uint4& b, const tp_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tp_f16x8, a), __builtin_bit_cast(tp_f16x8, b), c, 0, 0, 0);
    }
};

struct Tp8Args {
    const uint16_t* in;
    const uint16_t* wpk;     // [9 steps][64 lanes][8]; step order: (pd,ph) = (0,0) | (0,1): th 0,1 | (1,0): td 0,1 | (1,1): td,th
    const float* scale;
    const float* bias;
    const float* floor;
    const uint16_t* skip;
    void* out;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, Di, Hi, Wi;
    int epi;
    int ntd, nth, ntw;
    unsigned mg_td, mg_th, mg_tw;
};

constexpr int TP_TD = 2, TP_TH = 4, TP_BD = TP_TD + 1, TP_BH = TP_TH + 1, TP_BW = 17;
constexpr int TP_VS = 32;   // 16 ch x 2 B, no pad: conflict-free for the lane groups of ds_read_b128 (conv3d.hip, conv_vs; 48 was 2-way)
PSCV_PROF_BUFFER(t2p8)

template <typename H>
__global__ __launch_bounds__(256) void conv3d_t2p8_kernel(const Tp8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[TP_BD * TP_BH * TP_BW * TP_VS];
    constexpr int NVOX = TP_BD * TP_BH * TP_BW;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot;
    const int tw_i = fast_divmod(wg, a.ntw, a.mg_tw);
    const int th_i = fast_divmod(wg, a.nth, a.mg_th);
    const int td_i = fast_divmod(wg, a.ntd, a.mg_td);
    const int b = wg;
    const int t0d = td_i * TP_TD, t0h = th_i * TP_TH, t0w = tw_i * 16;

    const int tid = threadIdx.x;
    PSCV_PROF_BEGIN
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;

    // ---- output / skip addressing: rows of 16 input voxels have wave-uniform (od, oh); a lane adds one precomputed 32-bit
    // element offset (its output x = 2 iw + (g >> 1), channel group (g & 1) * 4).  The skip values of the wave's 8 output rows
    // are requested before the brick, so the workgroup waits for memory once.
    const int c0 = (g & 1) * 4;
    const int ox = 2 * (t0w + n) + (g >> 1);
    const bool lane_ok = t0w + n < a.Wi;
    const unsigned lane_out = (unsigned)(ox * a.out_cs + c0), lane_skip = (unsigned)(ox * a.skip_cs + c0);
    const unsigned lane_out8 = (unsigned)(ox * a.out_cs);                      // 16-byte stores: all 8 channels of the voxel
    const bool wide = !a.out_f32 && !((a.out_cs | a.out_co) & 7);              // workgroup-uniform
    const long plane_out = (long)Ho * Wo * a.out_cs, plane_skip = (long)Ho * Wo * a.skip_cs;
    const int row_out = Wo * a.out_cs, row_skip = Wo * a.skip_cs;
    const int bDo = b * Do;
    uint2 skv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = wave * 2 + i;
        const int id = t0d + mt / TP_TH, ih = t0h + mt % TP_TH;
        const bool in_ok = id < a.Di && ih < a.Hi;     // wave-uniform
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            skv[i][cl] = make_uint2(0u, 0u);
            if (a.skip && in_ok) {
                const uint16_t* row = a.skip + ((long)(bDo + 2 * id + (cl >> 1)) * plane_skip + (long)(2 * ih + (cl & 1)) * row_skip + a.skip_co);
                if (lane_ok) skv[i][cl] = *reinterpret_cast<const uint2*>(row + lane_skip);
            }
        }
    }

    // ---- stage the 3 x 5 x 17 input brick: a thread owns one (row, column, channel half) position of every plane ----
    {
        constexpr int PC = TP_BH * TP_BW * 2;     // 170 chunks per plane
        const int v = tid >> 1, cc = tid & 1;
        const int bh = v / TP_BW, bw = v - bh * TP_BW;
        const int gh = t0h + bh, gw = t0w + bw;
        const bool ok = tid < PC && gh < a.Hi && gw < a.Wi;
        const unsigned goff = ok ? (unsigned)(gh * a.Wi + gw) * (unsigned)(a.in_cs * 2) + (unsigned)(cc * 16) : 0u;
        const unsigned long plane_bytes = (unsigned long)a.Hi * a.Wi * a.in_cs * 2;
        const char* inb = reinterpret_cast<const char*>(a.in + (long)b * a.Di * a.Hi * a.Wi * a.in_cs + a.in_co);
        uint4 val[TP_BD];
#pragma unroll
        for (int p = 0; p < TP_BD; ++p) {
            const int gd = t0d + p;                                    // wave-uniform
            val[p] = make_uint4(0u, 0u, 0u, 0u);
            if (gd < a.Di && ok) val[p] = *reinterpret_cast<const uint4*>(inb + (unsigned long)gd * plane_bytes + goff);
        }
        PSCV_STAMP(0)
        if (tid < PC) {
#pragma unroll
            for (int p = 0; p < TP_BD; ++p)
                *reinterpret_cast<uint4*>(smem + (p * (TP_BH * TP_BW) + v) * TP_VS + cc * 16) = val[p];
        }
    }
    uint4 wf[9];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.wpk);
#pragma unroll
        for (int s = 0; s < 9; ++s) wf[s] = wp[s * 64 + lane];
    }
    float sc[4], bi[4], fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = a.scale ? a.scale[c0 + k] : 1.0f;
        bi[k] = a.bias ? a.bias[c0 + k] : 0.0f;
        fl[k] = (a.epi & PSCV_EPI_RELU_PRE) ? (a.floor ? a.floor[c0 + k] : 0.0f) : -__builtin_inff();   // ReLU switches as clamps
    }
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    PSCV_STAMP_WAIT(1)
    __syncthreads();
    PSCV_STAMP(2)

    // B operand of lane (n, g): input voxel column n + (g >> 1), channel half g & 1
    const int lane_off = (n + (g >> 1)) * TP_VS + (g & 1) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = wave * 2 + i;
        const int td = mt / TP_TH, th = mt % TP_TH;
        const int id = t0d + td, ih = t0h + th;
        const bool in_ok = id < a.Di && ih < a.Hi;                    // wave-uniform
        int step = 0;
#pragma unroll
        for (int pd = 0; pd < 2; ++pd) {
            // both H-parities of this D-parity first, then ONE 16-byte store per lane: lanes (g, g ^ 1) hold channels 0-3 / 4-7 of
            // the same output voxel, so the row pairs swap halves (`v_permlane16_swap`: odd rows of the first operand <-> even rows
            // of the second) -- the even row ends up with all 8 channels of the ph = 0 voxel, the odd row with those of the ph = 1
            // voxel.  8-byte stores kept this kernel store-issue bound on large volumes (2.2 TB/s at 256 x 144 x 200).
            uint2 pk[2];
            float yf[2][4];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                tp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sd = 0; sd <= pd; ++sd)
#pragma unroll
                    for (int sh = 0; sh <= ph; ++sh) {
                        const int od = (pd && sd == 0) ? 1 : 0, oh = (ph && sh == 0) ? 1 : 0;   // sub 0 = kernel index 0 at input offset +1
                        const uint4 xf = *reinterpret_cast<const uint4*>(smem + (((td + od) * TP_BH + th + oh) * TP_BW) * TP_VS + lane_off);
                        acc = TpMfma<H>::run(wf[step], xf, acc);
                        ++step;
                    }
                const uint2 sv = skv[i][pd * 2 + ph];
                float* y = yf[ph];
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = relu_floor(fmaf(acc[k], sc[k], bi[k]), fl[k]);
                y[0] = relu_floor(y[0] + Half16<H>::lo(sv.x), lo_post); y[1] = relu_floor(y[1] + Half16<H>::hi(sv.x), lo_post);
                y[2] = relu_floor(y[2] + Half16<H>::lo(sv.y), lo_post); y[3] = relu_floor(y[3] + Half16<H>::hi(sv.y), lo_post);
                pk[ph] = make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
            }
            PSCV_STAMP(3)
            if (in_ok) {
                const long orow0 = (long)(bDo + 2 * id + pd) * plane_out + (long)(2 * ih) * row_out + a.out_co;
                if (wide) {
                    const auto rx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                    const auto ry = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                    if (lane_ok)
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + orow0 + (long)(g & 1) * row_out + lane_out8) =
                            make_uint4(rx[0], ry[0], rx[1], ry[1]);
                } else if (lane_ok) {
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        const long orow = orow0 + (long)ph * row_out;
                        if (a.out_f32)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + orow + lane_out) = make_float4(yf[ph][0], yf[ph][1], yf[ph][2], yf[ph][3]);
                        else
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + orow + lane_out) = pk[ph];
                    }
                }
            }
            PSCV_STAMP(4)
        }
    }
    PSCV_STAMP_WAIT(5)
    PSCV_PROF_END(t2p8, blockIdx.x)
}

}  // namespace pscv

PSCV_PROF_EXPORT(t2p8)

int pscv_conv3d_t2p8_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed,
                            const float* scale, const float* bias, const float* floor, const void* skip, int skip_cstride,
                            int skip_coff, void* out, int out_cstride, int out_coff, int out_dtype, int B, int Di, int Hi,
                            int Wi, int epi_flags, hipStream_t st) {
    using namespace pscv;
    Tp8Args a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = packed; a.scale = scale; a.bias = bias; a.floor = floor;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.out_cs = out_cstride; a.out_co = out_coff; a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.epi = epi_flags;
    a.ntd = (Di + TP_TD - 1) / TP_TD; a.nth = (Hi + TP_TH - 1) / TP_TH; a.ntw = (Wi + 15) / 16;
    a.mg_td = fast_div_magic(a.ntd); a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw);
    const long nblk = (long)B * a.ntd * a.nth * a.ntw;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(t2p8): bad grid %ld", nblk); return -1; }
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(conv3d_t2p8_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3d_t2p8_kernel<f16_t>, dim3((unsigned)nblk), dim3(256), 0, st, a);
    return 0;
}
