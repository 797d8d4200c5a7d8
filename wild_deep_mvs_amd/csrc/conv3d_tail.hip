// Fused tail of MVSNet's regulariser: transposed conv 16 -> 8 (stride 2) + BatchNorm + ReLU + skip add, then the 1-channel
// `prob` head, as ONE depth sweep.  gfx950.
//
// Unfused, the tail is conv11^T (reads the 16-channel half-resolution volume + the 8-channel skip, writes the 8-channel
// full-resolution volume u11: 142 MB at the headline size), `prob` (reads u11, writes fp32 logits: 79 MB) and the softmax
// pass.  u11 is written once and read once; here it never leaves the CU:
//
//   workgroup = a tile of 8 x 28 output pixels x a chunk of depth planes, 256 threads;
//   PRODUCE   pair step P(i): the odd output plane 2i+1 of input plane i and the even output plane 2i+2 of input plane i+1 --
//             the parity-pair MFMA formulation of conv3d_t2p8.hip (9 MFMAs per 16 input voxels cover all four (pd, ph) classes,
//             rows 0-7 / 8-15 = output x parity) -- over the tile + 1 voxel of halo (6 input rows x 16 input columns), epilogue
//             (affine, ReLU, skip add, 16-bit rounding: the same operation chain, so the same stored bits as the unfused layer),
//             into an 8-slot LDS ring of u11 planes (10 rows x 32 columns x 16 B); voxels outside the volume are written as the
//             zero padding `prob` expects;
//   CONSUME   `prob` on a block of 6 output planes from 8 ring planes: the depth-in-rows (Toeplitz) MFMA formulation of
//             conv3d_c1.hip (18 MFMAs per 16 pixels x 6 planes, same accumulation chains, same bits), fp32 logits to HBM;
//   one block = 3 pair steps (6 new planes) + 1 consume, two barriers; the 16-channel input planes of the NEXT block are staged
//   into a 5-slot LDS ring during the consume phase, their loads and the skip values of the next block are requested one block
//   ahead, so a block never waits on memory it asked for itself.
//
// HBM traffic at the headline size: skip 63 MB x 1.34 (halo) + input 15.7 MB x 1.9 + logits 15.7 MB = ~130 MB against 221 MB.
//
// Replaces (fdarmon/wild_deep_mvs): conv11 = Sequential(ConvTranspose3d(16, 8, k3, p1, op1, s2), BatchNorm3d, ReLU), the skip add
// and `prob` of CostRegNet.forward, models/MVSNet/model.py:67-72,81-82.
#include <type_traits>

#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 tl_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 tl_f16x8;
typedef __attribute__((ext_vector_type(4))) float tl_f32x4;

template <typename H> struct TlMfma;
template <> struct TlMfma<bf16_t> {
    __device__ static __forceinline__ tl_f32x4 run(const uint4& a, const uint4& b, const tl_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, a), __builtin_bit_cast(tl_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct TlMfma<f16_t> {
    __device__ static __forceinline__ tl_f32x4 run(const uint4& a, const uint4& b, const tl_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tl_f16x8, a), __builtin_bit_cast(tl_f16x8, b), c, 0, 0, 0);
    }
};

struct TailArgs {
    const uint16_t* in;       // [B, Di, Hi, Wi, in_cs], channels [in_co, in_co + 16)
    const uint4* w_up;        // T2P8 packing of the [16, 8, 3, 3, 3] transposed weight: [9 steps][64 lanes]
    const float* up_scale;    // [8] each, may be null (1 / 0 / 0)
    const float* up_bias;
    const float* up_floor;
    const uint16_t* skip;     // [B, 2 Di, 2 Hi, 2 Wi, skip_cs], channels [skip_co, skip_co + 8); may be null
    const uint4* w_head;      // S1C1 packing of the [1, 8, 3, 3, 3] head: [18 steps][64 lanes]
    const float* hd_scale;    // [1] each, may be null
    const float* hd_bias;
    const float* hd_floor;
    float* logits;            // [B, D, H, W] fp32
    int in_cs, in_co, skip_cs, skip_co;
    int up_epi, hd_epi;
    int B, Di, Hi, Wi;
    int nth, ntw, ndc, nblocks;   // tiles along h (8 rows) and w (28 columns), depth chunks, 6-plane blocks of the volume
    unsigned mg_th, mg_tw, mg_dc;
    // fused softmax statistics (FUSE instantiations; null = off): per (batch, depth chunk, pixel) the running max, the sum of exponentials,
    // the sum of exp x depth plane and the sum of exp x plane index of the chunk's logits; softargmin_merge_kernel (conv3d_c1.hip) finishes
    const float* depth;       // [B][D] planes, row stride depth_bstride
    long depth_bstride;
    float* part;              // [B][ndc][4][H][W]
};

constexpr int TL_TH = 8, TL_TW = 28;                 // output pixels per tile
constexpr int TL_BR = TL_TH + 2, TL_BC = 32;         // u11 brick: rows 8 s - 1 .. 8 s + 8, columns 28 t - 2 .. 28 t + 29
constexpr int TL_PB = TL_BR * TL_BC * 16;            // bytes per u11 ring plane
constexpr int TL_NSLOT = 8;
constexpr int TL_IR = 7, TL_IC = 17, TL_VS = 32;     // staged input rows 4 s - 1 .. 4 s + 5, columns 14 t - 1 .. 14 t + 15; 32 B, NO pad: voxels one
                                                     // stride apart + chunk g & 1 are conflict-free at 2 mod 4 granules (conv3d.hip, conv_vs; 48 was 2-way)
constexpr int TL_IPB = TL_IR * TL_IC * TL_VS;        // bytes per input ring plane
constexpr int TL_ISLOT = 5;
constexpr int TL_ICH = TL_IR * TL_IC * 2;            // 16-byte chunks per input plane (238)
constexpr int TL_DEPTHS = 96;                        // depth planes of a chunk kept in LDS by the FUSE instantiations (16 blocks of 6)
constexpr int TL_LDS = TL_NSLOT * TL_PB + TL_ISLOT * TL_IPB + TL_DEPTHS * 4;
constexpr int TL_P = 6;                              // output planes per consume block
static_assert(TL_LDS <= 80 * 1024, "two workgroups per CU");

Knob g_tail_nbk = {0, KNOB_TAIL_NBK};                  // pscv_set_tuning("tail_nbk", n): 6-plane blocks per depth chunk (0 = default heuristic)

// UP_POST: the transposed layer has a ReLU after the skip add; HD_CLAMP: the head has any ReLU.  MVSNet's tail has neither: the
// <H, false, false> instantiation carries no dead clamp instructions (a NaN-propagating clamp is a compare + select per value).
template <typename H, bool UP_POST, bool HD_CLAMP, bool FUSE>
__global__ __launch_bounds__(256, 2) void conv3d_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;                           // u11 planes
    unsigned char* const iring = smem + TL_NSLOT * TL_PB;       // input planes

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int D = 2 * a.Di, Hh = 2 * a.Hi, W = 2 * a.Wi;
    const int h0 = thi * TL_TH, w0 = twi * TL_TW;               // tile origin (output pixels); both even
    const int kb0 = (a.nblocks * dci) / a.ndc, kb1 = (a.nblocks * (dci + 1)) / a.ndc;   // this chunk's 6-plane blocks (balanced partition)
    const int nbk = kb1 - kb0;
    const int dbeg = kb0 * TL_P;                                // first output plane of the chunk (even)
    const int i0 = dbeg >> 1;                                   // input plane of output plane dbeg
    const int ih0 = (h0 >> 1) - 1, ix0 = (w0 >> 1) - 1;         // input coordinates of staged row 0 / column 0

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    // ---- produce-phase roles: a unit = (pair step, input row r in 0..5) has four (pd, ph) classes, class id = pd * 2 + ph in
    //      conv3d_t2p8's order.  Waves 0, 1 take classes 0 and 3 (1 + 4 MFMAs), waves 2, 3 classes 1 and 2 (2 + 2 MFMAs); waves 0, 2
    //      the first half of the units of a phase, waves 1, 3 the second half: 18 class epilogues per wave and block, no ragged tail ----
    const bool typeA = wave < 2;

    // ---- weights: the head's fragments and this wave's share of the transposed layer's, resident for the whole sweep ----
    uint4 wu[5], wh[18];
    {
        const int idx[2][5] = {{0, 5, 6, 7, 8}, {1, 2, 3, 4, 4}};
#pragma unroll
        for (int s = 0; s < 5; ++s) wu[s] = a.w_up[(typeA ? idx[0][s] : idx[1][s]) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < 18; ++s) wh[s] = a.w_head[s * 64 + lane];
    const int c0 = (g & 1) * 4;                                  // this lane's channel group of a produced voxel
    float sc[4], bi[4], fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = a.up_scale ? a.up_scale[c0 + k] : 1.0f;
        bi[k] = a.up_bias ? a.up_bias[c0 + k] : 0.0f;
        fl[k] = (a.up_epi & PSCV_EPI_RELU_PRE) ? (a.up_floor ? a.up_floor[c0 + k] : 0.0f) : -__builtin_inff();
    }
    const float e_scale = a.hd_scale ? a.hd_scale[0] : 1.0f, e_bias = a.hd_bias ? a.hd_bias[0] : 0.0f;
    const float lo_pre = (a.hd_epi & PSCV_EPI_RELU_PRE) ? (a.hd_floor ? a.hd_floor[0] : 0.0f) : -__builtin_inff();
    const float lo_post = (a.hd_epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();

    // ---- input staging: chunk id -> (voxel, channel half); raw buffer loads, hardware zero fill outside the image / volume ----
    const unsigned in_plane_b = (unsigned)a.Hi * a.Wi * a.in_cs * 2;
    const unsigned in_plane_sz = in_plane_b - (unsigned)a.in_co * 2;
    const char* inb = reinterpret_cast<const char*>(a.in) + ((unsigned long)b * a.Di * in_plane_b + (unsigned long)a.in_co * 2);
    unsigned igoff[2];
    int iloff[2];
    bool ival[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i;
        const int v = id >> 1, half = id & 1;
        const int r = v / TL_IC, c = v - r * TL_IC;
        const int gh = ih0 + r, gw = ix0 + c;
        ival[i] = id < TL_ICH;
        const bool ok = ival[i] && (unsigned)gh < (unsigned)a.Hi && (unsigned)gw < (unsigned)a.Wi;
        igoff[i] = ok ? ((unsigned)(gh * a.Wi + gw) * (unsigned)a.in_cs + (unsigned)(half * 8)) * 2u : 0x7ffffff0u;
        iloff[i] = v * TL_VS + half * 16;
    }
    auto ifetch = [&](int plane, uint4 (&reg)[2]) {
        const bool pv = plane >= 0 && plane < a.Di;                                              // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(inb + (unsigned long)(pv ? plane : 0) * in_plane_b), (short)0, pv ? (int)in_plane_sz : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) reg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)igoff[i], 0, 0));
    };
    auto istash = [&](int islot, const uint4 (&reg)[2]) {
        unsigned char* sp = iring + islot * TL_IPB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (ival[i]) *reinterpret_cast<uint4*>(sp + iloff[i]) = reg[i];
    };

    // ---- skip values through a buffer descriptor over this batch item's skip volume (< 2 GiB, checked by the host): the lane adds
    //      a 32-bit offset (its output x, channel group; out of range when x is outside the image), the plane / row a scalar one;
    //      a plane or row outside the volume gets an empty descriptor -- the hardware returns zeros, no 64-bit address arithmetic ----
    const int ox = 2 * (ix0 + n) + (g >> 1);
    const bool x_ok = (unsigned)ox < (unsigned)W;
    const unsigned sk_row_b = (unsigned)W * a.skip_cs * 2, sk_plane_b = (unsigned)Hh * sk_row_b;
    const unsigned sk_size = a.skip ? (unsigned)D * sk_plane_b : 0u;
    const char* skb = reinterpret_cast<const char*>(a.skip) + ((unsigned long)b * sk_size + (unsigned long)a.skip_co * 2);
    const unsigned sk_voff = x_ok ? ((unsigned)ox * a.skip_cs + c0) * 2u : 0x7ffffff0u;
    auto skip_load = [&](int op, int orow) -> uint2 {            // op, orow wave-uniform
        const bool ok = (unsigned)op < (unsigned)D && (unsigned)orow < (unsigned)Hh;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(skb), (short)0, ok ? (int)(sk_size - a.skip_co * 2) : 0, 0x00020000);
        const unsigned soff = ok ? (unsigned)op * sk_plane_b + (unsigned)orow * sk_row_b : 0u;
        return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)sk_voff, (int)soff, 0));
    };
    // B operand of the produce MFMAs: staged input column n + (g >> 1), channel half g & 1
    const int pl_off = (n + (g >> 1)) * TL_VS + (g & 1) * 16;
    // ring offset of this lane's produced voxel inside a brick row: column 2 n + (g >> 1), channel half
    const int pw_off = (2 * n + (g >> 1)) * 16 + (g & 1) * 8;

    // output plane of class cls of pair step P(i): pd = cls >> 1 -> plane 2 i + 1 (pd = 1, input plane i) or 2 i + 2 (pd = 0, input plane
    // i + 1).  Output row of (r, ph): 2 (ih0 + r) + ph = brick row 2 r + ph - 1 -- a COMPILE-TIME constant per unrolled unit (r and
    // the wave's classes are template constants of its code path), so rows outside the brick cost nothing.
    auto cls_plane = [&](int i, int cls) { return (cls >> 1) ? 2 * i + 1 : 2 * i + 2; };
    auto epilogue = [&](const tl_f32x4& acc, int op, int rb, const uint2& sv) {     // rb in [0, TL_BR)
        const int orow = h0 - 1 + rb;
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = relu_floor(fmaf(acc[k], sc[k], bi[k]), fl[k]);
        y[0] += Half16<H>::lo(sv.x); y[1] += Half16<H>::hi(sv.x); y[2] += Half16<H>::lo(sv.y); y[3] += Half16<H>::hi(sv.y);
        if (UP_POST) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = relu_floor(y[k], 0.0f);
        }
        uint2 pk = make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
        // outside the volume the head sees its zero padding
        const bool inside = x_ok && (unsigned)op < (unsigned)D && (unsigned)orow < (unsigned)Hh;
        if (!inside) pk = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(ring + ((op - dbeg + 1) & (TL_NSLOT - 1)) * TL_PB + rb * (TL_BC * 16) + pw_off) = pk;
    };
    // skip values of unit (i, R) for this wave's two classes (CA, CB compile-time)
    auto unit_skips = [&](auto ca, auto cb, auto rc, int i, uint2 (&sk)[2]) {
        constexpr int CA = decltype(ca)::value, CB = decltype(cb)::value, R = decltype(rc)::value;
        constexpr int rba = 2 * R + (CA & 1) - 1, rbb = 2 * R + (CB & 1) - 1;
        sk[0] = sk[1] = make_uint2(0u, 0u);
        if (rba >= 0 && rba < TL_BR) sk[0] = skip_load(cls_plane(i, CA), h0 - 1 + rba);
        if (rbb >= 0 && rbb < TL_BR) sk[1] = skip_load(cls_plane(i, CB), h0 - 1 + rbb);
    };
    // classes of conv3d_t2p8's step order: (0,0): w0 (i+1, r) | (0,1): w1 (i+1, r+1), w2 (i+1, r) | (1,0): w3 (i+1, r), w4 (i, r) |
    // (1,1): w5 (i+1, r+1), w6 (i+1, r), w7 (i, r+1), w8 (i, r); same chains, same accumulation order, same bits
    auto produce_unit = [&](auto ta, auto rc, int i, int islot_i, int islot_i1, const uint2 (&sk)[2]) {
        constexpr bool TA = decltype(ta)::value;
        constexpr int R = decltype(rc)::value;
        constexpr int CA = TA ? 0 : 1, CB = TA ? 3 : 2;
        constexpr int rba = 2 * R + (CA & 1) - 1, rbb = 2 * R + (CB & 1) - 1;
        const unsigned char* pi = iring + islot_i * TL_IPB + pl_off;       // input plane i
        const unsigned char* pj = iring + islot_i1 * TL_IPB + pl_off;      // input plane i + 1
        constexpr int ro0 = R * TL_IC * TL_VS, ro1 = (R + 1) * TL_IC * TL_VS;
        tl_f32x4 acca, accb;
        const tl_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (TA) {
            const uint4 xj0 = *reinterpret_cast<const uint4*>(pj + ro0), xj1 = *reinterpret_cast<const uint4*>(pj + ro1);
            const uint4 xi0 = *reinterpret_cast<const uint4*>(pi + ro0), xi1 = *reinterpret_cast<const uint4*>(pi + ro1);
            acca = TlMfma<H>::run(wu[0], xj0, z);
            accb = TlMfma<H>::run(wu[1], xj1, z);
            accb = TlMfma<H>::run(wu[2], xj0, accb);
            accb = TlMfma<H>::run(wu[3], xi1, accb);
            accb = TlMfma<H>::run(wu[4], xi0, accb);
        } else {
            const uint4 xj0 = *reinterpret_cast<const uint4*>(pj + ro0), xj1 = *reinterpret_cast<const uint4*>(pj + ro1);
            const uint4 xi0 = *reinterpret_cast<const uint4*>(pi + ro0);
            acca = TlMfma<H>::run(wu[0], xj1, z);
            acca = TlMfma<H>::run(wu[1], xj0, acca);
            accb = TlMfma<H>::run(wu[2], xj0, z);
            accb = TlMfma<H>::run(wu[3], xi0, accb);
        }
        if (rba >= 0 && rba < TL_BR) epilogue(acca, cls_plane(i, CA), rba, sk[0]);
        if (rbb >= 0 && rbb < TL_BR) epilogue(accb, cls_plane(i, CB), rbb, sk[1]);
    };
    using std::integral_constant;
    // a wave's units of a phase with `NU` units per pair step... as compile-time lists: block = 18 units (pair step u / 6, row u % 6),
    // units 9 UH .. 9 UH + 8; pre-step = 6 units (rows 3 UH .. 3 UH + 2)
    auto block_skips = [&](auto ta, auto uh, int ib, uint2 (&sk)[9][2]) {
        constexpr bool TA = decltype(ta)::value;
        constexpr int UH = decltype(uh)::value;
        using CA = integral_constant<int, TA ? 0 : 1>;
        using CB = integral_constant<int, TA ? 3 : 2>;
#define PSCV_TL_SK(J) unit_skips(CA{}, CB{}, integral_constant<int, (9 * UH + J) % 6>{}, ib + (9 * UH + J) / 6, sk[J]);
        PSCV_TL_SK(0) PSCV_TL_SK(1) PSCV_TL_SK(2) PSCV_TL_SK(3) PSCV_TL_SK(4) PSCV_TL_SK(5) PSCV_TL_SK(6) PSCV_TL_SK(7) PSCV_TL_SK(8)
#undef PSCV_TL_SK
    };
    auto block_produce = [&](auto ta, auto uh, int ib, int islot0, const uint2 (&sk)[9][2]) {
        constexpr int UH = decltype(uh)::value;
#define PSCV_TL_PU(J) { constexpr int ps = (9 * UH + J) / 6; int sa = islot0 + ps; sa = sa >= TL_ISLOT ? sa - TL_ISLOT : sa;   \
                        int sb = sa + 1; sb = sb >= TL_ISLOT ? sb - TL_ISLOT : sb;                                              \
                        produce_unit(ta, integral_constant<int, (9 * UH + J) % 6>{}, ib + ps, sa, sb, sk[J]); }
        PSCV_TL_PU(0) PSCV_TL_PU(1) PSCV_TL_PU(2) PSCV_TL_PU(3) PSCV_TL_PU(4) PSCV_TL_PU(5) PSCV_TL_PU(6) PSCV_TL_PU(7) PSCV_TL_PU(8)
#undef PSCV_TL_PU
    };
    auto pre_skips = [&](auto ta, auto uh, uint2 (&sk)[3][2]) {
        constexpr bool TA = decltype(ta)::value;
        constexpr int UH = decltype(uh)::value;
        using CA = integral_constant<int, TA ? 0 : 1>;
        using CB = integral_constant<int, TA ? 3 : 2>;
        unit_skips(CA{}, CB{}, integral_constant<int, 3 * UH + 0>{}, i0 - 1, sk[0]);
        unit_skips(CA{}, CB{}, integral_constant<int, 3 * UH + 1>{}, i0 - 1, sk[1]);
        unit_skips(CA{}, CB{}, integral_constant<int, 3 * UH + 2>{}, i0 - 1, sk[2]);
    };
    auto pre_produce = [&](auto ta, auto uh, const uint2 (&sk)[3][2]) {
        constexpr int UH = decltype(uh)::value;
        produce_unit(ta, integral_constant<int, 3 * UH + 0>{}, i0 - 1, 0, 1, sk[0]);
        produce_unit(ta, integral_constant<int, 3 * UH + 1>{}, i0 - 1, 0, 1, sk[1]);
        produce_unit(ta, integral_constant<int, 3 * UH + 2>{}, i0 - 1, 0, 1, sk[2]);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
#define PSCV_TL_ROLE(CALL)                                              \
    switch (wave) {                                                     \
        case 0: { auto ta = T_{}; auto uh = I0{}; CALL; } break;         \
        case 1: { auto ta = T_{}; auto uh = I1{}; CALL; } break;         \
        case 2: { auto ta = F_{}; auto uh = I0{}; CALL; } break;         \
        default: { auto ta = F_{}; auto uh = I1{}; CALL; } break;        \
    }

    // ---- prologue: input planes i0 - 1 .. i0 + 3 -> slots 0..4; pre-step P(i0 - 1) (ring planes 0, 1): rows 3 uhalf .. 3 uhalf + 2 ----
    {
        uint4 r5[5][2];
#pragma unroll
        for (int p = 0; p < 5; ++p) ifetch(i0 - 1 + p, r5[p]);
        uint2 sk0[3][2];
        PSCV_TL_ROLE(pre_skips(ta, uh, sk0))
#pragma unroll
        for (int p = 0; p < 5; ++p) istash(p, r5[p]);
        __syncthreads();
        PSCV_TL_ROLE(pre_produce(ta, uh, sk0))
    }
    // skip values of block 0: this wave's units u = 9 uhalf + j (pair step u / 6, row u % 6)
    uint2 sk[9][2];
    PSCV_TL_ROLE(block_skips(ta, uh, i0, sk))
    int islot0 = 1;                       // slot of input plane i0 + 3 k (block k's first plane); planes i0+3k .. i0+3k+3 follow mod 5

    // consume-phase constants (conv3d_c1_sweep_kernel's mapping): wave w owns tile rows 2 w, 2 w + 1; lane group g reads ring planes
    // 4 (g >> 1) + (g & 1) and + 2 of the block's 8-plane window.  Logits through a buffer descriptor over this batch item's volume:
    // lane offset = (row, column, 4 g planes), scalar offset = the block's first plane + r.
    const int pA = 4 * (g >> 1) + (g & 1);
    // FUSE: the chunk's depth planes in LDS (behind the two rings); running softmax statistics of this lane's own logits -- planes
    // d0 + 4 g + r of the pixels (h0 + 2 wave + rr, w0 + 16 ct + n) -- merged across the lane groups g = 0, 1 once per chunk
    float* const sdep = reinterpret_cast<float*>(smem + TL_NSLOT * TL_PB + TL_ISLOT * TL_IPB);
    if (FUSE && tid < TL_DEPTHS) sdep[tid] = a.depth[(long)b * a.depth_bstride + min(dbeg + tid, D - 1)];      // (first read behind block 0's produce barrier)
    float sM[2][2], sZ[2][2], sD[2][2], sI[2][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { sM[rr][ct] = -__builtin_inff(); sZ[rr][ct] = 0.f; sD[rr][ct] = 0.f; sI[rr][ct] = 0.f; }
    const unsigned lg_plane_b = (unsigned)Hh * W * 4;
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.logits) + (unsigned long)b * D * lg_plane_b, (short)0, (int)((unsigned)D * lg_plane_b), 0x00020000);

    for (int k = 0; k < nbk; ++k) {
        const int d0 = dbeg + k * TL_P;
            const int ib = i0 + 3 * k;
        // ---- produce: pair steps P(ib), P(ib + 1), P(ib + 2) -> ring planes 6 k + 2 .. 6 k + 7 ----
        PSCV_TL_ROLE(block_produce(ta, uh, ib, islot0, sk))
        // the next block's input planes (ib + 4 .. ib + 6) and skip values: requested now, used after this block's consume phase
        const bool more = k + 1 < nbk;
        uint4 nx[3][2];
        if (more) {
#pragma unroll
            for (int p = 0; p < 3; ++p) ifetch(ib + 4 + p, nx[p]);
            PSCV_TL_ROLE(block_skips(ta, uh, ib + 3, sk))
        }
        __syncthreads();                  // ring planes 6 k .. 6 k + 7 complete; nobody reads the input ring any more
        // ---- consume: 6 output planes d0 .. d0 + 5 of the tile ----
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int tr = 2 * wave + rr;
            const unsigned char* spA = ring + ((TL_P * k + pA) & (TL_NSLOT - 1)) * TL_PB + (tr * TL_BC + n + 1) * 16;
            const unsigned char* spB = ring + ((TL_P * k + pA + 2) & (TL_NSLOT - 1)) * TL_PB + (tr * TL_BC + n + 1) * 16;
            tl_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc0b = {0.f, 0.f, 0.f, 0.f}, acc1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int off = ((t / 3) * TL_BC + (t % 3)) * 16;
                const uint4 x0 = *reinterpret_cast<const uint4*>(spA + off);
                const uint4 x1 = *reinterpret_cast<const uint4*>(spA + off + 16 * 16);
                const uint4 x2 = *reinterpret_cast<const uint4*>(spB + off);
                const uint4 x3 = *reinterpret_cast<const uint4*>(spB + off + 16 * 16);
                acc0 = TlMfma<H>::run(wh[t], x0, acc0);
                acc1 = TlMfma<H>::run(wh[t], x1, acc1);
                acc0b = TlMfma<H>::run(wh[9 + t], x2, acc0b);
                acc1b = TlMfma<H>::run(wh[9 + t], x3, acc1b);
            }
            acc0 += acc0b;
            acc1 += acc1b;
            // lane (n, g < 2) holds output planes d0 + 4 g + r (r < 4; g = 1: r < 2) of pixels (h0 + tr, w0 + ct * 16 + n)
            const int oh = h0 + tr;
            const unsigned soff0 = (unsigned)d0 * lg_plane_b;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = ct * 16 + n;
                const bool ok = g < 2 && oh < Hh && col < TL_TW && w0 + col < W;
                const unsigned voff = ((unsigned)(oh * W + w0 + col)) * 4u + (unsigned)(4 * g) * lg_plane_b;
                float yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = fmaf(ct ? acc1[r] : acc0[r], e_scale, e_bias);
                    if (HD_CLAMP) y = clamp_lo(clamp_lo(y, lo_pre), lo_post);
                    yv[r] = y;
                    const bool okr = ok && 4 * g + r < TL_P && d0 + 4 * g + r < D;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), lrs, (int)(okr ? voff : 0x7ffffff0u),
                                                          (int)(soff0 + (unsigned)r * lg_plane_b), 0);
                }
                if (FUSE && g < 2) {
                    // fold the block's logits (fp32, as stored) into the lane's statistics (conv3d_c1_sweep_kernel's FUSE = 1 update)
                    bool okp[4];
                    float lmax = -__builtin_inff();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        okp[r] = 4 * g + r < TL_P && d0 + 4 * g + r < D;
                        if (okp[r]) lmax = fmaxf(lmax, yv[r]);
                    }
                    if (lmax > -__builtin_inff()) {
                        const float mo = sM[rr][ct], mn = fmaxf(mo, lmax);
                        const float scl = mo > -__builtin_inff() ? __expf(mo - mn) : 0.0f;
                        float z = sZ[rr][ct] * scl, sd = sD[rr][ct] * scl, si = sI[rr][ct] * scl;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (okp[r]) {
                                const float e = __expf(yv[r] - mn);
                                z += e;
                                sd = fmaf(e, sdep[min(k * TL_P + 4 * g + r, TL_DEPTHS - 1)], sd);
                                si = fmaf(e, (float)(d0 + 4 * g + r), si);
                            }
                        }
                        sM[rr][ct] = mn; sZ[rr][ct] = z; sD[rr][ct] = sd; sI[rr][ct] = si;
                    }
                }
            }
        }
        // the next block's input planes into the slots nobody reads now: planes ib + 4 .. ib + 6 = slots islot0 + 4, + 5, + 6 (mod 5)
        if (more) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                int s = islot0 + 4 + p;
                s = s >= 2 * TL_ISLOT ? s - 2 * TL_ISLOT : s >= TL_ISLOT ? s - TL_ISLOT : s;
                istash(s, nx[p]);
            }
        }
        islot0 += 3;
        islot0 = islot0 >= TL_ISLOT ? islot0 - TL_ISLOT : islot0;
        __syncthreads();                  // consume done (ring planes 6 k .. 6 k + 5 are free), input planes of block k + 1 staged
    }
    if (FUSE) {
        // merge the lane groups g = 0 (planes 0..3 of every block) and g = 1 (planes 4, 5): lane n + 16 -> lane n; lanes g = 0 write
        const long hw = (long)Hh * W;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float m1 = __shfl_down(sM[rr][ct], 16, 64), z1 = __shfl_down(sZ[rr][ct], 16, 64);
                const float d1 = __shfl_down(sD[rr][ct], 16, 64), i1 = __shfl_down(sI[rr][ct], 16, 64);
                const float m0 = sM[rr][ct], mn = fmaxf(m0, m1);
                const float f0 = m0 > -__builtin_inff() ? __expf(m0 - mn) : 0.0f, f1 = m1 > -__builtin_inff() ? __expf(m1 - mn) : 0.0f;
                const int oh = h0 + 2 * wave + rr, col = ct * 16 + n, ow = w0 + col;
                if (g == 0 && oh < Hh && col < TL_TW && ow < W) {
                    float* pp = a.part + (((long)b * a.ndc + dci) * 4) * hw + (long)oh * W + ow;
                    pp[0] = mn;
                    pp[hw] = sZ[rr][ct] * f0 + z1 * f1;
                    pp[2 * hw] = sD[rr][ct] * f0 + d1 * f1;
                    pp[3 * hw] = sI[rr][ct] * f0 + i1 * f1;
                }
            }
    }
}

}  // namespace pscv

void pscv_softargmin_merge_launch(const float* part, const float* logits, int ndc, int B, int D, long hw, float* o_depth, float* o_conf, hipStream_t st);

// Fused tail.  Returns 0 if launched, 1 if this shape is not covered (the caller runs the two layers), negative on error.
// depth != null: the sweep also keeps softmax statistics per depth chunk and a merge launch writes the regressed depth (and the 4-plane
// photometric confidence) -- models/MVSNet/model.py:207-215; workspace: pscv_tail_sweep_workspace(B, Di, Hi, Wi) floats.
extern "C" long pscv_tail_sweep_workspace(int B, int Di, int Hi, int Wi) {
    const int nblocks = (2 * Di + pscv::TL_P - 1) / pscv::TL_P;
    return (long)B * (nblocks / 2 > 0 ? nblocks / 2 : 1) * 4 * (2L * Hi) * (2L * Wi);      // at most one chunk per two blocks
}

extern "C" int pscv_tail_sweep(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed_up, const float* up_scale,
                               const float* up_bias, const float* up_floor, int up_epi, const void* skip, int skip_cstride, int skip_coff,
                               const uint16_t* packed_head, const float* hd_scale, const float* hd_bias, const float* hd_floor, int hd_epi,
                               float* logits, const float* depth, long depth_bstride, float* workspace, long workspace_floats,
                               float* out_depth, float* out_conf, int B, int Di, int Hi, int Wi, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(in && packed_up && packed_head && logits, "pscv_tail_sweep: null pointer argument");
    PSCV_CHECK_ARG(!depth || (workspace && out_depth), "pscv_tail_sweep: the regression needs a workspace and an output");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_tail_sweep: dtype %d", dtype);
    PSCV_CHECK_ARG(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "pscv_tail_sweep: bad sizes");
    PSCV_CHECK_ARG(in_cstride >= in_coff + 16 && (!skip || skip_cstride >= skip_coff + 8), "pscv_tail_sweep: channel slices");
    if ((in_cstride | in_coff) & 7 || (skip && ((skip_cstride | skip_coff) & 3))) return 1;      // 16-byte input chunks, 8-byte skip pieces
    // 32-bit buffer offsets: one input plane, one batch item's skip volume and one batch item's logits below 2 GiB
    if ((long)Hi * Wi * in_cstride * 2 >= (1L << 31) || (long)8 * Di * Hi * Wi * (skip ? skip_cstride : 1) * 2 >= (1L << 31)
        || (long)8 * Di * Hi * Wi * 4 >= (1L << 31)) return 1;
    TailArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.w_up = reinterpret_cast<const uint4*>(packed_up);
    a.up_scale = up_scale; a.up_bias = up_bias; a.up_floor = up_floor; a.up_epi = up_epi;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.w_head = reinterpret_cast<const uint4*>(packed_head);
    a.hd_scale = hd_scale; a.hd_bias = hd_bias; a.hd_floor = hd_floor; a.hd_epi = hd_epi;
    a.logits = logits;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    const int D = 2 * Di, H = 2 * Hi, W = 2 * Wi;
    a.nth = (H + TL_TH - 1) / TL_TH; a.ntw = (W + TL_TW - 1) / TL_TW;
    const int nblocks = (D + TL_P - 1) / TL_P;
    const long tiles = (long)B * a.nth * a.ntw;
    // depth chunks: ONE generation of workgroups where the volume allows it (two resident per CU on 256 CUs = 512 slots): the largest
    // chunk count with tiles * ndc <= 512, at least 2 blocks per chunk (a chunk recomputes one ring plane pair per end); blocks are
    // dealt out evenly (chunk c = blocks [nblocks c / ndc, nblocks (c + 1) / ndc))
    int ndc;
    if (g_tail_nbk > 0) ndc = (nblocks + (int)g_tail_nbk - 1) / (int)g_tail_nbk;
    else {
        ndc = (int)(512 / tiles);
        if (ndc < 1) ndc = 1;
    }
    if (ndc > nblocks / 2) ndc = nblocks / 2 > 0 ? nblocks / 2 : 1;
    const bool fuse = depth != nullptr;
    if (fuse) {
        // the chunk's depth planes sit in LDS (TL_DEPTHS): more chunks if a chunk would be longer
        while ((nblocks + ndc - 1) / ndc * TL_P > TL_DEPTHS) ++ndc;
        if ((long)B * ndc * 4 * (2L * Hi) * (2L * Wi) > workspace_floats) { set_error("pscv_tail_sweep: workspace of %ld floats is too small", workspace_floats); return -1; }
    }
    a.depth = depth; a.depth_bstride = depth_bstride; a.part = fuse ? workspace : nullptr;
    a.nblocks = nblocks;
    a.ndc = ndc;
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    const long nblk = tiles * a.ndc;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_tail_sweep: bad grid %ld", nblk); return -1; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool up_post = (up_epi & PSCV_EPI_RELU_POST) != 0, hd_clamp = (hd_epi & (PSCV_EPI_RELU_PRE | PSCV_EPI_RELU_POST)) != 0;
    const bool plain = !up_post && !hd_clamp;
#define PSCV_TAIL_LAUNCH(HT, P, C, F)                                                                                  \
    {                                                                                                                  \
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(conv3d_tail_kernel<HT, P, C, F>), TL_LDS);         \
        if (e != hipSuccess) { set_error("pscv_tail_sweep: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; } \
        hipLaunchKernelGGL((conv3d_tail_kernel<HT, P, C, F>), dim3((unsigned)nblk), dim3(256), TL_LDS, st, a);         \
    }
#define PSCV_TAIL_DT(HT)                                                                                               \
    if (plain) { if (fuse) PSCV_TAIL_LAUNCH(HT, false, false, true) else PSCV_TAIL_LAUNCH(HT, false, false, false) }   \
    else { if (fuse) PSCV_TAIL_LAUNCH(HT, true, true, true) else PSCV_TAIL_LAUNCH(HT, true, true, false) }
    if (dtype == PSCV_BF16) { PSCV_TAIL_DT(bf16_t) } else { PSCV_TAIL_DT(f16_t) }
#undef PSCV_TAIL_DT
#undef PSCV_TAIL_LAUNCH
    if (fuse) pscv_softargmin_merge_launch(workspace, logits, ndc, B, D, (long)H * W, out_depth, out_conf, st);
    PSCV_CHECK_LAUNCH("pscv_tail_sweep");
    return 0;
}
