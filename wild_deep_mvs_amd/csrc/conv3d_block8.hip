// A whole residual block of the Vis-MVSNet 3-D U-Net in ONE depth sweep (gfx950):
//     t   = act1(s1 * conv3x3x3(x; w1) + b1)            8 -> 8      BasicBlock.conv1 + bn1 + relu   (nn_utils.py:27-31)
//     out = act2(s2 * conv3x3x3(t; w2) + b2 [+ x])      8 -> 8      BasicBlock.conv2 + bn2 (+ identity) + relu   (nn_utils.py:32-37)
// The two layers ran as two launches of the narrow depth sweep (conv3d_sweep.hip): x in, t out, t in + x in (residual), out out
// = 93 B per voxel through the CU's memory path (halo included), which bounds them (3.9 TB/s of algorithmic bytes at
// configuration 5, 54 launches = 20 % of its forward).  Here the intermediate volume never leaves the CU:
//   * output tile 8 x 14 pixels; t is needed on its 1-halo (10 x 16: one 16-pixel MFMA row tile per row), x on the 2-halo (12 x 18);
//   * two LDS plane rings of 8 slots each: x planes (216 voxels x 16 B) and t planes (160 voxels x 16 B, 16-bit like the stored
//     volume was: layer 2 sees the very same values as in the two-launch path -- same bits out);
//   * iteration k of the sweep: layer 1 turns x planes 2k .. 2k+3 into t planes 2k, 2k+1 (plane-pair packed MFMA rows, 9 MFMAs per
//     row tile and plane pair, weights register-resident: as conv3d_sweepc_kernel), layer 2 turns t planes 2k-4 .. 2k-1 into
//     output planes 2k-4, 2k-3 -- the two are independent within an iteration (one barrier per iteration, their MFMA / LDS
//     streams interleave), the residual comes out of the x ring, the next two x planes travel global -> registers -> ring under
//     the MFMAs;
//   * zero padding of layer 2 = t is ZERO outside the volume (not act1(bias)): masked when t is written.
// Bytes per output voxel through the CU path: 16 x 216/112 in + 16 out = 47 (93 before); MFMAs 1.43 + 1.14 row tiles per output
// row (2 before).  8-byte stores like the narrow sweep.
#include "pscv_common.h"
#include <type_traits>

namespace pscv {

typedef __attribute__((ext_vector_type(4))) float b8_f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 b8_f16x8;
template <typename H> struct B8Mfma;
template <> struct B8Mfma<bf16_t> {
    __device__ static __forceinline__ b8_f32x4 run(const uint4& a, const uint4& b, const b8_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_bf16x8, a), __builtin_bit_cast(b8_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct B8Mfma<f16_t> {
    __device__ static __forceinline__ b8_f32x4 run(const uint4& a, const uint4& b, const b8_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(b8_f16x8, a), __builtin_bit_cast(b8_f16x8, b), c, 0, 0, 0);
    }
};

struct Block8Args {
    const uint16_t* in;
    const uint16_t* wpk1;    // PSCV_CONV_S1P8 packing of layer 1: [9 taps][64 lanes][8]
    const uint16_t* wpk2;
    const float *scale1, *bias1, *floor1, *scale2, *bias2, *floor2;
    uint16_t* out;
    int in_cs, in_co, out_cs, out_co;
    int B, D, Hh, W;
    int epi1, epi2, residual;
    int dbg;                 // measurement: 1 = no MFMAs, 2 = no plane fetches in the loop, 4 = no epilogues
    int nth, ntw, ndc, dc;   // tiles along h (8 rows), w (14 columns); depth chunks and planes per chunk (even)
    unsigned mg_th, mg_tw, mg_dc;
};

constexpr int B8_TH = 8, B8_TW = 14;
constexpr int B8_XH = B8_TH + 4, B8_XW = 18;            // x tile: 12 x 18 voxels
constexpr int B8_TROWS = B8_TH + 2, B8_TCOLS = 16;      // t tile: 10 x 16 voxels
constexpr int B8_XV = B8_XH * B8_XW;                    // 216
constexpr int B8_XSLOT = 3584;                          // bytes per x plane slot (216 x 16 = 3456, up to a multiple of 256)
constexpr int B8_TSLOT = B8_TROWS * B8_TCOLS * 16;      // 2560 = 10 x 256
constexpr int B8_NSLOT = 8;
constexpr int B8_PF = 3;                               // x plane pairs in flight (register FIFO)
constexpr int B8_LDS = B8_NSLOT * (B8_XSLOT + B8_TSLOT) + 64;   // (+64: layer 2's discarded columns 14, 15 read two voxels past a row)

template <typename H>
__global__ __launch_bounds__(256, 3) void conv3d_block8_kernel(const Block8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xs = smem;                                   // x ring
    unsigned char* const ts = smem + B8_NSLOT * B8_XSLOT;             // t ring

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * B8_TH, w0 = twi * B8_TW;
    const int dbeg = dci * a.dc, dend = min(a.D, dbeg + a.dc);
    const int npair = (dend - dbeg + 1) >> 1;                         // output plane pairs of this chunk

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    uint4 wf1[9], wf2[9];
    {
        const uint4* w1 = reinterpret_cast<const uint4*>(a.wpk1);
        const uint4* w2 = reinterpret_cast<const uint4*>(a.wpk2);
#pragma unroll
        for (int m = 0; m < 9; ++m) { wf1[m] = w1[m * 64 + lane]; wf2[m] = w2[m * 64 + lane]; }
    }

    // ---- x planes: thread v < 216 owns voxel v of the 12 x 18 tile (raw buffer loads: zeros outside the image / the volume) ----
    const long plane_stride = (long)a.Hh * a.W * a.in_cs;
    const unsigned plane_bytes = (unsigned)(plane_stride * 2 - a.in_co * 2);
    const uint16_t* inb = a.in + (long)b * a.D * plane_stride + a.in_co;
    unsigned goff;
    const bool in_tile = tid < B8_XV;
    {
        const int bh = tid / B8_XW, bw = tid - bh * B8_XW;
        const int gh = h0 - 2 + bh, gw = w0 - 2 + bw;
        const bool gval = in_tile && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
        goff = gval ? (unsigned)(gh * a.W + gw) * (unsigned)a.in_cs * 2u : 0x7ffffff0u;
    }
    auto fetch = [&](int plane) -> uint4 {
        const bool pv = plane >= 0 && plane < a.D;                                                   // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(inb + (long)(pv ? plane : 0) * plane_stride), (short)0, pv ? (int)plane_bytes : 0, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff, 0, 0));
    };
    auto stash = [&](int rel, const uint4& v) {                       // rel = plane - (dbeg - 2)
        if (in_tile) *reinterpret_cast<uint4*>(xs + (rel & (B8_NSLOT - 1)) * B8_XSLOT + tid * 16) = v;
    };

    // ---- per-lane constants ----
    const int c0 = (g & 1) * 4;                                       // this lane's 4 output channels; its plane of a pair is g >> 1
    float sc1[4], bi1[4], fl1[4], sc2[4], bi2[4], fl2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc1[k] = a.scale1 ? a.scale1[c0 + k] : 1.0f; bi1[k] = a.bias1 ? a.bias1[c0 + k] : 0.0f; fl1[k] = a.floor1 ? a.floor1[c0 + k] : 0.0f;
        sc2[k] = a.scale2 ? a.scale2[c0 + k] : 1.0f; bi2[k] = a.bias2 ? a.bias2[c0 + k] : 0.0f; fl2[k] = a.floor2 ? a.floor2[c0 + k] : 0.0f;
    }
    const bool tcol_ok = (unsigned)(w0 - 1 + n) < (unsigned)a.W;
    const bool ocol_ok = n < B8_TW && w0 + n < a.W;
    const long vplane = (long)a.Hh * a.W;

    // ---- prologue: x planes dbeg-2 .. dbeg+1 (rel 0..3) into the ring; the planes of the next B8_PF iterations into a register
    // FIFO (one 16-byte load per thread and plane: a deep FIFO is cheap, and with three workgroups per CU the bytes in flight --
    // not the MFMAs -- set the pace: one iteration of look-ahead left the sweep waiting on HBM latency, 156 us for a block the two
    // separate launches finish in 160) ----
    {
        const uint4 p0 = fetch(dbeg - 2), p1 = fetch(dbeg - 1), p2 = fetch(dbeg), p3 = fetch(dbeg + 1);
        stash(0, p0); stash(1, p1); stash(2, p2); stash(3, p3);
    }
    uint4 nx[B8_PF][2];
#pragma unroll
    for (int s = 0; s < B8_PF; ++s) { nx[s][0] = fetch(dbeg + 2 + 2 * s); nx[s][1] = fetch(dbeg + 3 + 2 * s); }
    __syncthreads();

    // Rows per wave are ADJACENT: the B fragment of input row q serves output rows q, q-1, q-2 (kh = 0, 1, 2), so a wave with 3 (2)
    // output rows reads 5 (4) input rows x 3 column taps instead of 9 per row -- these 8-channel layers need a fresh 1 KB fragment
    // per MFMA and `ds_read_b128` (4 LDS cycles) against `v_mfma_16x16x32` (16 cycles on each of 4 SIMDs) is exactly the LDS peak:
    // the reads, not the MFMAs, bound them.  Layer 1: t rows {0,1,2} {3,4,5} {6,7} {8,9}; layer 2: output rows {2w, 2w+1}.
    const int rb1 = wave < 2 ? 3 * wave : 2 * wave + 2;
    const int rb2 = 2 * wave;
    auto iteration = [&](auto nr1c, int k, uint4 (&nxs)[2]) {
        constexpr int NR1 = decltype(nr1c)::value;
        // the x planes of layer 1's NEXT iteration into slots nobody reads now; their registers are refilled at once
        stash(2 * k + 4, nxs[0]);
        stash(2 * k + 5, nxs[1]);
        if (!(a.dbg & 2)) {
            nxs[0] = fetch(dbeg + 2 * k + 2 + 2 * B8_PF);
            nxs[1] = fetch(dbeg + 2 * k + 3 + 2 * B8_PF);
        }

        // layer 1: t planes (dbeg - 1 + 2k, + 1) from x planes rel 2k .. 2k+3 (lane group g reads plane rel 2k + g);
        // layer 2: output planes (dbeg + 2j, + 1), j = k - 2, from t planes rel 2j .. 2j+3, the residual from the x ring
        const bool l1 = k <= npair && !(a.dbg & 1), l2 = k >= 2 && !(a.dbg & 1);
        const bool e1 = k <= npair && !(a.dbg & 4), e2 = k >= 2 && !(a.dbg & 4);
        const int j = k - 2;
        const unsigned char* sp1 = xs + ((2 * k + g) & (B8_NSLOT - 1)) * B8_XSLOT + (rb1 * B8_XW + n) * 16;
        const unsigned char* sp2 = ts + ((2 * j + g) & (B8_NSLOT - 1)) * B8_TSLOT + (rb2 * B8_TCOLS + n) * 16;
        b8_f32x4 a1[NR1], a2[2];
#pragma unroll
        for (int i = 0; i < NR1; ++i) a1[i] = b8_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) a2[i] = b8_f32x4{0.f, 0.f, 0.f, 0.f};
        if (l1) {
#pragma unroll
            for (int q = 0; q < NR1 + 2; ++q)                        // input row rb1 + q
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const uint4 xf = *reinterpret_cast<const uint4*>(sp1 + (q * B8_XW + kw) * 16);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int i = q - kh;
                        if (i >= 0 && i < NR1) a1[i] = B8Mfma<H>::run(wf1[kh * 3 + kw], xf, a1[i]);
                    }
                }
        }
        if (l2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const uint4 xf = *reinterpret_cast<const uint4*>(sp2 + (q * B8_TCOLS + kw) * 16);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int i = q - kh;
                        if (i >= 0 && i < 2) a2[i] = B8Mfma<H>::run(wf2[kh * 3 + kw], xf, a2[i]);
                    }
                }
        }
        if (e1) {
            unsigned char* tp = ts + ((2 * k + (g >> 1)) & (B8_NSLOT - 1)) * B8_TSLOT + (rb1 * B8_TCOLS + n) * 16 + c0 * 2;
            const int pt = dbeg - 1 + 2 * k + (g >> 1);
            const bool pl_ok = (unsigned)pt < (unsigned)a.D && tcol_ok;
#pragma unroll
            for (int i = 0; i < NR1; ++i) {
                float y[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    y[c] = fmaf(a1[i][c], sc1[c], bi1[c]);
                    if (a.epi1 & PSCV_EPI_RELU_PRE) y[c] = relu_floor(y[c], fl1[c]);
                    if (a.epi1 & PSCV_EPI_RELU_POST) y[c] = relu_floor(y[c], 0.0f);
                }
                const bool ok = pl_ok && (unsigned)(h0 - 1 + rb1 + i) < (unsigned)a.Hh;
                *reinterpret_cast<uint2*>(tp + i * (B8_TCOLS * 16)) =
                    ok ? make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3])) : make_uint2(0u, 0u);
            }
        }
        if (e2) {
            const unsigned char* rp = xs + ((2 * j + 2 + (g >> 1)) & (B8_NSLOT - 1)) * B8_XSLOT + ((rb2 + 2) * B8_XW + 2 + n) * 16 + c0 * 2;
            const int od = dbeg + 2 * j + (g >> 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float y[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    y[c] = fmaf(a2[i][c], sc2[c], bi2[c]);
                    if (a.epi2 & PSCV_EPI_RELU_PRE) y[c] = relu_floor(y[c], fl2[c]);
                }
                if (a.residual) {
                    const uint2 sv = *reinterpret_cast<const uint2*>(rp + i * (B8_XW * 16));
                    y[0] += Half16<H>::lo(sv.x); y[1] += Half16<H>::hi(sv.x); y[2] += Half16<H>::lo(sv.y); y[3] += Half16<H>::hi(sv.y);
                }
                if (a.epi2 & PSCV_EPI_RELU_POST) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[c] = relu_floor(y[c], 0.0f);
                }
                if (od < dend && ocol_ok && h0 + rb2 + i < a.Hh) {
                    const long vox = (((long)b * a.D + od) * a.Hh + (h0 + rb2 + i)) * a.W + w0 + n;
                    *reinterpret_cast<uint2*>(a.out + vox * a.out_cs + a.out_co + c0) =
                        make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                }
            }
        }
        __syncthreads();
    };
    for (int k0 = 0; k0 <= npair + 1; k0 += B8_PF) {
#pragma unroll
        for (int s = 0; s < B8_PF; ++s) {
            const int k = k0 + s;
            if (k <= npair + 1) {                                     // workgroup-uniform
                if (wave < 2) iteration(std::integral_constant<int, 3>{}, k, nx[s]);
                else iteration(std::integral_constant<int, 2>{}, k, nx[s]);
            }
        }
    }
}

}  // namespace pscv

pscv::Knob g_block8_slots = {0, pscv::KNOB_SPARE3};   // pscv_set_tuning("block8_slots", n): workgroup slots the depth chunks are sized for

extern "C" int pscv_conv3d_block8(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed1, const float* scale1,
                                  const float* bias1, const float* floor1, int epi1, const uint16_t* packed2, const float* scale2,
                                  const float* bias2, const float* floor2, int epi2, int residual, void* out, int out_cstride, int out_coff,
                                  int B, int D, int H, int W, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(in && packed1 && packed2 && out, "pscv_conv3d_block8: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pscv_conv3d_block8: bad sizes");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_conv3d_block8: storage dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(in_cstride % 8 == 0 && in_coff % 8 == 0 && in_coff + 8 <= in_cstride, "pscv_conv3d_block8: input channel slice must be 8-aligned");
    PSCV_CHECK_ARG(out_cstride % 4 == 0 && out_coff % 4 == 0 && out_coff + 8 <= out_cstride, "pscv_conv3d_block8: bad output channel slice");
    PSCV_CHECK_ARG((long)H * W * in_cstride * 2 < 0x7fffffffL, "pscv_conv3d_block8: an input plane of %d x %d x %d channels exceeds 2 GiB", H, W, in_cstride);
    PSCV_CHECK_ARG(in != out, "pscv_conv3d_block8: in-place operation is not supported (halo voxels are re-read by neighbouring tiles)");
    Block8Args a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk1 = packed1; a.wpk2 = packed2;
    a.scale1 = scale1; a.bias1 = bias1; a.floor1 = floor1; a.scale2 = scale2; a.bias2 = bias2; a.floor2 = floor2;
    a.out = reinterpret_cast<uint16_t*>(out);
    a.in_cs = in_cstride; a.in_co = in_coff; a.out_cs = out_cstride; a.out_co = out_coff;
    a.B = B; a.D = D; a.Hh = H; a.W = W; a.epi1 = epi1; a.epi2 = epi2; a.residual = residual;
    a.nth = (H + B8_TH - 1) / B8_TH;
    a.ntw = (W + B8_TW - 1) / B8_TW;
    // one resident round of workgroups (3 per CU); each depth-chunk seam recomputes three x planes and one t plane
    const long tiles = (long)B * a.nth * a.ntw;
    const int knob = g_block8_slots;
    a.dbg = knob >> 16;
    const long slots = (knob & 0xffff) > 0 ? (long)(knob & 0xffff) : 768;
    const long ndc_want = tiles >= slots ? 1 : slots / tiles;
    int dc = (int)((D + ndc_want - 1) / ndc_want);
    dc = (dc + 1) & ~1;
    dc = dc < 8 ? 8 : dc;
    dc = dc > D ? ((D + 1) & ~1) : dc;
    a.dc = dc;
    a.ndc = (D + dc - 1) / dc;
    const long nblk = tiles * a.ndc;
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d_block8: bad grid %ld", nblk); return -1; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int di = dtype == PSCV_BF16 ? 0 : 1;
    const void* kern = di == 0 ? reinterpret_cast<const void*>(conv3d_block8_kernel<bf16_t>) : reinterpret_cast<const void*>(conv3d_block8_kernel<f16_t>);
    {
        hipError_t e = ensure_dyn_lds(kern, B8_LDS);
        if (e != hipSuccess) { set_error("pscv_conv3d_block8: hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    if (di == 0) hipLaunchKernelGGL(conv3d_block8_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), B8_LDS, st, a);
    else hipLaunchKernelGGL(conv3d_block8_kernel<f16_t>, dim3((unsigned)nblk), dim3(256), B8_LDS, st, a);
    PSCV_CHECK_LAUNCH("pscv_conv3d_block8");
    return 0;
}
