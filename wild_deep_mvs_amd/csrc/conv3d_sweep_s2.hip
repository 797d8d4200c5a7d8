// Depth-sweep 3x3x3 STRIDE-2 convolution for 8 input channels (MVSNet CostRegNet.conv1 8 -> 16 at full resolution; the Vis
// U-Net's strided BasicBlock conv fused with its 1x1x1 shortcut, 8 -> 32).  gfx950, wave64.
//
// Why a sweep: the brick kernel (conv3d.hip) gives a workgroup 2 x 2 x 16 output voxels -- 7 MFMAs per wave behind a workgroup
// decode, a 5 x 5 x 33 brick (2.5 input planes per output plane, 1.6x halo in-plane), two barriers and an LDS reduction; at the
// headline size the layer took 30 us for 79 MB (the workgroup lifetime of ~7 K cycles is all fixed cost).  Here a workgroup owns
// 8 x 16 output pixels and walks the output planes of a depth chunk:
//   * input planes live in a 5-slot LDS ring (17 x 33 voxels of 16 B, columns split by parity so that the 16 lanes of an MFMA
//     column read consecutive 16-byte slots for every kw); output plane o reads planes 2o-1, 2o, 2o+1 while the two planes of
//     o+1 are written into the other two slots -- every input plane is fetched once per sweep (in-plane halo 1.13x);
//   * the next two planes are requested one iteration ahead into registers (issue early, write late);
//   * k = (tap 4 s + g, 8 channels): the standard dense packing of pscv_pack_conv3d_weights (kind S2) -- 7 A fragments per
//     16-channel N-tile, resident in 28 (56) VGPRs for the whole sweep; a lane's tap decides its plane slot and in-plane offset.
// Selected by pscv_conv3d for kind S2, c_in = 8, c_out <= 32 on volumes with at least S2S_MIN_VOXELS output voxels
// (pscv_set_tuning("conv_s2_sweep", 0) keeps the brick kernel).
//
// Replaces (fdarmon/wild_deep_mvs): ConvBnReLU3D(8, 16, stride=2) models/MVSNet/model.py:49,76 (block models/MVSNet/module.py:41-48);
// BasicBlock(8, 16, stride 2).conv1 + downsample of models/VisMVSNet/nn_utils.py:27-37, 215-226.
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 s2_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 s2_f16x8;
typedef __attribute__((ext_vector_type(4))) float s2_f32x4;

template <typename H> struct S2Mfma;
template <> struct S2Mfma<bf16_t> {
    __device__ static __forceinline__ s2_f32x4 run(const uint4& a, const uint4& b, const s2_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s2_bf16x8, a), __builtin_bit_cast(s2_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct S2Mfma<f16_t> {
    __device__ static __forceinline__ s2_f32x4 run(const uint4& a, const uint4& b, const s2_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s2_f16x8, a), __builtin_bit_cast(s2_f16x8, b), c, 0, 0, 0);
    }
};

Knob g_conv_s2_sweep = {1, KNOB_CONV_S2_SWEEP};   // pscv_set_tuning("conv_s2_sweep", 0): always the brick kernel; 2: the sweep at any size
Knob g_s2s_slots = {0, KNOB_S2S_SLOTS};       // pscv_set_tuning("s2s_slots", n): resident-workgroup target that sizes the depth chunks (0 = 768)

struct S2sArgs {
    const uint16_t* in;
    const uint4* wpk;        // [7 steps][nt_total][64 lanes] x 8 halves: the dense S2 packing
    const float* scale;
    const float* bias;
    const float* floor;
    const uint16_t* skip;
    void* out;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int cout, epi;
    int nth, ntw, ndc, dc;   // output tiles along h, w; depth chunks and output planes per chunk
    unsigned mg_th, mg_tw, mg_dc;
};

constexpr int S2S_TH = 8, S2S_R = 2;                 // output rows per workgroup / per wave
constexpr int S2S_BH = 2 * S2S_TH + 1, S2S_BW = 33;  // input rows / columns of a plane tile
constexpr int S2S_PITCH = 34;                        // 17 even + 17 (16 used) odd column slots per row
constexpr int S2S_PB = S2S_BH * S2S_PITCH * 16;      // bytes per plane slot (9248)
constexpr int S2S_NSLOT = 5;
constexpr int S2S_NLD = (S2S_BH * S2S_BW + 255) / 256;   // 16-byte chunks per thread per plane (3)
constexpr int S2S_STEPS = 7;                         // ceil(27 taps x 8 ch / 32)
constexpr long S2S_MIN_VOXELS = 65536;
PSCV_PROF_BUFFER(s2s)

// PLAIN: no skip tensor and 16-bit output (every call of the models): the epilogue is straight-line code without the two branches
template <typename H, int NT, bool PLAIN>
__global__ __launch_bounds__(256) void conv3d_sweep_s2_kernel(const S2sArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int oh0 = thi * S2S_TH, ow0 = twi * 16;
    const int obeg = dci * a.dc, oend = min(a.Do, obeg + a.dc);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    PSCV_PROF_BEGIN

    // ---- A fragments of the whole layer ----
    uint4 wf[S2S_STEPS][NT];
#pragma unroll
    for (int s = 0; s < S2S_STEPS; ++s)
#pragma unroll
        for (int m = 0; m < NT; ++m) wf[s][m] = a.wpk[(s * NT + m) * 64 + lane];

    // ---- B operand: lane group g of step s holds tap 4 s + g (taps >= 27 carry zero weights: any staged voxel will do) ----
    int kd_s[S2S_STEPS], boff[S2S_STEPS][S2S_R];
#pragma unroll
    for (int s = 0; s < S2S_STEPS; ++s) {
        int tap = 4 * s + g;
        tap = tap > 26 ? 26 : tap;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        kd_s[s] = kd;
        const int cpos = kw == 1 ? 17 + n : n + (kw >> 1);          // input column 2 n + kw in the parity-split row
#pragma unroll
        for (int r = 0; r < S2S_R; ++r) boff[s][r] = ((2 * (wave * S2S_R + r) + kh) * S2S_PITCH + cpos) * 16;
    }

    // ---- staging descriptors: thread t owns chunks t, t + 256, t + 512 of every plane ----
    // Loads are raw buffer loads: the descriptor covers exactly one input plane, a chunk outside the image carries an
    // out-of-range offset and a plane outside the volume an empty descriptor -- the hardware returns zeros (= the conv's padding),
    // so a plane costs three load instructions and no predicate, zero fill or 64-bit address arithmetic per chunk.
    unsigned goff[S2S_NLD];
    int loff[S2S_NLD];
    bool lval[S2S_NLD];
    const long plane_stride = (long)a.Hi * a.Wi * a.in_cs;
    const unsigned plane_bytes = (unsigned)(plane_stride * 2 - a.in_co * 2);
    const uint16_t* inb = a.in + (long)b * a.Di * plane_stride + a.in_co;
#pragma unroll
    for (int i = 0; i < S2S_NLD; ++i) {
        const int id = tid + 256 * i;
        const int bh = id / S2S_BW, bw = id - bh * S2S_BW;
        const int gh = 2 * oh0 - 1 + bh, gw = 2 * ow0 - 1 + bw;
        lval[i] = id < S2S_BH * S2S_BW;
        const bool gval = lval[i] && (unsigned)gh < (unsigned)a.Hi && (unsigned)gw < (unsigned)a.Wi;
        goff[i] = gval ? (unsigned)(gh * a.Wi + gw) * (unsigned)(a.in_cs * 2) : 0x7ffffff0u;      // bytes; invalid -> out of range
        loff[i] = (bh * S2S_PITCH + (bw & 1) * 17 + (bw >> 1)) * 16;
    }
    const int plane_hi = min(a.Di - 1, 2 * oend - 1);   // last input plane this sweep reads
    auto fetch = [&](int plane, uint4 (&reg)[S2S_NLD]) {
        const bool pv = plane >= 0 && plane <= plane_hi;                                         // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(inb + (long)(pv ? plane : 0) * plane_stride), (short)0, pv ? (int)plane_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < S2S_NLD; ++i)
            reg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[i], 0, 0));
    };
    auto stash = [&](int ring, const uint4 (&reg)[S2S_NLD]) {
        unsigned char* sp = smem + ring * S2S_PB;
#pragma unroll
        for (int i = 0; i < S2S_NLD; ++i)
            if (i + 1 < S2S_NLD || lval[i]) *reinterpret_cast<uint4*>(sp + loff[i]) = reg[i];     // (only the last chunk is ragged)
    };

    // ---- epilogue constants: lane (n, g) owns channels m * 16 + g * 4 .. +3 of output pixel n of each of its rows ----
    float sc[NT][4], bi[NT][4], fl[NT][4];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = m * 16 + g * 4 + k;
            const bool cv = c < a.cout;
            sc[m][k] = (a.scale && cv) ? a.scale[c] : 1.0f;
            bi[m][k] = (a.bias && cv) ? a.bias[c] : 0.0f;
            fl[m][k] = (a.epi & PSCV_EPI_RELU_PRE) ? ((a.floor && cv) ? a.floor[c] : 0.0f) : -__builtin_inff();   // ReLU switches as clamps
        }
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    const long oplane = (long)a.Ho * a.Wo * a.out_cs, splane = (long)a.Ho * a.Wo * a.skip_cs;
    unsigned lane_out[S2S_R], lane_skip[S2S_R];
    bool lane_ok[S2S_R];
#pragma unroll
    for (int r = 0; r < S2S_R; ++r) {
        const int oh = oh0 + wave * S2S_R + r, ow = ow0 + n;
        lane_ok[r] = oh < a.Ho && ow < a.Wo;
        lane_out[r] = (unsigned)(oh * a.Wo + ow) * (unsigned)a.out_cs + (unsigned)(g * 4);
        lane_skip[r] = (unsigned)(oh * a.Wo + ow) * (unsigned)a.skip_cs + (unsigned)(g * 4);
    }

    // ---- prologue: planes 2 obeg - 1 .. 2 obeg + 1 into slots 0..2; the two planes of the next output plane into registers ----
    uint4 na[S2S_NLD], nb[S2S_NLD];
    {
        fetch(2 * obeg - 1, na); fetch(2 * obeg, nb);
        stash(0, na); stash(1, nb);
        fetch(2 * obeg + 1, na);
        stash(2, na);
    }
    fetch(2 * obeg + 2, na);
    fetch(2 * obeg + 3, nb);
    __syncthreads();
    PSCV_STAMP(0)

    int ring = 0;   // slot of plane 2 o - 1
    for (int o = obeg; o < oend; ++o) {
        // planes 2 o + 2, 2 o + 3 (requested one iteration ago) go to the two slots that are not read now; their registers are
        // refilled at once with the planes of iteration o + 1
        int s3 = ring + 3, s4 = ring + 4;
        s3 = s3 >= S2S_NSLOT ? s3 - S2S_NSLOT : s3;
        s4 = s4 >= S2S_NSLOT ? s4 - S2S_NSLOT : s4;
        stash(s3, na);
        stash(s4, nb);
        fetch(2 * o + 4, na);
        fetch(2 * o + 5, nb);
        PSCV_STAMP(1)

        int sb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int sl = ring + k;
            sl = sl >= S2S_NSLOT ? sl - S2S_NSLOT : sl;
            sb[k] = sl * S2S_PB;
        }
        s2_f32x4 acc[S2S_R][NT];
#pragma unroll
        for (int r = 0; r < S2S_R; ++r)
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[r][m] = s2_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S2S_STEPS; ++s) {
            const int base = kd_s[s] == 0 ? sb[0] : (kd_s[s] == 1 ? sb[1] : sb[2]);
#pragma unroll
            for (int r = 0; r < S2S_R; ++r) {
                const uint4 xf = *reinterpret_cast<const uint4*>(smem + base + boff[s][r]);
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[r][m] = S2Mfma<H>::run(wf[s][m], xf, acc[r][m]);
            }
        }
        PSCV_STAMP(2)

        // epilogue: the base of output plane o is scalar arithmetic; a lane adds its precomputed row / column / channel offset
        {
            fp16_ovfl_mode(true);     // saturating 16-bit stores; off again before the next plane's MFMAs (pscv_common.h)
            const long obase = ((long)b * a.Do + o) * oplane + a.out_co, sbase = ((long)b * a.Do + o) * splane + a.skip_co;
#pragma unroll
            for (int r = 0; r < S2S_R; ++r)
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    if (lane_ok[r] && m * 16 + g * 4 < a.cout) {
                        float y[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) y[k] = clamp_lo(fmaf(acc[r][m][k], sc[m][k], bi[m][k]), fl[m][k]);
                        if constexpr (PLAIN) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) y[k] = clamp_lo(y[k], lo_post);
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + obase + lane_out[r] + m * 16) =
                                make_uint2(Half16<H>::pack_ovfl(y[0], y[1]), Half16<H>::pack_ovfl(y[2], y[3]));
                            continue;
                        }
                        uint2 sv = make_uint2(0u, 0u);
                        if (a.skip) sv = *reinterpret_cast<const uint2*>(a.skip + sbase + lane_skip[r] + m * 16);
                        y[0] = clamp_lo(y[0] + Half16<H>::lo(sv.x), lo_post); y[1] = clamp_lo(y[1] + Half16<H>::hi(sv.x), lo_post);
                        y[2] = clamp_lo(y[2] + Half16<H>::lo(sv.y), lo_post); y[3] = clamp_lo(y[3] + Half16<H>::hi(sv.y), lo_post);
                        if (a.out_f32)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + obase + lane_out[r] + m * 16) = make_float4(y[0], y[1], y[2], y[3]);
                        else
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + obase + lane_out[r] + m * 16) =
                                make_uint2(Half16<H>::pack_ovfl(y[0], y[1]), Half16<H>::pack_ovfl(y[2], y[3]));
                    }
                }
            fp16_ovfl_mode(false);
        }
        PSCV_STAMP(3)
        ring += 2;
        ring = ring >= S2S_NSLOT ? ring - S2S_NSLOT : ring;
        __syncthreads();
        PSCV_STAMP(4)
    }
    PSCV_PROF_END(s2s, blockIdx.x)
}

template <typename H, int NT>
static int s2s_launch(const S2sArgs& a, long nblk, hipStream_t st) {
    if (!a.skip && !a.out_f32) hipLaunchKernelGGL((conv3d_sweep_s2_kernel<H, NT, true>), dim3((unsigned)nblk), dim3(256), S2S_NSLOT * S2S_PB, st, a);
    else hipLaunchKernelGGL((conv3d_sweep_s2_kernel<H, NT, false>), dim3((unsigned)nblk), dim3(256), S2S_NSLOT * S2S_PB, st, a);
    return 0;
}

}  // namespace pscv

PSCV_PROF_EXPORT(s2s)

// returns 1 when the layer is not one this kernel takes (the caller then uses the brick kernel), 0 on a launch, < 0 on error
int pscv_conv3d_sweep_s2_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                                const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                                int out_cstride, int out_coff, int out_dtype, int B, int Di, int Hi, int Wi, int c_in, int c_out,
                                int epi_flags, hipStream_t st) {
    using namespace pscv;
    const int Do = (Di + 1) / 2, Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
    if (!g_conv_s2_sweep || c_in != 8 || c_out > 32 || c_out % 4) return 1;
    if (g_conv_s2_sweep != 2 && (long)B * Do * Ho * Wo < S2S_MIN_VOXELS) return 1;      // (2: any size -- tests)
    // 32-bit in-plane element offsets on top of 64-bit plane bases
    if ((long)Hi * Wi * in_cstride * 2 >= 0x7fffffffL || (long)Ho * Wo * (out_cstride > skip_cstride ? out_cstride : skip_cstride) >= 0x7fffffffL) return 1;
    S2sArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = reinterpret_cast<const uint4*>(packed);
    a.scale = scale; a.bias = bias; a.floor = floor;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.out_cs = out_cstride; a.out_co = out_coff; a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.cout = c_out; a.epi = epi_flags;
    a.nth = (Ho + S2S_TH - 1) / S2S_TH;
    a.ntw = (Wo + 15) / 16;
    // depth chunks: about one resident round of workgroups (3 per CU), at least 4 output planes per sweep
    const long tiles = (long)B * a.nth * a.ntw;
    const long slots = g_s2s_slots > 0 ? g_s2s_slots : 768;
    const long ndc_want = tiles >= slots ? 1 : slots / tiles;
    int dc = (int)((Do + ndc_want - 1) / ndc_want);
    dc = dc < 4 ? 4 : dc;
    dc = dc > Do ? Do : dc;
    a.dc = dc;
    a.ndc = (Do + dc - 1) / dc;
    const long nblk = tiles * a.ndc;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(s2 sweep): bad grid %ld", nblk); return -1; }
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    const int nt = (c_out + 15) / 16;
    if (dtype == PSCV_BF16) return nt == 1 ? s2s_launch<bf16_t, 1>(a, nblk, st) : s2s_launch<bf16_t, 2>(a, nblk, st);
    return nt == 1 ? s2s_launch<f16_t, 1>(a, nblk, st) : s2s_launch<f16_t, 2>(a, nblk, st);
}
