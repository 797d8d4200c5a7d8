// 2-D convolution (k3 s1, k5 s2, k3 s2, k1 s1/s2, and the 2x2 parity sub-convolutions of a k3 s2 transposed convolution)
// over channels-last 16-bit feature maps as an MFMA implicit GEMM.  gfx950.
//
// The step BEFORE the plane-sweep path: MVSNet's FeatureNet is eight small-channel layers
// (3->8->8->16->16->16->32->32->32) on every view; at these widths a library convolution spends ~1 ms on 35 MB of
// traffic per view.  Same GEMM view as conv3d.hip: D[m][n] = sum_k W[m][k] X[k][n] with m = output channel,
// n = 16 x-adjacent output pixels, k = (tap, c_in), `v_mfma_f32_16x16x32_{f16,bf16}`; the input halo region of the
// workgroup's output tile sits in LDS, every input pixel is fetched once per tile, and the epilogue (folded BatchNorm
// affine, ReLU) ends in one 8-byte store per lane.  The last layer writes the [B,h,w,32] 16-bit map the warp kernel
// reads, so no layout conversion remains between the two stages.
//
// Stride 2: output pixel x reads input columns 2x + kw, which would put the 16 lanes of an operand read 2 voxels
// apart (4-way LDS bank conflicts for every voxel size that keeps 16-byte alignment).  The brick is therefore staged
// with its columns split by parity -- [even columns | odd columns] per row -- so that for a fixed tap the lanes read
// consecutive voxels exactly like the stride-1 case.
//
// Vis-MVSNet's FeatExt (a 2-D residual U-Net, models/VisMVSNet/model_cas.py:18-35 over nn_utils.py:123-278) adds: a
// residual input added BEFORE the activation (BasicBlock: relu(bn(conv) + shortcut)), 1x1 (strided) shortcut convs, k3 s2
// convs, up to 128 channels (the output tiles of a wide layer are split over blockIdx.y), writes into a channel slice of a
// wider map (the decoder's cat) and ConvTranspose2d(k3, s2, p1, op1) as four 2x2-tap parity sub-convolutions whose
// outputs interleave (no zero insertion): out[2i+a] takes input i with kernel index a+1 and, for a = 1, input i+1 with
// kernel index 0.
//
// Replaces (fdarmon/wild_deep_mvs): ConvBnReLU (models/MVSNet/module.py:23-38) inside FeatureNet
// (models/MVSNet/model.py:21-41) and its final plain Conv2d; conv + LeakyReLU(0.1) of CVP-MVSNet's FeaturePyramid
// (models/CVP_MVSNet/models/modules.py:24-28, net.py:21-47); every layer of Vis-MVSNet's FeatExt.
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 c2_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 c2_f16x8;
typedef __attribute__((ext_vector_type(4))) float c2_f32x4;

template <typename H> struct C2Mfma;
template <> struct C2Mfma<bf16_t> {
    __device__ static __forceinline__ c2_f32x4 run(const uint4& a, const uint4& b, const c2_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c2_bf16x8, a), __builtin_bit_cast(c2_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct C2Mfma<f16_t> {
    __device__ static __forceinline__ c2_f32x4 run(const uint4& a, const uint4& b, const c2_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c2_f16x8, a), __builtin_bit_cast(c2_f16x8, b), c, 0, 0, 0);
    }
};

struct Conv2dArgs {
    const uint16_t* in;      // [B,Hi,Wi,c_in]
    const uint4* wpk;        // [steps][NT][64 lanes] x 8 halves
    const float* scale;      // [c_out] or null
    const float* bias;
    void* out;               // [B,Hof,Wof,out_cs], written at channel offset out_co, pixel (oy*oys + oyo, ox*oxs + oxo)
    const uint16_t* skip;    // null or [B,Hof,Wof,skip_cs] read at skip_co: added before the activation
    int out_f32;
    int B, Hi, Wi, Ho, Wo, cout;
    int out_cs, out_co, skip_cs, skip_co;
    int oys, oxs, oyo, oxo, Hof, Wof;
    int nt_total;            // 16-channel output tiles of the layer (blockIdx.y picks this block's first NT tiles)
    float neg_slope;         // activation y -> max(y, neg_slope * y): 0 = ReLU, 0.1 = LeakyReLU(0.1), 1 = none
    int nth, ntw;
    unsigned mg_th, mg_tw;
};

__host__ __device__ constexpr int c2_ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int c2_vs(int cin) { return cin == 32 ? 96 : cin * 2 + 16; }   // LDS bytes per pixel (conv3d.hip)
constexpr int C2_TW = 32;    // output columns per workgroup (two 16-pixel MFMA column tiles)

__host__ __device__ constexpr int c2_pad(int ks) { return ks == 2 ? 0 : ks / 2; }   // the 2x2 parity sub-convs read (i, i+1)

template <typename H, int CIN, int NT, int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv2d_kernel(const Conv2dArgs a) {
    constexpr int TH = STRIDE == 1 ? 8 : 4;                 // output rows per workgroup
    constexpr int BH = STRIDE * (TH - 1) + KS, BW = STRIDE * (C2_TW - 1) + KS;   // input brick
    constexpr int HALFW = (BW + 1) / 2;
    constexpr int BWL = STRIDE == 1 ? BW : 2 * HALFW;       // LDS row length in pixels (parity split for stride 2)
    constexpr int VS = c2_vs(CIN), CCH = CIN / 8;
    constexpr int NTAPS = KS * KS, NSTEPS = c2_ceil_div(NTAPS * CIN, 32);
    constexpr int NMT = TH * 2, MB = NMT / 4;               // M-tiles per workgroup / per wave
    constexpr int PAD = c2_pad(KS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
    const int tw_i = fast_divmod(wg, a.ntw, a.mg_tw);
    const int th_i = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int oy0 = th_i * TH, ox0 = tw_i * C2_TW;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y * NT;

    // A fragments: the first PF k-steps (all of them for the small layers) are requested before the brick so that both
    // share one memory latency; wide layers (64 channels: 72 fragments) continue through a ring, step s + PF being
    // requested as soon as step s has been consumed (indices are compile-time: the k-loop is unrolled)
    constexpr int PF = NSTEPS * NT <= 28 ? NSTEPS : (24 / NT);
    uint4 wf[PF][NT];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int m = 0; m < NT; ++m) wf[s][m] = a.wpk[(s * a.nt_total + nt0 + m) * 64 + lane];

    {   // ---- stage the brick: a wave takes whole rows (row arithmetic is scalar), a lane its chunk(s) of the row ----
        constexpr int ROWCH = BW * CCH, NPASS = c2_ceil_div(ROWCH, 64);
        // rows per load batch: every batch exposes one global-memory latency, so the wide layers (whose k-loop needs few other
        // registers at this point) take all of a wave's rows in ONE batch (64 channels: 3 rows x 5 chunks = 60 VGPRs in flight)
        constexpr int RB_CAP = CIN >= 64 ? 16 : 8;
        constexpr int RB = (RB_CAP / NPASS) < 1 ? 1 : RB_CAP / NPASS;
        const unsigned row_bytes = (unsigned)a.Wi * CIN * 2u;
        const char* inb = reinterpret_cast<const char*>(a.in) + (unsigned long)b * a.Hi * row_bytes;
        unsigned goff[NPASS];
        int loff[NPASS];
        bool cok[NPASS], cin_row[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int qc = lane + 64 * k;
            const int bw = qc / CCH, cc = qc - bw * CCH;
            const int gx = ix0 + bw;
            cin_row[k] = qc < ROWCH;
            cok[k] = cin_row[k] && (unsigned)gx < (unsigned)a.Wi;
            goff[k] = cok[k] ? (unsigned)(gx * CIN + cc * 8) * 2u : 0u;
            const int lcol = STRIDE == 1 ? bw : (bw & 1) * HALFW + (bw >> 1);
            loff[k] = lcol * VS + cc * 16;
        }
        for (int r0 = wave; r0 < BH; r0 += 4 * RB) {
            uint4 val[RB][NPASS];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = r0 + 4 * j;                       // wave-uniform
                const int gy = iy0 + r;
                const bool rok = r < BH && (unsigned)gy < (unsigned)a.Hi;
                const char* rp = inb + (unsigned long)(rok ? gy : 0) * row_bytes;
#pragma unroll
                for (int k = 0; k < NPASS; ++k) {
                    val[j][k] = make_uint4(0u, 0u, 0u, 0u);
                    if (rok && cok[k]) val[j][k] = *reinterpret_cast<const uint4*>(rp + goff[k]);
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = r0 + 4 * j;
                if (r < BH) {
#pragma unroll
                    for (int k = 0; k < NPASS; ++k)
                        if (cin_row[k]) *reinterpret_cast<uint4*>(smem + r * (BWL * VS) + loff[k]) = val[j][k];
                }
            }
        }
    }
    // epilogue constants of this lane's 4 output channels per tile
    float sc[NT][4], bi[NT][4];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (nt0 + m) * 16 + g * 4 + k;
            sc[m][k] = (a.scale && c < a.cout) ? a.scale[c] : 1.0f;
            bi[m][k] = (a.bias && c < a.cout) ? a.bias[c] : 0.0f;
        }
    __syncthreads();

    c2_f32x4 acc[MB][NT];
    int anchor[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mt = wave * MB + i;
        const int row = mt >> 1, ct = mt & 1;
        anchor[i] = ((row * STRIDE) * BWL + ct * 16 + n) * VS;
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[i][m] = c2_f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        const int kk0 = s * 32 + g * 8;                     // lane-group dependent, folded per g by the compiler's select
        int tap = kk0 / CIN;
        const int c0 = kk0 - tap * CIN;
        tap = tap > NTAPS - 1 ? NTAPS - 1 : tap;            // k padding: any in-brick pixel (its weights are zero)
        const int kh = tap / KS, kw = tap - kh * KS;
        const int kcol = STRIDE == 1 ? kw : (kw & 1) * HALFW + (kw >> 1);
        const int koff = (kh * BWL + kcol) * VS + c0 * 2;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const uint4 xf = *reinterpret_cast<const uint4*>(smem + anchor[i] + koff);
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[i][m] = C2Mfma<H>::run(wf[s % PF][m], xf, acc[i][m]);
        }
        if (s + PF < NSTEPS) {
#pragma unroll
            for (int m = 0; m < NT; ++m) wf[s % PF][m] = a.wpk[((s + PF) * a.nt_total + nt0 + m) * 64 + lane];
        }
    }

    // ---- epilogue: lane (n, g) owns channels 16 m + 4 g .. + 3 of pixel n ----
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mt = wave * MB + i;
        const int oy = oy0 + (mt >> 1), ox = ox0 + (mt & 1) * 16 + n;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const unsigned long pix = ((unsigned long)b * a.Hof + (oy * a.oys + a.oyo)) * a.Wof + (ox * a.oxs + a.oxo);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const int c0 = (nt0 + m) * 16 + g * 4;
            if (c0 >= a.cout) continue;
            float v[4], y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaf(acc[i][m][k], sc[m][k], bi[m][k]);
            if (a.skip) {
                const uint2 sv = *reinterpret_cast<const uint2*>(a.skip + pix * a.skip_cs + a.skip_co + c0);
                v[0] += Half16<H>::lo(sv.x); v[1] += Half16<H>::hi(sv.x); v[2] += Half16<H>::lo(sv.y); v[3] += Half16<H>::hi(sv.y);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = fmaxf(v[k], v[k] * a.neg_slope);
            if (a.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + pix * a.out_cs + a.out_co + c0) = make_float4(y[0], y[1], y[2], y[3]);
            else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + pix * a.out_cs + a.out_co + c0) =
                     make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
        }
    }
}

template <typename H, int CIN, int NT, int KS, int STRIDE>
static int c2_launch(Conv2dArgs& a, int n_split, hipStream_t st) {
    constexpr int TH = STRIDE == 1 ? 8 : 4;
    constexpr int BH = STRIDE * (TH - 1) + KS, BW = STRIDE * (C2_TW - 1) + KS;
    constexpr int BWL = STRIDE == 1 ? BW : 2 * ((BW + 1) / 2);
    constexpr int LDS = BH * BWL * c2_vs(CIN) + 64;
    static_assert(LDS <= 160 * 1024, "brick does not fit the LDS");
    a.nth = c2_ceil_div(a.Ho, TH); a.ntw = c2_ceil_div(a.Wo, C2_TW);
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw);
    const long nblk = (long)a.B * a.nth * a.ntw;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv2d: bad grid %ld", nblk); return -1; }
    auto kern = conv2d_kernel<H, CIN, NT, KS, STRIDE>;
    if (LDS > 60000) {
        static bool done = false;   // per instantiation
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) { set_error("pscv_conv2d: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_split), dim3(256), LDS, st, a);
    return 0;
}

template <typename H>
static int c2_dispatch(Conv2dArgs& a, int c_in, int c_out, int ks, int stride, hipStream_t st) {
    const int nt = (c_out + 15) / 16;
    a.nt_total = nt;
    // (c_in, output tiles per workgroup, kernel size, stride); layers wider than 4 tiles (64 channels) split their output
    // tiles over blockIdx.y in groups of 4
    const int ntw = nt > 4 ? 4 : nt;
    const int split = (nt + ntw - 1) / ntw;
    if (nt > 4 && nt % 4) { set_error("pscv_conv2d: c_out=%d above 64 must be a multiple of 64", c_out); return -1; }
#define PSCV_C2(CI, NTV, K, S) if (c_in == CI && ntw == NTV && ks == K && stride == S) return c2_launch<H, CI, NTV, K, S>(a, split, st);
    PSCV_C2(8, 1, 3, 1) PSCV_C2(16, 1, 3, 1) PSCV_C2(32, 2, 3, 1) PSCV_C2(16, 2, 3, 1) PSCV_C2(32, 1, 3, 1)
    PSCV_C2(8, 4, 3, 1) PSCV_C2(64, 4, 3, 1) PSCV_C2(64, 2, 3, 1)                       // CVP pyramid: 3->64, 64->64, 64->32
    PSCV_C2(8, 1, 5, 2) PSCV_C2(16, 2, 5, 2) PSCV_C2(8, 2, 5, 2) PSCV_C2(16, 1, 5, 2)
    // Vis FeatExt: 1x1 shortcuts, k3 s2 down-sampling convs, 128-channel layers, 2x2 parity sub-convs of the deconvs
    PSCV_C2(16, 2, 1, 1) PSCV_C2(32, 4, 1, 2) PSCV_C2(64, 4, 1, 2)
    PSCV_C2(32, 4, 3, 2) PSCV_C2(64, 4, 3, 2)
    PSCV_C2(128, 4, 3, 1) PSCV_C2(128, 2, 3, 1)
    PSCV_C2(128, 4, 2, 1) PSCV_C2(64, 2, 2, 1)
#undef PSCV_C2
    set_error("pscv_conv2d: unsupported layer c_in=%d c_out=%d k=%d stride=%d", c_in, c_out, ks, stride);
    return -1;
}

}  // namespace pscv

extern "C" long pscv_pack_conv2d_weights(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, uint16_t* packed) {
    using namespace pscv;
    PSCV_CHECK_ARG(c_in > 0 && c_in <= c_in_padded && c_in_padded % 8 == 0 && c_out > 0 && (ks == 1 || ks == 2 || ks == 3 || ks == 5),
                   "pscv_pack_conv2d_weights: bad layer %d(%d) -> %d, k=%d", c_in, c_in_padded, c_out, ks);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_pack_conv2d_weights: dtype %d must be bf16 or fp16", dtype);
    const int nt = (c_out + 15) / 16, ntaps = ks * ks;
    const int nsteps = (ntaps * c_in_padded + 31) / 32;
    const long n = (long)nsteps * nt * 64 * 8;
    if (!packed) return n;
    PSCV_CHECK_ARG(w, "pscv_pack_conv2d_weights: null weight pointer");
    // A[m][k]: lane (m = lane & 15, g = lane >> 4), element j holds k = 32 s + 8 g + j = tap * c_in_padded + ci
    for (int s = 0; s < nsteps; ++s)
        for (int t = 0; t < nt; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = t * 16 + (lane & 15), k = s * 32 + (lane >> 4) * 8 + j;
                    const int tap = k / c_in_padded, ci = k % c_in_padded;
                    float v = 0.f;
                    if (co < c_out && tap < ntaps && ci < c_in) v = w[((long)co * c_in + ci) * ntaps + tap];
                    packed[(((long)s * nt + t) * 64 + lane) * 8 + j] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
                }
    return n;
}

extern "C" int pscv_conv2d_ex(const void* in, int dtype, const uint16_t* packed, const float* scale, const float* bias,
                              const void* skip, int skip_cstride, int skip_coff, void* out, int out_cstride, int out_coff,
                              int out_dtype, int B, int Hi, int Wi, int c_in, int c_out, int ks, int stride, int parity,
                              float neg_slope, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(in && packed && out, "pscv_conv2d: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0, "pscv_conv2d: bad sizes");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_conv2d: storage dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(out_dtype == dtype || out_dtype == PSCV_F32, "pscv_conv2d: out dtype %d must be the storage dtype or fp32", out_dtype);
    PSCV_CHECK_ARG(c_out % 4 == 0, "pscv_conv2d: c_out=%d must be a multiple of 4", c_out);
    PSCV_CHECK_ARG((ks == 3 && (stride == 1 || stride == 2)) || (ks == 5 && stride == 2) || (ks == 1 && (stride == 1 || stride == 2)) ||
                       (ks == 2 && stride == 1),
                   "pscv_conv2d: supported layers are k3 s1|s2 p1, k5 s2 p2, k1 s1|s2 p0 and the k2 parity sub-conv (got k%d s%d)", ks, stride);
    PSCV_CHECK_ARG(parity < 0 || ((ks == 2 || ks == 3) && stride == 1 && parity < 4),
                   "pscv_conv2d: an output parity (0..3) only applies to the stride-1 k2 / k3 sub-convolutions of a transposed conv");
    PSCV_CHECK_ARG(ks != 2 || parity >= 0, "pscv_conv2d: a k2 sub-convolution needs parity 0..3 (2 * row parity + column parity)");
    PSCV_CHECK_ARG((long)Hi * Wi * c_in * 2 < (1L << 32), "pscv_conv2d: one input map must stay below 4 GiB");
    PSCV_CHECK_ARG(out_cstride % 4 == 0 && out_coff % 4 == 0 && out_coff + c_out <= out_cstride, "pscv_conv2d: bad output channel slice");
    PSCV_CHECK_ARG(!skip || (skip_cstride % 4 == 0 && skip_coff % 4 == 0 && skip_coff + c_out <= skip_cstride), "pscv_conv2d: bad skip channel slice");
    Conv2dArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = reinterpret_cast<const uint4*>(packed);
    a.scale = scale; a.bias = bias; a.out = out; a.out_f32 = out_dtype == PSCV_F32;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out_cs = out_cstride; a.out_co = out_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.B = B; a.Hi = Hi; a.Wi = Wi;
    const int pad = ks == 2 ? 0 : ks / 2;
    a.Ho = ks == 2 ? Hi : (Hi + 2 * pad - ks) / stride + 1;
    a.Wo = ks == 2 ? Wi : (Wi + 2 * pad - ks) / stride + 1;
    a.oys = a.oxs = 1; a.oyo = a.oxo = 0; a.Hof = a.Ho; a.Wof = a.Wo;
    if (parity >= 0) { a.oys = a.oxs = 2; a.oyo = parity >> 1; a.oxo = parity & 1; a.Hof = 2 * Hi; a.Wof = 2 * Wi; }
    PSCV_CHECK_ARG(neg_slope >= 0.0f && neg_slope <= 1.0f, "pscv_conv2d: neg_slope=%g outside [0,1]", (double)neg_slope);
    a.cout = c_out; a.neg_slope = neg_slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc = dtype == PSCV_BF16 ? c2_dispatch<bf16_t>(a, c_in, c_out, ks, stride, st) : c2_dispatch<f16_t>(a, c_in, c_out, ks, stride, st);
    if (rc) return rc;
    PSCV_CHECK_LAUNCH("pscv_conv2d");
    return 0;
}

extern "C" int pscv_conv2d(const void* in, int dtype, const uint16_t* packed, const float* scale, const float* bias, void* out,
                           int out_dtype, int B, int Hi, int Wi, int c_in, int c_out, int ks, int stride, float neg_slope, void* stream) {
    PSCV_CHECK_ARG((ks == 3 && stride == 1) || (ks == 5 && stride == 2), "pscv_conv2d: only k3 s1 p1 and k5 s2 p2 layers (got k%d s%d); "
                   "see pscv_conv2d_ex", ks, stride);
    return pscv_conv2d_ex(in, dtype, packed, scale, bias, nullptr, 0, 0, out, c_out, 0, out_dtype, B, Hi, Wi, c_in, c_out, ks, stride, -1,
                          neg_slope, stream);
}
