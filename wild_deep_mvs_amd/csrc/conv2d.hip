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

    // ---- epilogue, 16-bit output with full channel tiles: the packed values of an M-tile (16 pixels x NT x 32 contiguous bytes
    // each) go through a per-wave LDS row (the brick is dead by now) and leave as 16-byte stores, a lane covering 8 consecutive
    // channels of a pixel; a residual input comes in the same way.  8-byte stores per lane and channel tile are store-ISSUE bound
    // (~7 B/clk/CU): the layers that write more than they read (3 -> 64, the stride-2 and 1x1 layers, the deconv parity
    // sub-convs) spent most of their time there.  Same arithmetic, same bits. ----
    // (measured per layer: 3 -> 64 118.6 -> 90.1 us at 1200 x 1600, 16 -> 32 k3 132 -> 108, 16 -> 32 k1 94 -> 72, the 64 -> 32 parity
    // sub-convs 54 -> 43; the layers with long reductions -- 128 -> 128 k3: 36 k-steps -- are MFMA / weight-fetch bound and lose
    // ~8 % to the extra barrier, so they keep the direct stores)
    constexpr bool WIDE = NSTEPS <= 16;
    if (WIDE && !a.out_f32 && !((a.out_cs | a.out_co) & 7) && (nt0 + NT) * 16 <= a.cout && (!a.skip || !((a.skip_cs | a.skip_co) & 7))) {
        constexpr int OPITCH = NT * 32 + 16;                 // (+16: conflict-free 8-byte writes)
        constexpr int NPC = 32 * NT, NPK = (NPC + 63) / 64;  // 16-byte pieces of an M-tile / per lane
        static_assert(!WIDE || 4 * 16 * OPITCH <= BH * BWL * VS + 64, "output staging rows must fit the brick's LDS");
        __syncthreads();                                     // every wave is done with the brick
        unsigned char* const so = smem + wave * (16 * OPITCH);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int mt = wave * MB + i;
            const int oy = oy0 + (mt >> 1), oxb = ox0 + (mt & 1) * 16;
            const unsigned long rowpix = ((unsigned long)b * a.Hof + (oy * a.oys + a.oyo)) * a.Wof;
            if (a.skip) {
#pragma unroll
                for (int k = 0; k < NPK; ++k) {
                    const int q = lane + 64 * k, p = q / (NT * 2), c = q - p * (NT * 2);
                    const int ox = oxb + p;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (q < NPC && oy < a.Ho && ox < a.Wo)
                        v = *reinterpret_cast<const uint4*>(a.skip + (rowpix + (ox * a.oxs + a.oxo)) * a.skip_cs + a.skip_co + nt0 * 16 + c * 8);
                    if (q < NPC) *reinterpret_cast<uint4*>(so + p * OPITCH + c * 16) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                float v[4], y[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaf(acc[i][m][k], sc[m][k], bi[m][k]);
                unsigned char* const slot = so + n * OPITCH + (m * 16 + g * 4) * 2;
                if (a.skip) {
                    const uint2 sv = *reinterpret_cast<const uint2*>(slot);
                    v[0] += Half16<H>::lo(sv.x); v[1] += Half16<H>::hi(sv.x); v[2] += Half16<H>::lo(sv.y); v[3] += Half16<H>::hi(sv.y);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = fmaxf(v[k], v[k] * a.neg_slope);
                *reinterpret_cast<uint2*>(slot) = make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
            }
            __builtin_amdgcn_wave_barrier();                 // (LDS executes a wave's operations in order)
            if (oy < a.Ho) {
#pragma unroll
                for (int k = 0; k < NPK; ++k) {
                    const int q = lane + 64 * k, p = q / (NT * 2), c = q - p * (NT * 2);
                    const int ox = oxb + p;
                    if (q < NPC && ox < a.Wo)
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (rowpix + (ox * a.oxs + a.oxo)) * a.out_cs + a.out_co + nt0 * 16 + c * 8) =
                            *reinterpret_cast<const uint4*>(so + p * OPITCH + c * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
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

// ---- 64-channel 3x3 stride-1 layers (CVP-MVSNet's FeaturePyramid 64 -> 64 / 64 -> 32, Vis FeatExt 64 -> 64) -------------------
// conv2d_kernel fetches its A fragments (the weights) from global memory per WAVE: at 64 input channels that is 18 k-steps x
// NT tiles x 1 KB = 74 KB per wave and 295 KB per 256-pixel workgroup against a 49 KB input brick -- six times more bytes
// through the CU's vector-memory path for the weights than for the image (TA busy 68 %, 25 % of the MFMA peak; round-2 notes).
// Here the layer's packed weights live in LDS, loaded ONCE per workgroup, and the workgroups are persistent: one per CU
// (74 KB weights + 49 KB brick), 512 threads, each wave one 32-pixel row of the 8 x 32 tile (2 M-tiles x NT N-tiles: 2 operand +
// NT weight `ds_read_b128` per 2 NT MFMAs), looping over tiles with the NEXT tile's brick in flight (global -> registers)
// while the current one is contracted, so HBM latency hides under the MFMAs instead of under other workgroups.
constexpr int C2W_TH = 8, C2W_BH = 10, C2W_BW = 34, C2W_THREADS = 512;
// 64 channels: brick pixels are 128 bytes apart and the eight 16-byte chunks of pixel p (= row * 34 + column) sit at chunk ^ (p & 7): a
// `ds_read_b128` lane group is 8 lanes with chunk A on pixels p0 + {0-3, 12-15} and 8 lanes with chunk A ^ 1 on p0 + {4-11}; under the
// XOR the sixteen 16-byte bank granules they touch are all different (the 144-byte padded pitch of conv2d_kernel collides for 7 of
// the 8 pairs of such a mixed group).
// 32 channels (CVP's 32 -> 32 / 32 -> 16 pyramid layers, the Vis extractor's full-resolution blocks): 64 bytes per pixel, so the
// granule of (p, chunk) is 4 (p mod 4) + chunk': the four pixels p, p + 4, p + 8, p + 12 of a lane group's 16-pixel window share
// 4 (p mod 4) and carry chunks (A, A ^ 1, A ^ 1, A); chunk' = chunk ^ (2 if (p >> 2) is odd) makes them (A, A ^ 3, A ^ 1, A ^ 2) or
// (A ^ 2, A ^ 1, A ^ 3, A): all different, for either kind of group.
template <int CIN> struct C2WGeom {
    static_assert(CIN == 32 || CIN == 64, "weights-in-LDS conv2d: 32 or 64 input channels");
    static constexpr int VS = CIN * 2;                          // bytes per brick pixel
    static constexpr int CPP = CIN / 8;                         // 16-byte chunks per pixel
    static constexpr int KSTEPS = CIN / 32;                     // 32-channel k-steps per tap
    static constexpr int STEPS = 9 * KSTEPS;
    static constexpr int BRICK = C2W_BH * C2W_BW * VS;          // 48 960 B / 21 760 B
    static constexpr int CHUNKS = C2W_BH * C2W_BW * CPP;        // 2720 / 1360
    static constexpr int NLD = (CHUNKS + C2W_THREADS - 1) / C2W_THREADS;   // 6 / 3 per thread
    static __device__ __forceinline__ int lds(int p, int chunk) {
        if constexpr (CIN == 64) return p * VS + ((chunk ^ (p & 7)) << 4);
        else return p * VS + ((chunk ^ ((p >> 1) & 2)) << 4);
    }
};

PSCV_PROF_BUFFER(c2w)
template <typename H, int NT, int CIN>
__global__ __launch_bounds__(C2W_THREADS, 2) void conv2d_wlds_kernel(const Conv2dArgs a, int n_tiles) {
    using G = C2WGeom<CIN>;
    constexpr int C2W_STEPS = G::STEPS, C2W_BRICK = G::BRICK, C2W_CHUNKS = G::CHUNKS, C2W_NLD = G::NLD, CPP = G::CPP;
    constexpr int W_BYTES = C2W_STEPS * NT * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int OPITCH = NT * 32 + 16;             // bytes per pixel of the output staging rows (+16: conflict-free 8-byte writes)
    unsigned char* const sw = smem;                  // [step][tile][lane] x 16 B
    unsigned char* const sb = smem + W_BYTES;        // brick
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const so = smem + W_BYTES + C2W_BRICK + wave * (32 * OPITCH);   // this wave's output row: 32 pixels
    const int n = lane & 15, g = lane >> 4;

    // weights -> LDS (once)
    for (int q = tid; q < C2W_STEPS * NT * 64; q += C2W_THREADS) {
        const int st = q / (NT * 64), rem = q - st * (NT * 64);
        const int m = rem >> 6, ln = rem & 63;
        *reinterpret_cast<uint4*>(sw + q * 16) = a.wpk[(st * a.nt_total + m) * 64 + ln];
    }
    // this thread's chunks of a brick (tile-invariant part)
    int cr[C2W_NLD], cbw[C2W_NLD], loff[C2W_NLD];
    unsigned ccoff[C2W_NLD];
    bool cin[C2W_NLD];
#pragma unroll
    for (int k = 0; k < C2W_NLD; ++k) {
        const int q = tid + C2W_THREADS * k;
        cin[k] = q < C2W_CHUNKS;
        const int r = q / (C2W_BW * CPP), rem = q - r * (C2W_BW * CPP);
        cr[k] = r; cbw[k] = rem / CPP;
        ccoff[k] = (unsigned)(rem % CPP) * 16u;
        loff[k] = G::lds(r * C2W_BW + cbw[k], rem % CPP);
    }
    float sc[NT][4], bi[NT][4];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = m * 16 + g * 4 + k;
            sc[m][k] = (a.scale && c < a.cout) ? a.scale[c] : 1.0f;
            bi[m][k] = (a.bias && c < a.cout) ? a.bias[c] : 0.0f;
        }
    const unsigned row_bytes = (unsigned)a.Wi * (unsigned)G::VS;
    const int tiles_per_img = a.nth * a.ntw;
    unsigned roff[C2W_NLD];                             // chunk offset from the brick's first pixel (interior tiles)
#pragma unroll
    for (int k = 0; k < C2W_NLD; ++k) roff[k] = cin[k] ? (unsigned)cr[k] * row_bytes + (unsigned)cbw[k] * (unsigned)G::VS + ccoff[k] : 0u;
    auto fetch = [&](int t, uint4 (&val)[C2W_NLD]) {
        const int b = t / tiles_per_img, rt = t - b * tiles_per_img;
        const int th_i = rt / a.ntw, tw_i = rt - th_i * a.ntw;
        const int iy0 = th_i * C2W_TH - 1, ix0 = tw_i * C2_TW - 1;
        const char* inb = reinterpret_cast<const char*>(a.in) + (unsigned long)b * a.Hi * row_bytes;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + C2W_BH <= a.Hi && ix0 + C2W_BW <= a.Wi) {
            // interior tile (workgroup-uniform): one scalar base + a precomputed 32-bit lane offset per chunk, no bounds tests --
            // the general form below cost ~1100 cycles per tile in 64-bit address arithmetic and predicates
            const char* base = inb + (unsigned long)iy0 * row_bytes + (unsigned)ix0 * (unsigned)G::VS;
#pragma unroll
            for (int k = 0; k < C2W_NLD; ++k) val[k] = *reinterpret_cast<const uint4*>(base + roff[k]);
            return;
        }
#pragma unroll
        for (int k = 0; k < C2W_NLD; ++k) {
            const int gy = iy0 + cr[k], gx = ix0 + cbw[k];
            const bool ok = cin[k] && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi;
            val[k] = make_uint4(0u, 0u, 0u, 0u);
            if (ok) val[k] = *reinterpret_cast<const uint4*>(inb + (unsigned long)gy * row_bytes + (unsigned)gx * (unsigned)G::VS + ccoff[k]);
        }
    };
    const int p_anchor = wave * C2W_BW + n;                      // this wave's row, column tile 0; tile 1 = + 16 pixels
    uint4 val[C2W_NLD];
    int t = blockIdx.x;
    if (t < n_tiles) fetch(t, val);
    PSCV_PROF_BEGIN   // (profile builds, summed over the tiles: brick landed + written | barrier | next fetch issued | k-loop | epilogue | barrier)
    for (; t < n_tiles; t += gridDim.x) {
#pragma unroll
        for (int k = 0; k < C2W_NLD; ++k)
            if (cin[k]) *reinterpret_cast<uint4*>(sb + loff[k]) = val[k];
        PSCV_STAMP_WAIT(0)
        __syncthreads();
        PSCV_STAMP(1)
        const int tn = t + gridDim.x;
        if (tn < n_tiles) fetch(tn, val);                        // in flight during the contraction below
        PSCV_STAMP(2)

        c2_f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < NT; ++m) acc[i][m] = c2_f32x4{0.f, 0.f, 0.f, 0.f};
        // k-loop, software-pipelined by hand: the operands of step s + 1 are requested BEFORE the MFMAs of step s (two register
        // sets).  Left to the compiler each group of four MFMAs waited for `ds_read`s issued right in front of it, and with two
        // waves per SIMD the LDS latency was exposed at every step (29 % of the MFMA peak).
        uint4 wf[2][NT], xf[2][2];
        auto request = [&](int s_, uint4 (&w_)[NT], uint4 (&x_)[2]) {
            const int tap = s_ / G::KSTEPS, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int m = 0; m < NT; ++m) w_[m] = *reinterpret_cast<const uint4*>(sw + ((s_ * NT + m) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) x_[i] = *reinterpret_cast<const uint4*>(sb + G::lds(p_anchor + i * 16 + kh * C2W_BW + kw, (s_ % G::KSTEPS) * 4 + g));
        };
        request(0, wf[0], xf[0]);
#pragma unroll
        for (int s = 0; s < C2W_STEPS; ++s) {
            if (s + 1 < C2W_STEPS) request(s + 1, wf[(s + 1) & 1], xf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < NT; ++m) acc[i][m] = C2Mfma<H>::run(wf[s & 1][m], xf[s & 1][i], acc[i][m]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PSCV_STAMP(3)
        // ---- epilogue: lane (n, g) owns channels 16 m + 4 g .. + 3 of pixel n of each of its two column tiles ----
        {
            const int b = t / tiles_per_img, rt = t - b * tiles_per_img;
            const int th_i = rt / a.ntw, tw_i = rt - th_i * a.ntw;
            const int oy = th_i * C2W_TH + wave;
            if (!a.out_f32 && !((a.out_cs | a.out_co) & 7) && a.cout == NT * 16 && (!a.skip || !((a.skip_cs | a.skip_co) & 7))) {
                // 16-bit output without a residual (every CVP pyramid layer): the wave's row is 32 pixels x NT x 32 contiguous
                // bytes in memory.  An 8-byte store per lane and (column tile, channel tile) is store-ISSUE bound (~7 B/clk/CU: the
                // 32 KB of a tile cost as many cycles as its MFMAs); the packed values go through a per-wave LDS row instead and
                // leave as 16-byte stores, a lane covering 8 consecutive channels of a pixel (same bits)
                const unsigned long rowpix = ((unsigned long)b * a.Hof + (oy * a.oys + a.oyo)) * a.Wof;
                if (a.skip) {
                    // the residual input (the Vis extractor's blocks) takes the same road in: 16-byte loads of the row into the
                    // staging row, from which each lane picks the 8 bytes of its own channels (fp32 add before the activation,
                    // like conv2d_kernel: same bits)
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const int q = lane + 64 * k, p = q / (NT * 2), c = q - p * (NT * 2);
                        const int ox = tw_i * C2_TW + p;
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (oy < a.Ho && ox < a.Wo) v = *reinterpret_cast<const uint4*>(a.skip + (rowpix + (ox * a.oxs + a.oxo)) * a.skip_cs + a.skip_co + c * 8);
                        *reinterpret_cast<uint4*>(so + p * OPITCH + c * 16) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        float v[4], y[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = fmaf(acc[i][m][k], sc[m][k], bi[m][k]);
                        unsigned char* const slot = so + (i * 16 + n) * OPITCH + (m * 16 + g * 4) * 2;
                        if (a.skip) {
                            const uint2 sv = *reinterpret_cast<const uint2*>(slot);
                            v[0] += Half16<H>::lo(sv.x); v[1] += Half16<H>::hi(sv.x); v[2] += Half16<H>::lo(sv.y); v[3] += Half16<H>::hi(sv.y);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) y[k] = fmaxf(v[k], v[k] * a.neg_slope);
                        *reinterpret_cast<uint2*>(slot) = make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                    }
                __builtin_amdgcn_wave_barrier();          // (LDS executes a wave's operations in order: the reads below see the row)
                if (oy < a.Ho) {
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const int q = lane + 64 * k, p = q / (NT * 2), c = q - p * (NT * 2);
                        const uint4 v = *reinterpret_cast<const uint4*>(so + p * OPITCH + c * 16);
                        const int ox = tw_i * C2_TW + p;
                        if (ox < a.Wo)
                            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (rowpix + (ox * a.oxs + a.oxo)) * a.out_cs + a.out_co + c * 8) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // (issuing these stores one tile late, between the k-steps of the next tile, was measured: the k-loop grows by what
                // the epilogue loses -- the CU moves 43.5 KB in + 32 KB out per tile through a ~10 B/clk memory path either way)
            } else
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ox = tw_i * C2_TW + i * 16 + n;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                const unsigned long pix = ((unsigned long)b * a.Hof + (oy * a.oys + a.oyo)) * a.Wof + (ox * a.oxs + a.oxo);
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const int c0 = m * 16 + g * 4;
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
        PSCV_STAMP(4)
        __syncthreads();                                         // the brick is free for the next tile
        PSCV_STAMP(5)
    }
    PSCV_PROF_END(c2w, blockIdx.x)
}

}  // namespace pscv
PSCV_PROF_EXPORT(c2w)
namespace pscv {
Knob g_conv2d_wlds = {1, KNOB_SPARE1};    // pscv_set_tuning("conv2d_wlds", 0): 64-channel k3 s1 layers back on conv2d_kernel; 2: at any size

template <typename H, int NT, int CIN>
static int c2w_launch(Conv2dArgs& a, hipStream_t st) {
    constexpr int LDS = C2WGeom<CIN>::STEPS * NT * 1024 + C2WGeom<CIN>::BRICK + 8 * 32 * (NT * 32 + 16);
    static_assert(LDS <= 160 * 1024, "weights + brick + output rows must fit the LDS");
    a.nth = c2_ceil_div(a.Ho, C2W_TH); a.ntw = c2_ceil_div(a.Wo, C2_TW);
    const long tiles = (long)a.B * a.nth * a.ntw;
    if (tiles <= 0 || tiles > 0x7fffffffL) { set_error("pscv_conv2d: bad grid %ld", tiles); return -1; }
    auto kern = conv2d_wlds_kernel<H, NT, CIN>;
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS);
        if (e != hipSuccess) { set_error("pscv_conv2d: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
    }
    const int n_cu = device_cu_count();            // of the current device (cached per device)
    if (n_cu <= 0) { set_error("pscv_conv2d: device query failed"); return -2; }
    const int per_cu = LDS <= 80 * 1024 ? 2 : 1;
    const long grid = tiles < (long)n_cu * per_cu ? tiles : (long)n_cu * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C2W_THREADS), LDS, st, a, (int)tiles);
    return 0;
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
        {
            hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS);
            if (e != hipSuccess) { set_error("pscv_conv2d: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e)); return -2; }
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
    if (ks == 3 && stride == 1 && g_conv2d_wlds &&
        (g_conv2d_wlds == 2 || (long)a.B * c2_ceil_div(a.Ho, C2W_TH) * c2_ceil_div(a.Wo, C2_TW) >= 512)) {
        if (c_in == 64 && nt == 4) return c2w_launch<H, 4, 64>(a, st);
        if (c_in == 64 && nt == 2) return c2w_launch<H, 2, 64>(a, st);
        if (c_in == 32 && nt == 2) return c2w_launch<H, 2, 32>(a, st);
        if (c_in == 32 && nt == 1) return c2w_launch<H, 1, 32>(a, st);
    }
    const int ntw = nt > 4 ? 4 : nt;
    const int split = (nt + ntw - 1) / ntw;
    if (nt > 4 && nt % 4) { set_error("pscv_conv2d: c_out=%d above 64 must be a multiple of 64", c_out); return -1; }
#define PSCV_C2(CI, NTV, K, S) if (c_in == CI && ntw == NTV && ks == K && stride == S) return c2_launch<H, CI, NTV, K, S>(a, split, st);
    PSCV_C2(8, 1, 3, 1) PSCV_C2(16, 1, 3, 1) PSCV_C2(32, 2, 3, 1) PSCV_C2(16, 2, 3, 1) PSCV_C2(32, 1, 3, 1)
    PSCV_C2(8, 4, 3, 1) PSCV_C2(64, 4, 3, 1) PSCV_C2(64, 2, 3, 1)                       // CVP pyramid: 3->64, 64->64, 64->32
    PSCV_C2(32, 4, 3, 1)                                                                // (its 64->32 layer's adjoint: 32->64, training)
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

// Same packing on the device (w and packed are device pointers): a training step rebuilds the packed forward and adjoint layers of
// the 2-D extractor after every optimiser step, and the host version costs a device -> host copy (a stream synchronisation), a
// host loop and an upload per layer -- 21 round trips per MVSNet training step.
namespace pscv {
__global__ void pack_conv2d_kernel(const float* __restrict__ w, int c_in, int c_in_padded, int c_out, int ntaps, int nt, int dtype, long n,
                                   uint16_t* __restrict__ packed, int adjoint) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long blk = idx >> 9;
    const int t = (int)(blk % nt), s = (int)(blk / nt);
    const int co = t * 16 + (lane & 15), k = s * 32 + (lane >> 4) * 8 + j;
    const int tap = k / c_in_padded, ci = k % c_in_padded;
    float v = 0.f;
    // adjoint: w is the weight [c_in][c_out][k][k] of the FORWARD layer (c_in of this layer = its output channels); the adjoint conv
    // swaps the channel axes and flips the taps -- read in place instead of flip / transpose / contiguous launches per layer and step
    if (co < c_out && tap < ntaps && ci < c_in)
        v = adjoint ? w[((long)ci * c_out + co) * ntaps + (ntaps - 1 - tap)] : w[((long)co * c_in + ci) * ntaps + tap];
    packed[idx] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
}
}  // namespace pscv

extern "C" int pscv_pack_conv2d_weights_device_ex(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, int adjoint,
                                                  uint16_t* packed, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(c_in > 0 && c_in <= c_in_padded && c_in_padded % 8 == 0 && c_out > 0 && (ks == 1 || ks == 2 || ks == 3 || ks == 5),
                   "pscv_pack_conv2d_weights_device: bad layer %d(%d) -> %d, k=%d", c_in, c_in_padded, c_out, ks);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_pack_conv2d_weights_device: dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(w && packed, "pscv_pack_conv2d_weights_device: null pointer argument");
    const int nt = (c_out + 15) / 16, ntaps = ks * ks;
    const int nsteps = (ntaps * c_in_padded + 31) / 32;
    const long n = (long)nsteps * nt * 64 * 8;
    hipLaunchKernelGGL(pack_conv2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, c_in,
                       c_in_padded, c_out, ntaps, nt, dtype, n, packed, adjoint ? 1 : 0);
    PSCV_CHECK_LAUNCH("pscv_pack_conv2d_weights_device");
    return 0;
}
extern "C" int pscv_pack_conv2d_weights_device(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, uint16_t* packed,
                                               void* stream) {
    return pscv_pack_conv2d_weights_device_ex(w, c_in, c_in_padded, c_out, ks, dtype, 0, packed, stream);
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
