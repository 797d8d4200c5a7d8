// Training-mode pieces of the 3-D regulariser that are byte work, not contractions (gfx950):
//   - per-channel batch statistics of a channels-last 16-bit volume (BatchNorm3d in train(): the reference's
//     ConvBnReLU3D / Sequential(ConvTranspose3d, BatchNorm3d, ReLU) blocks, models/MVSNet/module.py:41-58,
//     models/MVSNet/model.py:57-70, normalise with the statistics of the current batch),
//   - the normalise + ReLU + skip-add pass,
//   - the two passes of the BatchNorm backward (reduce, then apply),
//   - the backward of softmax-over-D + depth regression (models/MVSNet/model.py:207-209).
// All are HBM-bound single passes: one 16-byte chunk (8 channels of a voxel) per lane per iteration, a lane keeps
// the same channel group for its whole grid-stride walk so per-channel constants sit in registers and per-channel
// sums need no cross-lane traffic until the end.  Reductions are two-phase (block partials in a caller-provided
// workspace, then one small finishing block) so results are bit-reproducible run to run: no float atomics.
#include "pscv_common.h"

namespace pscv {

constexpr int RED_BLOCKS = 1024;   // partial-sum blocks of the two-phase reductions (workspace = RED_BLOCKS * 2 * 64 floats)

template <typename H> __device__ __forceinline__ void unpack8(const uint4& a, float (&v)[8]) {
    v[0] = Half16<H>::lo(a.x); v[1] = Half16<H>::hi(a.x); v[2] = Half16<H>::lo(a.y); v[3] = Half16<H>::hi(a.y);
    v[4] = Half16<H>::lo(a.z); v[5] = Half16<H>::hi(a.z); v[6] = Half16<H>::lo(a.w); v[7] = Half16<H>::hi(a.w);
}
template <typename H> __device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(Half16<H>::pack(v[0], v[1]), Half16<H>::pack(v[2], v[3]), Half16<H>::pack(v[4], v[5]),
                      Half16<H>::pack(v[6], v[7]));
}

// Block-level sum of two 8-channel register vectors into partials[blk][2][C]; lanes with the same channel group
// (tid % CG) are combined through LDS in a fixed order.
template <int CG>
__device__ __forceinline__ void block_reduce2(const float (&s0)[8], const float (&s1)[8], float* partials, int C) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[tid][j] = s0[j]; red[tid][8 + j] = s1[j]; }
    __syncthreads();
    // thread (cg, j16) sums the 256 / CG lanes of its channel group, in lane order
    if (tid < CG * 16) {
        const int cg = tid / 16, j = tid % 16;
        float acc = 0.0f;
        for (int l = cg; l < 256; l += CG) acc += red[l][j];
        const int which = j >> 3, c = cg * 8 + (j & 7);
        partials[((long)blockIdx.x * 2 + which) * C + c] = acc;
    }
}

// out[i] = sum_b partials[b][i] in a fixed order.  One block per 16 outputs: 16 columns x 16 row groups, each thread walks
// its rows with four independent accumulators, then the 16 groups are combined through LDS in group order.  (A first
// version used one thread per output walking all 1024 rows: 70 us of dependent L2 latency per call.)
__global__ __launch_bounds__(256) void finish_partials_kernel(const float* __restrict__ partials, int nblk, int n, float* __restrict__ out) {
    __shared__ float red[16][17];
    partials += (long)blockIdx.y * nblk * n;     // groups (blockIdx.y): each has its own run of partial rows and its own output row
    out += (long)blockIdx.y * n;
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int b = g;
        for (; b + 48 < nblk; b += 64) {
            a0 += partials[(long)b * n + i]; a1 += partials[(long)(b + 16) * n + i];
            a2 += partials[(long)(b + 32) * n + i]; a3 += partials[(long)(b + 48) * n + i];
        }
        for (; b < nblk; b += 16) a0 += partials[(long)b * n + i];
    }
    red[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][c];
        out[i] = s;
    }
}

// ---- batch statistics: sum and sum of squares per channel ---------------------------------------------------------
template <typename H, int CG>
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint4* __restrict__ y, long nchunk, float* __restrict__ partials, int C) {
    // groups (blockIdx.y): the statistics of group g cover chunks [g nchunk, (g + 1) nchunk) -- the views of a 2-D extractor batch,
    // each normalised with its own batch statistics like the reference's per-view calls (models/MVSNet/model.py:101-107)
    y += (long)blockIdx.y * nchunk;
    partials += (long)blockIdx.y * gridDim.x * 2 * C;
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float v[8];
        unpack8<H>(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] += v[j]; s1[j] = fmaf(v[j], v[j], s1[j]); }
    }
    block_reduce2<CG>(s0, s1, partials, C);
}

// ---- forward: out = [relu](y * scale + bias) + skip -----------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(256) void bn_act_kernel(const uint4* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ bias, const uint4* __restrict__ skip,
                                                     uint4* __restrict__ out, long nchunk, int CG, int relu, int pstride) {
    const int cg = threadIdx.x % CG;   // 256 and the grid stride are multiples of CG
    y += (long)blockIdx.y * nchunk; out += (long)blockIdx.y * nchunk;           // groups: own slice, own constants
    if (skip) skip += (long)blockIdx.y * nchunk;
    scale += (long)blockIdx.y * pstride; bias += (long)blockIdx.y * pstride;
    float sc[8], bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; bi[j] = bias[cg * 8 + j]; }
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float v[8];
        unpack8<H>(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = fmaf(v[j], sc[j], bi[j]);
            if (relu == 1) v[j] = relu_floor(v[j], 0.0f);
        }
        if (skip) {
            float s[8];
            unpack8<H>(skip[i], s);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += s[j];
        }
        if (relu == 2) {   // ReLU after the residual add (Vis BasicBlock, models/VisMVSNet/nn_utils.py:123-171)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = relu_floor(v[j], 0.0f);
        }
        out[i] = pack8<H>(v);
    }
}

// ---- backward of a ReLU applied AFTER a residual add: dpre = dout * [out > 0] ------------------------------------------
// (slope = 0: ReLU; slope = 0.1: the LeakyReLU of CVP-MVSNet's 2-D pyramid, whose output has the sign of its input)
template <typename H>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ out,
                                                       uint4* __restrict__ dpre, long nchunk, float slope) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float g[8], o[8];
        unpack8<H>(dout[i], g);
        unpack8<H>(out[i], o);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.0f ? g[j] : g[j] * slope;
        dpre[i] = pack8<H>(g);
    }
}

// the same pass that also returns sum_voxels dpre per channel (the bias gradient of a conv + bias + LeakyReLU layer: CVP-MVSNet's 2-D
// pyramid; a separate statistics pass re-read dpre: ~1 ms of its training step)
template <typename H, int CG>
__global__ __launch_bounds__(256) void relu_bwd_sum_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ out,
                                                           uint4* __restrict__ dpre, long nchunk, float slope, float* __restrict__ partials, int C) {
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float g[8], o[8];
        unpack8<H>(dout[i], g);
        unpack8<H>(out[i], o);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.0f ? g[j] : g[j] * slope;
        const uint4 pk = pack8<H>(g);
        dpre[i] = pk;
        float r[8];
        unpack8<H>(pk, r);                       // the sum of what was STORED (what the weight-gradient and adjoint launches read)
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] += r[j];
    }
    block_reduce2<CG>(s0, s1, partials, C);
}

// ---- backward, pass 1: dz = dact * [z > 0];  sums of dz and dz * y per channel --------------------------------------
template <typename H, int CG>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint4* __restrict__ dact, const uint4* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            long nchunk, float* __restrict__ partials, int C, int relu, int pstride) {
    const int cg = threadIdx.x % CG;
    dact += (long)blockIdx.y * nchunk; y += (long)blockIdx.y * nchunk;
    scale += (long)blockIdx.y * pstride; bias += (long)blockIdx.y * pstride;
    partials += (long)blockIdx.y * gridDim.x * 2 * C;
    float sc[8], bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; bi[j] = bias[cg * 8 + j]; }
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float g[8], v[8];
        unpack8<H>(dact[i], g);
        unpack8<H>(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(v[j], sc[j], bi[j]);
            const float dz = (relu && !(z > 0.0f)) ? 0.0f : g[j];
            s0[j] += dz;
            s1[j] = fmaf(dz, v[j], s1[j]);
        }
    }
    block_reduce2<CG>(s0, s1, partials, C);
}

// ---- backward, pass 2: dy = ca * dz + cb * y + cc --------------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint4* __restrict__ dact, const uint4* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ cc, uint4* __restrict__ dy, long nchunk,
                                                           int CG, int relu, int pstride, int cstride) {
    const int cg = threadIdx.x % CG;
    dact += (long)blockIdx.y * nchunk; y += (long)blockIdx.y * nchunk; dy += (long)blockIdx.y * nchunk;
    scale += (long)blockIdx.y * pstride; bias += (long)blockIdx.y * pstride;
    ca += (long)blockIdx.y * cstride; cb += (long)blockIdx.y * cstride; cc += (long)blockIdx.y * cstride;
    float sc[8], bi[8], a[8], b[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[cg * 8 + j]; bi[j] = bias[cg * 8 + j];
        a[j] = ca[cg * 8 + j]; b[j] = cb[cg * 8 + j]; c[j] = cc[cg * 8 + j];
    }
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += stride) {
        float g[8], v[8], o[8];
        unpack8<H>(dact[i], g);
        unpack8<H>(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(v[j], sc[j], bi[j]);
            const float dz = (relu && !(z > 0.0f)) ? 0.0f : g[j];
            o[j] = fmaf(a[j], dz, fmaf(b[j], v[j], c[j]));
        }
        dy[i] = pack8<H>(o);
    }
}

// ---- softmax over D + regression heads, backward ----------------------------------------------------------------------
// p = softmax(logits).  Heads (each optional) and their derivative w.r.t. logit_d:
//   depth   = sum_d p_d depth_d            g_depth p_d (depth_d - depth)                 models/MVSNet/model.py:207-209
//   index   = sum_d p_d d                  g_index p_d (d - index)                       models/VisMVSNet/nn_utils.py:453-466
//   entropy = -sum_d p_d log clamp(p_d, 1e-9, 1)   g_ent p_d (a_d - sum_k p_k a_k), a_d = -(log clamp(p_d) + [p_d > 1e-9])   nn_utils.py:469-470
// 32 pixels x 8 depth slices per workgroup, three LDS meetings of the slices (max; sum; expectations); reads and writes are coalesced.
// Output: the gradient volume in the conv engine's layout, [B,D,h,w,8] 16-bit with the value in channel 0 and
// channels 1-7 zero (the 1-channel heads' backward then runs on the 8-channel MFMA kernels).
// 256 threads = 32 x-adjacent pixels x 8 depth slices (the forward kernel's decomposition): a thread walks D / 8 planes per pass and the
// slices meet in LDS three times (max; sum e; the expectations).  One lane per pixel over all of D -- 20 K lanes at the headline size,
// 80 workgroups on 256 CUs, four dependent passes -- took 166 us.
constexpr int SB_PX = 32, SB_NS = 8;
template <typename H>
__global__ __launch_bounds__(256) void softargmin_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ depth,
                                                             long depth_bstride, int depth_per_pixel,
                                                             const float* __restrict__ gdepth, const float* __restrict__ gindex,
                                                             const float* __restrict__ gentropy, uint4* __restrict__ dl8, int B,
                                                             int D, int hw) {
    __shared__ float sh[3][SB_NS][SB_PX];
    const int px = threadIdx.x % SB_PX, sl = threadIdx.x / SB_PX;
    const long npix = (long)B * hw;
    long p = (long)blockIdx.x * SB_PX + px;
    const bool active = p < npix;
    p = active ? p : npix - 1;
    const int b = (int)(p / hw);
    const int pix = (int)(p - (long)b * hw);
    const float* lp = logits + (long)b * D * hw + pix;
    const float* dp = gdepth ? depth + (long)b * depth_bstride + (depth_per_pixel ? pix : 0) : nullptr;
    const long dstep = depth_per_pixel ? hw : 1;
    const int per = (D + SB_NS - 1) / SB_NS;
    const int d0 = sl * per, d1 = min(D, d0 + per);
    auto reduce = [&](int k, float v, bool is_max) -> float {
        sh[k][sl][px] = v;
        __syncthreads();
        float r = sh[k][0][px];
#pragma unroll
        for (int i = 1; i < SB_NS; ++i) r = is_max ? fmaxf(r, sh[k][i][px]) : r + sh[k][i][px];      // fixed order: every slice gets the same bits
        return r;
    };
    float m = -INFINITY;
#pragma unroll 4
    for (int d = d0; d < d1; ++d) m = fmaxf(m, lp[(long)d * hw]);
    m = reduce(0, m, true);
    float z = 0.0f;
#pragma unroll 4
    for (int d = d0; d < d1; ++d) z += __expf(lp[(long)d * hw] - m);
    const float inv = 1.0f / reduce(1, z, false);
    float e_depth = 0.0f, e_index = 0.0f, e_a = 0.0f;
#pragma unroll 4
    for (int d = d0; d < d1; ++d) {
        const float pd = __expf(lp[(long)d * hw] - m) * inv;
        if (gdepth) e_depth = fmaf(pd, dp[d * dstep], e_depth);
        e_index = fmaf(pd, (float)d, e_index);
        if (gentropy) e_a = fmaf(pd, -(__logf(fminf(fmaxf(pd, 1e-9f), 1.0f)) + (pd > 1e-9f ? 1.0f : 0.0f)), e_a);
    }
    __syncthreads();                         // (the three tables are reused below)
    e_index = reduce(0, e_index, false);
    if (gdepth) e_depth = reduce(1, e_depth, false);
    if (gentropy) e_a = reduce(2, e_a, false);
    if (!active) return;
    const float gd = gdepth ? gdepth[p] : 0.0f, gi = gindex ? gindex[p] : 0.0f, ge = gentropy ? gentropy[p] : 0.0f;
    uint4* op = dl8 + (long)b * D * hw + pix;
#pragma unroll 4
    for (int d = d0; d < d1; ++d) {
        const float pd = __expf(lp[(long)d * hw] - m) * inv;
        float v = gi * ((float)d - e_index);
        if (gdepth) v = fmaf(gd, dp[d * dstep] - e_depth, v);
        if (gentropy) v = fmaf(ge, -(__logf(fminf(fmaxf(pd, 1e-9f), 1.0f)) + (pd > 1e-9f ? 1.0f : 0.0f)) - e_a, v);
        op[(long)d * hw] = make_uint4(Half16<H>::pack(v * pd, 0.0f), 0u, 0u, 0u);
    }
}

static int grid_for(long nchunk) {
    long nb = (nchunk + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    return (int)nb;
}
static int red_grid_for(long nchunk) {
    long nb = (nchunk + 255) / 256;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// ---- per-channel BatchNorm bookkeeping: the dozen C-element tensor ops between the reductions and the apply passes, one launch each
// (a Vis-MVSNet training step has 75 BatchNorm applications: ~2000 launches of 8-64-element ATen kernels were 10 % of its wall time) ----
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float nvox, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ run_mean, float* __restrict__ run_var,
                                   long long* __restrict__ batches, float* __restrict__ out, int C, int groups) {
    const int c = threadIdx.x;
    if (c == 0 && batches) *batches += groups;
    if (c >= C) return;
    // groups in order: the module saw them as consecutive forward calls (one per view), each updating the running statistics
    for (int g = 0; g < groups; ++g, sums += 2 * C, out += 4 * C) {
        const float mean = sums[c] / nvox;
        const float var = fmaxf(sums[C + c] / nvox - mean * mean, 0.0f);
        const float invstd = 1.0f / sqrtf(var + eps);
        const float scale = (gamma ? gamma[c] : 1.0f) * invstd;
        out[c] = scale;
        out[C + c] = (beta ? beta[c] : 0.0f) - mean * scale;
        out[2 * C + c] = mean;
        out[3 * C + c] = invstd;
        if (run_mean) {      // nn.BatchNorm3d in train(): running = (1 - m) running + m batch, unbiased variance
            run_mean[c] = run_mean[c] * (1.0f - momentum) + mean * momentum;
            run_var[c] = run_var[c] * (1.0f - momentum) + var * (nvox / fmaxf(nvox - 1.0f, 1.0f)) * momentum;
        }
    }
}

__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ s, const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, float nvox, float* __restrict__ out, int C, int mstride) {
    const int c = threadIdx.x;
    if (c >= C) return;
    s += (long)blockIdx.x * 2 * C; out += (long)blockIdx.x * 5 * C;               // groups (blockIdx.x)
    mean += (long)blockIdx.x * mstride; invstd += (long)blockIdx.x * mstride;
    const float s1 = s[c];                                        // sum dz       = d beta
    const float s2 = invstd[c] * (s[C + c] - mean[c] * s1);       // sum dz xhat  = d gamma
    const float k = (gamma ? gamma[c] : 1.0f) * invstd[c];
    out[c] = k;                                                   // dy = ca dz + cb y + cc
    out[C + c] = -k * invstd[c] * s2 / nvox;
    out[2 * C + c] = -k * s1 / nvox + k * invstd[c] * mean[c] * s2 / nvox;
    out[3 * C + c] = s2;
    out[4 * C + c] = s1;
}

}  // namespace pscv

using namespace pscv;

extern "C" long pscv_train_workspace_floats(void) { return (long)RED_BLOCKS * 2 * 128; }

// ---- grouped forms: `groups` consecutive slices of nvox voxels each, every group with its own statistics / constants (the views of
// a 2-D extractor batch, normalised per view like the reference's per-view calls, models/MVSNet/model.py:101-107).  Constants of
// group g start `param_stride` floats after those of group g - 1 (pscv_bn_finalize_grouped writes [groups][4][C]:
// scale, bias, mean, invstd; pscv_bn_bwd_coeffs_grouped [groups][5][C]: ca, cb, cc, d gamma, d beta).  groups = 1 is the plain form.
static int bn_check(const char* fn, const void* y, int dtype, long nvox, int groups, int C) {
    PSCV_CHECK_ARG(y, "%s: null pointer argument", fn);
    PSCV_CHECK_ARG(C == 8 || C == 16 || C == 32 || C == 64 || C == 128, "%s: C=%d must be 8, 16, 32, 64 or 128", fn, C);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "%s: dtype %d must be bf16 or fp16", fn, dtype);
    PSCV_CHECK_ARG(nvox > 0 && groups >= 1 && groups <= 256, "%s: empty volume or bad group count %d", fn, groups);
    return 0;
}
static int red_grid_grouped(long nchunk, int groups) {
    int nb = red_grid_for(nchunk);
    const int cap = RED_BLOCKS / groups;           // the workspace holds RED_BLOCKS partial rows in total
    return nb > cap ? (cap < 1 ? 1 : cap) : nb;
}

extern "C" int pscv_bn_stats_grouped(const void* y, int dtype, long nvox, int groups, int C, float* workspace, float* sums, void* stream) {
    if (bn_check("pscv_bn_stats", y, dtype, nvox, groups, C)) return -1;
    PSCV_CHECK_ARG(workspace && sums, "pscv_bn_stats: null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const int nb = red_grid_grouped(nchunk, groups);
    const uint4* yp = reinterpret_cast<const uint4*>(y);
    const dim3 grid(nb, groups);
#define PSCV_STATS(HT)                                                                                     \
    switch (C / 8) {                                                                                       \
        case 1: hipLaunchKernelGGL((bn_stats_kernel<HT, 1>), grid, dim3(256), 0, st, yp, nchunk, workspace, C); break; \
        case 2: hipLaunchKernelGGL((bn_stats_kernel<HT, 2>), grid, dim3(256), 0, st, yp, nchunk, workspace, C); break; \
        case 4: hipLaunchKernelGGL((bn_stats_kernel<HT, 4>), grid, dim3(256), 0, st, yp, nchunk, workspace, C); break; \
        case 8: hipLaunchKernelGGL((bn_stats_kernel<HT, 8>), grid, dim3(256), 0, st, yp, nchunk, workspace, C); break; \
        default: hipLaunchKernelGGL((bn_stats_kernel<HT, 16>), grid, dim3(256), 0, st, yp, nchunk, workspace, C); break; \
    }
    if (dtype == PSCV_BF16) { PSCV_STATS(bf16_t) } else { PSCV_STATS(f16_t) }
#undef PSCV_STATS
    PSCV_CHECK_LAUNCH("pscv_bn_stats");
    hipLaunchKernelGGL(finish_partials_kernel, dim3((2 * C + 15) / 16, groups), dim3(256), 0, st, workspace, nb, 2 * C, sums);
    PSCV_CHECK_LAUNCH("pscv_bn_stats(finish)");
    return 0;
}
extern "C" int pscv_bn_stats(const void* y, int dtype, long nvox, int C, float* workspace, float* sums, void* stream) {
    return pscv_bn_stats_grouped(y, dtype, nvox, 1, C, workspace, sums, stream);
}

extern "C" int pscv_bn_act_grouped(const void* y, int dtype, long nvox, int groups, int C, const float* scale, const float* bias,
                                   int param_stride, int relu, const void* skip, void* out, void* stream) {
    if (bn_check("pscv_bn_act", y, dtype, nvox, groups, C)) return -1;
    PSCV_CHECK_ARG(scale && bias && out, "pscv_bn_act: null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const dim3 grid(grid_for(nchunk), groups);
    if (dtype == PSCV_BF16)
        hipLaunchKernelGGL(bn_act_kernel<bf16_t>, grid, dim3(256), 0, st, (const uint4*)y, scale, bias, (const uint4*)skip, (uint4*)out, nchunk, C / 8, relu, param_stride);
    else
        hipLaunchKernelGGL(bn_act_kernel<f16_t>, grid, dim3(256), 0, st, (const uint4*)y, scale, bias, (const uint4*)skip, (uint4*)out, nchunk, C / 8, relu, param_stride);
    PSCV_CHECK_LAUNCH("pscv_bn_act");
    return 0;
}
extern "C" int pscv_bn_act(const void* y, int dtype, long nvox, int C, const float* scale, const float* bias, int relu,
                           const void* skip, void* out, void* stream) {
    return pscv_bn_act_grouped(y, dtype, nvox, 1, C, scale, bias, 0, relu, skip, out, stream);
}

extern "C" int pscv_bn_bwd_reduce_grouped(const void* dact, const void* y, int dtype, long nvox, int groups, int C, const float* scale,
                                          const float* bias, int param_stride, int relu, float* workspace, float* sums, void* stream) {
    if (bn_check("pscv_bn_bwd_reduce", y, dtype, nvox, groups, C)) return -1;
    PSCV_CHECK_ARG(dact && scale && bias && workspace && sums, "pscv_bn_bwd_reduce: null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const int nb = red_grid_grouped(nchunk, groups);
    const dim3 grid(nb, groups);
    const uint4 *gp = (const uint4*)dact, *yp = (const uint4*)y;
#define PSCV_RED(HT)                                                                                                              \
    switch (C / 8) {                                                                                                              \
        case 1: hipLaunchKernelGGL((bn_bwd_reduce_kernel<HT, 1>), grid, dim3(256), 0, st, gp, yp, scale, bias, nchunk, workspace, C, relu, param_stride); break; \
        case 2: hipLaunchKernelGGL((bn_bwd_reduce_kernel<HT, 2>), grid, dim3(256), 0, st, gp, yp, scale, bias, nchunk, workspace, C, relu, param_stride); break; \
        case 4: hipLaunchKernelGGL((bn_bwd_reduce_kernel<HT, 4>), grid, dim3(256), 0, st, gp, yp, scale, bias, nchunk, workspace, C, relu, param_stride); break; \
        case 8: hipLaunchKernelGGL((bn_bwd_reduce_kernel<HT, 8>), grid, dim3(256), 0, st, gp, yp, scale, bias, nchunk, workspace, C, relu, param_stride); break; \
        default: hipLaunchKernelGGL((bn_bwd_reduce_kernel<HT, 16>), grid, dim3(256), 0, st, gp, yp, scale, bias, nchunk, workspace, C, relu, param_stride); break; \
    }
    if (dtype == PSCV_BF16) { PSCV_RED(bf16_t) } else { PSCV_RED(f16_t) }
#undef PSCV_RED
    PSCV_CHECK_LAUNCH("pscv_bn_bwd_reduce");
    hipLaunchKernelGGL(finish_partials_kernel, dim3((2 * C + 15) / 16, groups), dim3(256), 0, st, workspace, nb, 2 * C, sums);
    PSCV_CHECK_LAUNCH("pscv_bn_bwd_reduce(finish)");
    return 0;
}
extern "C" int pscv_bn_bwd_reduce(const void* dact, const void* y, int dtype, long nvox, int C, const float* scale,
                                  const float* bias, int relu, float* workspace, float* sums, void* stream) {
    return pscv_bn_bwd_reduce_grouped(dact, y, dtype, nvox, 1, C, scale, bias, 0, relu, workspace, sums, stream);
}

extern "C" int pscv_bn_bwd_apply_grouped(const void* dact, const void* y, int dtype, long nvox, int groups, int C, const float* scale,
                                         const float* bias, int param_stride, int relu, const float* ca, const float* cb, const float* cc,
                                         int coeff_stride, void* dy, void* stream) {
    if (bn_check("pscv_bn_bwd_apply", y, dtype, nvox, groups, C)) return -1;
    PSCV_CHECK_ARG(dact && scale && bias && ca && cb && cc && dy, "pscv_bn_bwd_apply: null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const dim3 grid(grid_for(nchunk), groups);
    if (dtype == PSCV_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const uint4*)dact, (const uint4*)y, scale, bias, ca, cb, cc, (uint4*)dy, nchunk, C / 8, relu, param_stride, coeff_stride);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, grid, dim3(256), 0, st, (const uint4*)dact, (const uint4*)y, scale, bias, ca, cb, cc, (uint4*)dy, nchunk, C / 8, relu, param_stride, coeff_stride);
    PSCV_CHECK_LAUNCH("pscv_bn_bwd_apply");
    return 0;
}
extern "C" int pscv_bn_bwd_apply(const void* dact, const void* y, int dtype, long nvox, int C, const float* scale,
                                 const float* bias, int relu, const float* ca, const float* cb, const float* cc, void* dy,
                                 void* stream) {
    return pscv_bn_bwd_apply_grouped(dact, y, dtype, nvox, 1, C, scale, bias, 0, relu, ca, cb, cc, 0, dy, stream);
}

extern "C" int pscv_softargmin_bwd(const float* logits, const float* depth, long depth_bstride, int depth_per_pixel,
                                   const float* grad_depth, const float* grad_index, const float* grad_entropy, void* dlogits8,
                                   int dtype, int B, int D, int h, int w, void* stream) {
    PSCV_CHECK_ARG(logits && dlogits8, "pscv_softargmin_bwd: null pointer argument");
    PSCV_CHECK_ARG(grad_depth || grad_index || grad_entropy, "pscv_softargmin_bwd: no upstream gradient given");
    PSCV_CHECK_ARG(!grad_depth || depth, "pscv_softargmin_bwd: grad_depth needs the depth planes");
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_softargmin_bwd: bad sizes");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_softargmin_bwd: dtype %d must be bf16 or fp16", dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long npix = (long)B * h * w;
    const int nb = (int)((npix + SB_PX - 1) / SB_PX);
    if (dtype == PSCV_BF16)
        hipLaunchKernelGGL(softargmin_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, logits, depth, depth_bstride, depth_per_pixel, grad_depth, grad_index, grad_entropy, (uint4*)dlogits8, B, D, h * w);
    else
        hipLaunchKernelGGL(softargmin_bwd_kernel<f16_t>, dim3(nb), dim3(256), 0, st, logits, depth, depth_bstride, depth_per_pixel, grad_depth, grad_index, grad_entropy, (uint4*)dlogits8, B, D, h * w);
    PSCV_CHECK_LAUNCH("pscv_softargmin_bwd");
    return 0;
}

extern "C" int pscv_leaky_relu_bwd(const void* dout, const void* out, int dtype, long nvox, int C, float slope, void* dpre, void* stream) {
    PSCV_CHECK_ARG(dout && out && dpre, "pscv_leaky_relu_bwd: null pointer argument");
    PSCV_CHECK_ARG(C % 8 == 0 && C > 0 && nvox > 0, "pscv_leaky_relu_bwd: bad sizes");
    PSCV_CHECK_ARG(slope >= 0.0f && slope <= 1.0f, "pscv_leaky_relu_bwd: slope %g outside [0,1]", (double)slope);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_leaky_relu_bwd: dtype %d must be bf16 or fp16", dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const int nb = grid_for(nchunk);
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const uint4*)dout, (const uint4*)out, (uint4*)dpre, nchunk, slope);
    else hipLaunchKernelGGL(relu_bwd_kernel<f16_t>, dim3(nb), dim3(256), 0, st, (const uint4*)dout, (const uint4*)out, (uint4*)dpre, nchunk, slope);
    PSCV_CHECK_LAUNCH("pscv_leaky_relu_bwd");
    return 0;
}
extern "C" int pscv_leaky_relu_bwd_sum(const void* dout, const void* out, int dtype, long nvox, int C, float slope, void* dpre,
                                       float* workspace, float* sums, void* stream) {
    if (bn_check("pscv_leaky_relu_bwd_sum", out, dtype, nvox, 1, C)) return -1;
    PSCV_CHECK_ARG(dout && dpre && workspace && sums, "pscv_leaky_relu_bwd_sum: null pointer argument");
    PSCV_CHECK_ARG(slope >= 0.0f && slope <= 1.0f, "pscv_leaky_relu_bwd_sum: slope %g outside [0,1]", (double)slope);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nchunk = nvox * (C / 8);
    const int nb = red_grid_for(nchunk);
    const uint4 *gp = (const uint4*)dout, *op = (const uint4*)out;
#define PSCV_RBS(HT)                                                                                                              \
    switch (C / 8) {                                                                                                              \
        case 1: hipLaunchKernelGGL((relu_bwd_sum_kernel<HT, 1>), dim3(nb), dim3(256), 0, st, gp, op, (uint4*)dpre, nchunk, slope, workspace, C); break; \
        case 2: hipLaunchKernelGGL((relu_bwd_sum_kernel<HT, 2>), dim3(nb), dim3(256), 0, st, gp, op, (uint4*)dpre, nchunk, slope, workspace, C); break; \
        case 4: hipLaunchKernelGGL((relu_bwd_sum_kernel<HT, 4>), dim3(nb), dim3(256), 0, st, gp, op, (uint4*)dpre, nchunk, slope, workspace, C); break; \
        case 8: hipLaunchKernelGGL((relu_bwd_sum_kernel<HT, 8>), dim3(nb), dim3(256), 0, st, gp, op, (uint4*)dpre, nchunk, slope, workspace, C); break; \
        default: hipLaunchKernelGGL((relu_bwd_sum_kernel<HT, 16>), dim3(nb), dim3(256), 0, st, gp, op, (uint4*)dpre, nchunk, slope, workspace, C); break; \
    }
    if (dtype == PSCV_BF16) { PSCV_RBS(bf16_t) } else { PSCV_RBS(f16_t) }
#undef PSCV_RBS
    PSCV_CHECK_LAUNCH("pscv_leaky_relu_bwd_sum");
    hipLaunchKernelGGL(finish_partials_kernel, dim3((2 * C + 15) / 16, 1), dim3(256), 0, st, workspace, nb, 2 * C, sums);
    PSCV_CHECK_LAUNCH("pscv_leaky_relu_bwd_sum(finish)");
    return 0;
}
extern "C" int pscv_relu_bwd(const void* dout, const void* out, int dtype, long nvox, int C, void* dpre, void* stream) {
    return pscv_leaky_relu_bwd(dout, out, dtype, nvox, C, 0.0f, dpre, stream);
}

extern "C" int pscv_bn_finalize_grouped(const float* sums, long nvox, int groups, int C, const float* gamma, const float* beta, float eps,
                                        float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                                        float* out, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(sums && out && nvox > 0 && C > 0 && C <= 1024 && groups >= 1, "pscv_bn_finalize: bad arguments");
    PSCV_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pscv_bn_finalize: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3((C + 63) / 64 * 64), 0, reinterpret_cast<hipStream_t>(stream), sums, (float)nvox,
                       gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, out, C, groups);
    PSCV_CHECK_LAUNCH("pscv_bn_finalize");
    return 0;
}
extern "C" int pscv_bn_finalize(const float* sums, long nvox, int C, const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, long long* num_batches_tracked, float* out, void* stream) {
    return pscv_bn_finalize_grouped(sums, nvox, 1, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, out, stream);
}

extern "C" int pscv_bn_bwd_coeffs_grouped(const float* sums, const float* mean, const float* invstd, int stat_stride, const float* gamma,
                                          long nvox, int groups, int C, float* out, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(sums && mean && invstd && out && nvox > 0 && C > 0 && C <= 1024 && groups >= 1, "pscv_bn_bwd_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(groups), dim3((C + 63) / 64 * 64), 0, reinterpret_cast<hipStream_t>(stream), sums, mean, invstd,
                       gamma, (float)nvox, out, C, stat_stride);
    PSCV_CHECK_LAUNCH("pscv_bn_bwd_coeffs");
    return 0;
}
extern "C" int pscv_bn_bwd_coeffs(const float* sums, const float* mean, const float* invstd, const float* gamma, long nvox, int C,
                                  float* out, void* stream) {
    return pscv_bn_bwd_coeffs_grouped(sums, mean, invstd, 0, gamma, nvox, 1, C, out, stream);
}
