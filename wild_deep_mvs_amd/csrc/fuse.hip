// Visibility-weighted fusion of the per-pair regularised volumes (Vis-MVSNet), one pass:
//     fused[b,d,y,x,c] = sum_v w_v[b,y,x] * interm_v[b,d,y,x,c] / sum_v w_v[b,y,x],   w_v = exp(-uncert_v)
// Replaces the running `fused_interm += interm * weight`, `weight_sum += weight` and the final division of
// models/VisMVSNet/model_cas.py:354-357,385-386 (mode 'soft'): every pair volume is read once, the fused volume is
// written once.  This is also the reduction point of the source-view shard: with the views
// spread over ranks each rank runs it on its own views with `normalise = 0` and the partial sums are all-reduced.
#include "pscv_common.h"

namespace pscv {

struct FuseArgs {
    const void* interm[PSCV_MAX_SRC];   // each [B,D,h,w,8], 16-bit
    const float* uncert[PSCV_MAX_SRC];  // each [B,h,w] fp32 (log-uncertainty s; weight = exp(-s))
    void* out;                          // [B,D,h,w,8] 16-bit (normalise) or fp32 partial sums (no normalise)
    float* wsum_out;                    // [B,h,w] fp32 or null
    int n_src, B, D, hw;
    int normalise;
};

template <typename H>
__global__ __launch_bounds__(256) void fuse_pairs_kernel(const FuseArgs a) {
    const long nvox = (long)a.B * a.D * a.hw;
    const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= nvox) return;
    const int b = (int)(vox / ((long)a.D * a.hw));
    const int pf = (int)(vox % a.hw);
    const long pix = (long)b * a.hw + pf;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float wsum = 0.f;
    for (int v = 0; v < a.n_src; ++v) {
        const float w = expf(-a.uncert[v][pix]);
        wsum += w;
        const f32x8 x = Elem<H>::load8(reinterpret_cast<const H*>(a.interm[v]) + vox * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(x.v[j], w, acc[j]);
    }
    if (a.normalise) {
        const float inv = 1.0f / wsum;
        f32x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j] * inv;
        Elem<H>::store8(reinterpret_cast<H*>(a.out) + vox * 8, o);
    } else {
        f32x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
        Elem<float>::store8(reinterpret_cast<float*>(a.out) + vox * 8, o);
    }
    if (a.wsum_out && vox / a.hw % a.D == 0) a.wsum_out[pix] = wsum;
}

// second half of a source-view-sharded fusion: out = partial_sum / weight_sum after the cross-rank all-reduce
template <typename H>
__global__ __launch_bounds__(256) void fuse_finish_kernel(const float* __restrict__ part, const float* __restrict__ wsum,
                                                         H* __restrict__ out, int B, int D, int hw) {
    const long nvox = (long)B * D * hw;
    const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= nvox) return;
    const int b = (int)(vox / ((long)D * hw));
    const float inv = 1.0f / wsum[(long)b * hw + (int)(vox % hw)];
    f32x8 x = Elem<float>::load8(part + vox * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) x.v[j] *= inv;
    Elem<H>::store8(out + vox * 8, x);
}

// Backward of the fusion (training): with fused = sum_v w_v I_v / W, W = sum_v w_v, w_v = exp(-u_v) and G the
// gradient of the fused volume,
//     d I_v = G w_v / W            d u_v = -w_v / W * sum_{d,c} G (I_v - fused)
// One lane per pixel walks the depth axis (coalesced 16-byte accesses across the wave), recomputes `fused` from the pair
// volumes and keeps the per-view sums in registers: every volume is read once, every d I_v written once.
struct FuseBwdArgs {
    const void* interm[PSCV_MAX_SRC];
    const float* uncert[PSCV_MAX_SRC];
    const void* g;                      // [B,D,h,w,8] 16-bit
    void* dinterm[PSCV_MAX_SRC];        // each [B,D,h,w,8] 16-bit
    float* duncert[PSCV_MAX_SRC];       // each [B,h,w] fp32
    int n_src, B, D, hw;
};

// 256 threads = 32 x-adjacent pixels x 8 depth slices: a thread walks D / 8 planes and the per-view sums of the slices meet in LDS at the
// end (fixed order).  One lane per pixel over all planes left the coarse stages (5 120 pixels x 64 planes) with 20 workgroups.
constexpr int FB_PX = 32, FB_NS = 8;
template <typename H>
__global__ __launch_bounds__(256) void fuse_pairs_bwd_kernel(const FuseBwdArgs a) {
    __shared__ float sh[PSCV_MAX_SRC][FB_NS][FB_PX];
    const int px = threadIdx.x % FB_PX, sl = threadIdx.x / FB_PX;
    const long npix = (long)a.B * a.hw;
    long p = (long)blockIdx.x * FB_PX + px;
    const bool active = p < npix;
    p = active ? p : npix - 1;
    const int b = (int)(p / a.hw);
    const int pf = (int)(p - (long)b * a.hw);
    float w[PSCV_MAX_SRC], acc[PSCV_MAX_SRC];
    float wsum = 0.f;
    for (int v = 0; v < a.n_src; ++v) { w[v] = expf(-a.uncert[v][p]); wsum += w[v]; acc[v] = 0.f; }
    const float inv = 1.0f / wsum;
    const int per = (a.D + FB_NS - 1) / FB_NS;
    const int d0 = sl * per, d1 = min(a.D, d0 + per);
    for (int d = d0; d < d1; ++d) {
        const long vox = ((long)b * a.D + d) * a.hw + pf;
        const f32x8 G = Elem<H>::load8(reinterpret_cast<const H*>(a.g) + vox * 8);
        float fused[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int v = 0; v < a.n_src; ++v) {
            const f32x8 x = Elem<H>::load8(reinterpret_cast<const H*>(a.interm[v]) + vox * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) fused[j] = fmaf(x.v[j], w[v], fused[j]);
        }
        float gf = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) gf = fmaf(G.v[j], fused[j] * inv, gf);
        for (int v = 0; v < a.n_src; ++v) {
            const f32x8 x = Elem<H>::load8(reinterpret_cast<const H*>(a.interm[v]) + vox * 8);   // (L1/L2 hit: read just above)
            float gi = 0.f;
            f32x8 o;
            const float s = w[v] * inv;
#pragma unroll
            for (int j = 0; j < 8; ++j) { gi = fmaf(G.v[j], x.v[j], gi); o.v[j] = G.v[j] * s; }
            acc[v] += gi - gf;
            if (active) Elem<H>::store8(reinterpret_cast<H*>(a.dinterm[v]) + vox * 8, o);
        }
    }
    for (int v = 0; v < a.n_src; ++v) sh[v][sl][px] = acc[v];
    __syncthreads();
    if (sl == 0 && active) {
        for (int v = 0; v < a.n_src; ++v) {
            float t = sh[v][0][px];
#pragma unroll
            for (int i = 1; i < FB_NS; ++i) t += sh[v][i][px];
            a.duncert[v][p] = -w[v] * inv * t;
        }
    }
}

// second half of a source-view-sharded variance: out = sum2 / N - mean^2 after the cross-rank all-reduce of the partial sums
template <typename H>
__global__ __launch_bounds__(256) void variance_finish_kernel(const float* __restrict__ s1, const float* __restrict__ s2, H* __restrict__ out,
                                                             long nchunk, float invN, float invN2, int cvp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nchunk) return;
    const f32x8 a = Elem<float>::load8(s1 + i * 8), b = Elem<float>::load8(s2 + i * 8);
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (cvp) { const float m = a.v[j] * invN; o.v[j] = b.v[j] * invN - m * m; }     // net.py:148
        else o.v[j] = b.v[j] * invN - (a.v[j] * a.v[j]) * invN2;                         // model.py:134
    }
    Elem<H>::store8(out + i * 8, o);
}

}  // namespace pscv

extern "C" int pscv_variance_finish(const float* sums, long n, int n_views, int cost, int dtype, void* out, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(sums && out, "pscv_variance_finish: null pointer argument");
    PSCV_CHECK_ARG(n > 0 && n % 8 == 0 && n_views >= 2, "pscv_variance_finish: bad sizes");
    PSCV_CHECK_ARG(cost == PSCV_COST_VARIANCE || cost == PSCV_COST_VARIANCE_CVP, "pscv_variance_finish: cost %d must be VARIANCE or VARIANCE_CVP", cost);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_variance_finish: dtype %d must be bf16 or fp16", dtype);
    const long nchunk = n / 8;
    const unsigned nblk = (unsigned)((nchunk + 255) / 256);
    const float invN = 1.0f / (float)n_views, invN2 = 1.0f / ((float)n_views * (float)n_views);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(variance_finish_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, sums, sums + n, reinterpret_cast<bf16_t*>(out), nchunk, invN, invN2, cost == PSCV_COST_VARIANCE_CVP);
    else hipLaunchKernelGGL(variance_finish_kernel<f16_t>, dim3(nblk), dim3(256), 0, st, sums, sums + n, reinterpret_cast<f16_t*>(out), nchunk, invN, invN2, cost == PSCV_COST_VARIANCE_CVP);
    PSCV_CHECK_LAUNCH("pscv_variance_finish");
    return 0;
}

extern "C" int pscv_fuse_pairs_bwd(const void* const* interm, const float* const* uncert, int n_src, int dtype, const void* grad_fused,
                                   void* const* dinterm, float* const* duncert, int B, int D, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(interm && uncert && grad_fused && dinterm && duncert, "pscv_fuse_pairs_bwd: null pointer argument");
    PSCV_CHECK_ARG(n_src >= 1 && n_src <= PSCV_MAX_SRC, "pscv_fuse_pairs_bwd: n_src=%d outside [1,%d]", n_src, PSCV_MAX_SRC);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_fuse_pairs_bwd: dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_fuse_pairs_bwd: bad sizes");
    FuseBwdArgs a;
    for (int i = 0; i < PSCV_MAX_SRC; ++i) {
        a.interm[i] = i < n_src ? interm[i] : nullptr; a.uncert[i] = i < n_src ? uncert[i] : nullptr;
        a.dinterm[i] = i < n_src ? dinterm[i] : nullptr; a.duncert[i] = i < n_src ? duncert[i] : nullptr;
    }
    for (int i = 0; i < n_src; ++i) PSCV_CHECK_ARG(interm[i] && uncert[i] && dinterm[i] && duncert[i], "pscv_fuse_pairs_bwd: null view %d", i);
    a.g = grad_fused; a.n_src = n_src; a.B = B; a.D = D; a.hw = h * w;
    const long npix = (long)B * h * w;
    const unsigned nblk = (unsigned)((npix + FB_PX - 1) / FB_PX);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(fuse_pairs_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(fuse_pairs_bwd_kernel<f16_t>, dim3(nblk), dim3(256), 0, st, a);
    PSCV_CHECK_LAUNCH("pscv_fuse_pairs_bwd");
    return 0;
}

extern "C" int pscv_fuse_finish(const float* partial, const float* wsum, int dtype, void* out, int B, int D, int h, int w,
                                void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(partial && wsum && out, "pscv_fuse_finish: null pointer argument");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_fuse_finish: dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_fuse_finish: bad sizes");
    const long nvox = (long)B * D * h * w;
    const unsigned nblk = (unsigned)((nvox + 255) / 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(fuse_finish_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, partial, wsum, reinterpret_cast<bf16_t*>(out), B, D, h * w);
    else hipLaunchKernelGGL(fuse_finish_kernel<f16_t>, dim3(nblk), dim3(256), 0, st, partial, wsum, reinterpret_cast<f16_t*>(out), B, D, h * w);
    PSCV_CHECK_LAUNCH("pscv_fuse_finish");
    return 0;
}

extern "C" int pscv_fuse_pairs(const void* const* interm, const float* const* uncert, int n_src, int dtype, void* out,
                               float* wsum_out, int normalise, int B, int D, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(interm && uncert && out, "pscv_fuse_pairs: null pointer argument");
    PSCV_CHECK_ARG(n_src >= 1 && n_src <= PSCV_MAX_SRC, "pscv_fuse_pairs: n_src=%d outside [1,%d]", n_src, PSCV_MAX_SRC);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_fuse_pairs: dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_fuse_pairs: bad sizes");
    FuseArgs a;
    for (int i = 0; i < PSCV_MAX_SRC; ++i) { a.interm[i] = i < n_src ? interm[i] : nullptr; a.uncert[i] = i < n_src ? uncert[i] : nullptr; }
    for (int i = 0; i < n_src; ++i) PSCV_CHECK_ARG(interm[i] && uncert[i], "pscv_fuse_pairs: null view %d", i);
    a.out = out; a.wsum_out = wsum_out; a.n_src = n_src; a.B = B; a.D = D; a.hw = h * w; a.normalise = normalise;
    const long nvox = (long)B * D * h * w;
    const unsigned nblk = (unsigned)((nvox + 255) / 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PSCV_BF16) hipLaunchKernelGGL(fuse_pairs_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(fuse_pairs_kernel<f16_t>, dim3(nblk), dim3(256), 0, st, a);
    PSCV_CHECK_LAUNCH("pscv_fuse_pairs");
    return 0;
}
