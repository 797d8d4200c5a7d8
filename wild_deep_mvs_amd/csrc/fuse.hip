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

}  // namespace pscv

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
