// Softmax over the depth axis fused with every expectation the three models take from it (gfx950).
//
// A workgroup owns 32 x-adjacent pixels; 8 lane groups each reduce one eighth of the depth planes (coalesced
// 128-byte rows), fp32 arithmetic:  pass 1 running max;  pass 2 sum exp, sum exp*depth, sum exp*index;  LDS
// log-sum-exp merge of the 8 slices;  optional pass 3 for the entropy / full probability volume; the
// confidence window re-reads only the 4-5 planes it needs.
// Also emits the per-shard (max, sum, sum*depth, sum*index) partials used by the depth-plane-sharded
// multi-GPU path (log-sum-exp merge = one tiny all-reduce).
//
// Replaces (fdarmon/wild_deep_mvs): F.softmax + depth_regression + photometric confidence
// models/MVSNet/model.py:207-215, models/MVSNet/module.py:174-178, models/CVP_MVSNet/models/net.py:161-162,
// 203-219; soft_argmin / entropy models/VisMVSNet/nn_utils.py:453-470.
#include "pscv_common.h"

namespace pscv {

struct SoftArgs {
    const void* logits;
    const float* depth;
    long depth_bstride;
    int depth_per_pixel;
    float *o_depth, *o_index, *o_conf, *o_entropy, *o_prob, *o_part;
    int conf_mode;
    float window;
    int index_offset;
    int B, D, h, w;
};

template <typename T> __device__ __forceinline__ float ldlogit(const T* p);
template <> __device__ __forceinline__ float ldlogit<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldlogit<bf16_t>(const bf16_t* p) { return bf16_to_f32(p->bits); }
template <> __device__ __forceinline__ float ldlogit<f16_t>(const f16_t* p) { return f16lo((uint32_t)p->bits); }

// 256 threads = PX pixels x NS depth slices: each thread reduces D/NS planes, the slices merge through LDS with
// the log-sum-exp rule (the same rule the multi-GPU depth-plane shard uses across ranks).
constexpr int SA_PX = 32, SA_NS = 8, SA_PMAX = 32;     // (other pixel x slice tilings were measured in round 5: none faster)

template <typename T>
__global__ __launch_bounds__(256) void softargmin_kernel(const SoftArgs a) {
    __shared__ float sh[4][SA_NS][SA_PX];
    const long hw = (long)a.h * a.w;
    const long npix = (long)a.B * hw;
    const int px = threadIdx.x % SA_PX, sl = threadIdx.x / SA_PX;
    long pix = (long)blockIdx.x * SA_PX + px;
    const bool active = pix < npix;
    pix = active ? pix : npix - 1;
    const int b = (int)(pix / hw);
    const long pf = pix - (long)b * hw;
    const T* lp = reinterpret_cast<const T*>(a.logits) + (long)b * a.D * hw + pf;
    const float* dp = a.depth ? a.depth + (long)b * a.depth_bstride + (a.depth_per_pixel ? pf : 0) : nullptr;
    const long dstep = a.depth_per_pixel ? hw : 1;
    const int per = (a.D + SA_NS - 1) / SA_NS;
    const int d0 = sl * per, d1 = min(a.D, d0 + per);

    // a slice of up to SA_PMAX planes is read ONCE into registers (all loads in flight together), then reduced;
    // longer slices fall back to two passes over memory
    float m = -INFINITY, se = 0.f, sd = 0.f, si = 0.f;
    if (per <= SA_PMAX) {
        float v[SA_PMAX];
#pragma unroll
        for (int k = 0; k < SA_PMAX; ++k) v[k] = (d0 + k < d1) ? ldlogit<T>(lp + (long)(d0 + k) * hw) : -INFINITY;
#pragma unroll
        for (int k = 0; k < SA_PMAX; ++k) m = fmaxf(m, v[k]);
#pragma unroll
        for (int k = 0; k < SA_PMAX; ++k) {
            if (d0 + k < d1) {
                const float e = expf(v[k] - m);
                se += e;
                if (dp) sd = fmaf(e, dp[(d0 + k) * dstep], sd);
                si = fmaf(e, (float)(d0 + k + a.index_offset), si);
            }
        }
    } else {
        for (int d = d0; d < d1; ++d) m = fmaxf(m, ldlogit<T>(lp + d * hw));
        for (int d = d0; d < d1; ++d) {
            const float e = expf(ldlogit<T>(lp + d * hw) - m);
            se += e;
            if (dp) sd = fmaf(e, dp[d * dstep], sd);
            si = fmaf(e, (float)(d + a.index_offset), si);
        }
    }
    sh[0][sl][px] = m; sh[1][sl][px] = se; sh[2][sl][px] = sd; sh[3][sl][px] = si;
    __syncthreads();
    // every thread merges all slices (cheap) so that each knows the global max / normaliser
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < SA_NS; ++k) M = fmaxf(M, sh[0][k][px]);
    float SE = 0.f, SD = 0.f, SI = 0.f;
#pragma unroll
    for (int k = 0; k < SA_NS; ++k) {
        const float mk = sh[0][k][px];
        const float f = mk > -INFINITY ? expf(mk - M) : 0.0f;   // empty slice (D < NS*per): contributes nothing
        SE = fmaf(sh[1][k][px], f, SE); SD = fmaf(sh[2][k][px], f, SD); SI = fmaf(sh[3][k][px], f, SI);
    }
    const float inv = 1.0f / SE;
    const float eidx = SI * inv;   // expected plane index            model.py:213, nn_utils.py:459
    if (sl == 0 && active) {
        if (a.o_depth) a.o_depth[pix] = SD * inv;
        if (a.o_index) a.o_index[pix] = eidx;
        if (a.o_part) {
            float* pp = a.o_part + (long)b * 4 * hw + pf;
            pp[0] = M; pp[hw] = SE; pp[2 * hw] = SD; pp[3 * hw] = SI;
        }
        if (a.o_conf) {
            float c = 0.f;
            const float lidx = eidx - (float)a.index_offset;   // index local to this logit block
            if (a.conf_mode == 0) {
                // planes i-1 .. i+2 around i = trunc(E[index]) (zero padded)        model.py:211-215
                const int i = (int)lidx;
                for (int k = -1; k <= 2; ++k) {
                    const int d = i + k;
                    if (d >= 0 && d < a.D) c += expf(ldlogit<T>(lp + d * hw) - M) * inv;
                }
            } else {
                // planes with |d - E[index]| <= window                              nn_utils.py:464-465
                int dlo = (int)ceilf(lidx - a.window), dhi = (int)floorf(lidx + a.window);
                dlo = dlo < 0 ? 0 : dlo;
                dhi = dhi > a.D - 1 ? a.D - 1 : dhi;
                for (int d = dlo; d <= dhi; ++d)
                    if (fabsf((float)d - lidx) <= a.window) c += expf(ldlogit<T>(lp + d * hw) - M) * inv;
            }
            a.o_conf[pix] = c;
        }
    }
    if (a.o_entropy || a.o_prob) {
        float ent = 0.f;
        float* pr = (a.o_prob && active) ? a.o_prob + (long)b * a.D * hw + pf : nullptr;
        for (int d = d0; d < d1; ++d) {
            const float p = expf(ldlogit<T>(lp + d * hw) - M) * inv;
            if (pr) pr[d * hw] = p;
            ent -= p * logf(fminf(fmaxf(p, 1e-9f), 1.0f));   // nn_utils.py:469-470
        }
        if (a.o_entropy) {
            __syncthreads();
            sh[0][sl][px] = ent;
            __syncthreads();
            if (sl == 0 && active) {
                float e = 0.f;
#pragma unroll
                for (int k = 0; k < SA_NS; ++k) e += sh[0][k][px];
                a.o_entropy[pix] = e;
            }
        }
    }
}

// Short depth axes (D <= 32: CVP's 8 refinement hypotheses, the 16 / 32-plane stages of Vis-MVSNet): ONE thread per pixel with the
// whole logit column in registers.  The kernel above gives a pixel 8 threads (one per depth slice) and merges them through LDS:
// at D = 8 that is one plane per thread, 256 threads for 32 pixels and two barriers -- 0.9 TB/s at 1024 x 1280 x 8.  Here
// consecutive lanes read consecutive pixels of every plane (coalesced), all D loads are in flight together, and the statistics
// are plain sequential sums over d (fp32; the order differs from the slice merge above in the last bits only).
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void softargmin_small_kernel(const SoftArgs a) {
    const long hw = (long)a.h * a.w;
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)a.B * hw) return;
    const int b = (int)(pix / hw);
    const long pf = pix - (long)b * hw;
    const T* lp = reinterpret_cast<const T*>(a.logits) + (long)b * a.D * hw + pf;
    const float* dp = a.depth ? a.depth + (long)b * a.depth_bstride + (a.depth_per_pixel ? pf : 0) : nullptr;
    const long dstep = a.depth_per_pixel ? hw : 1;
    float v[DMAX], dv[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) v[d] = d < a.D ? ldlogit<T>(lp + (long)d * hw) : -INFINITY;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) dv[d] = (dp && d < a.D) ? dp[d * dstep] : 0.0f;
    float M = -INFINITY;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) M = fmaxf(M, v[d]);
    float SE = 0.f, SD = 0.f, SI = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        v[d] = d < a.D ? expf(v[d] - M) : 0.0f;              // from here on: e_d
        SE += v[d];
        SD = fmaf(v[d], dv[d], SD);
        SI = fmaf(v[d], (float)(d + a.index_offset), SI);
    }
    const float inv = 1.0f / SE;
    const float eidx = SI * inv;
    if (a.o_depth) a.o_depth[pix] = SD * inv;
    if (a.o_index) a.o_index[pix] = eidx;
    if (a.o_conf) {
        const float lidx = eidx - (float)a.index_offset;
        float c = 0.f;
        if (a.conf_mode == 0) {
            const int i = (int)lidx;                          // planes i-1 .. i+2 (zero padded)          model.py:211-215
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                if (d >= i - 1 && d <= i + 2) c += v[d] * inv;
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; ++d)                    // |d - E[index]| <= window                  nn_utils.py:464-465
                if (fabsf((float)d - lidx) <= a.window) c += v[d] * inv;
        }
        a.o_conf[pix] = c;
    }
    if (a.o_entropy) {
        float ent = 0.f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const float p = v[d] * inv;
            if (d < a.D) ent -= p * logf(fminf(fmaxf(p, 1e-9f), 1.0f));   // nn_utils.py:469-470
        }
        a.o_entropy[pix] = ent;
    }
}

// Window probability of ONE depth shard under the globally merged softmax statistics (depth-plane shard across GPUs):
//     out[pix] = sum over this shard's planes d with |d + index_offset - E[index]| <= window of exp(l_d - M) / Z
// stats [B,3,h,w] = (global max M, global sum Z, global expected index); the shards' outputs are summed by an all-reduce.
__global__ __launch_bounds__(256) void softargmin_window_kernel(const float* __restrict__ logits, const float* __restrict__ stats,
                                                                float* __restrict__ out, float window, int index_offset, int B, int D,
                                                                long hw) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)B * hw) return;
    const int b = (int)(p / hw);
    const long pf = p - (long)b * hw;
    const float* sp = stats + (long)b * 3 * hw + pf;
    const float M = sp[0], inv = 1.0f / sp[hw], eidx = sp[2 * hw];
    const float* lp = logits + (long)b * D * hw + pf;
    float c = 0.f;
    for (int d = 0; d < D; ++d)
        if (fabsf((float)(d + index_offset) - eidx) <= window) c += expf(lp[(long)d * hw] - M) * inv;
    out[p] = c;
}

}  // namespace pscv

extern "C" int pscv_softargmin_window(const float* logits, const float* stats, float* out, float window, int index_offset, int B,
                                      int D, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(logits && stats && out, "pscv_softargmin_window: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_softargmin_window: bad sizes");
    const long npix = (long)B * h * w;
    hipLaunchKernelGGL(softargmin_window_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       logits, stats, out, window, index_offset, B, D, (long)h * w);
    PSCV_CHECK_LAUNCH("pscv_softargmin_window");
    return 0;
}

pscv::Knob g_softargmin_small = {1, pscv::KNOB_SPARE4};   // pscv_set_tuning("softargmin_small", 0): short depth axes back on the slice kernel

extern "C" int pscv_softargmin(const void* logits, int logit_dtype, const float* depth, long depth_bstride,
                               int depth_per_pixel, float* out_depth, float* out_index, float* out_conf,
                               float* out_entropy, float* out_prob, float* out_partials, int conf_mode, float window,
                               int index_offset, int B, int D, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(logits, "pscv_softargmin: logits is null");
    PSCV_CHECK_ARG(B > 0 && D > 0 && h > 0 && w > 0, "pscv_softargmin: bad sizes");
    PSCV_CHECK_ARG(depth || !out_depth, "pscv_softargmin: out_depth requested without depth planes");
    PSCV_CHECK_ARG(conf_mode == 0 || conf_mode == 1, "pscv_softargmin: conf_mode %d", conf_mode);
    SoftArgs a;
    a.logits = logits; a.depth = depth; a.depth_bstride = depth_bstride; a.depth_per_pixel = depth_per_pixel;
    a.o_depth = out_depth; a.o_index = out_index; a.o_conf = out_conf; a.o_entropy = out_entropy;
    a.o_prob = out_prob; a.o_part = out_partials;
    a.conf_mode = conf_mode; a.window = window; a.index_offset = index_offset;
    a.B = B; a.D = D; a.h = h; a.w = w;
    const long npix = (long)B * h * w;
    const unsigned nblk = (unsigned)((npix + pscv::SA_PX - 1) / pscv::SA_PX);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (D <= 32 && !out_prob && !out_partials && logit_dtype == PSCV_F32 && g_softargmin_small) {
        const unsigned nb = (unsigned)((npix + 255) / 256);
        if (D <= 8) hipLaunchKernelGGL((softargmin_small_kernel<float, 8>), dim3(nb), dim3(256), 0, st, a);
        else if (D <= 16) hipLaunchKernelGGL((softargmin_small_kernel<float, 16>), dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((softargmin_small_kernel<float, 32>), dim3(nb), dim3(256), 0, st, a);
        PSCV_CHECK_LAUNCH("pscv_softargmin(small)");
        return 0;
    }
    if (logit_dtype == PSCV_F32) hipLaunchKernelGGL(softargmin_kernel<float>, dim3(nblk), dim3(256), 0, st, a);
    else if (logit_dtype == PSCV_BF16) hipLaunchKernelGGL(softargmin_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, a);
    else if (logit_dtype == PSCV_F16) hipLaunchKernelGGL(softargmin_kernel<f16_t>, dim3(nblk), dim3(256), 0, st, a);
    else { set_error("pscv_softargmin: bad logit dtype %d", logit_dtype); return -1; }
    PSCV_CHECK_LAUNCH("pscv_softargmin");
    return 0;
}
