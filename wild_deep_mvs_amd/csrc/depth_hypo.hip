// CVP-MVSNet refinement hypotheses in eval mode, on the device with no host round trip (gfx950).
//
// Reference: calDepthHypo, models/CVP_MVSNet/models/modules.py:131-226 -- per batch item (a Python loop, fp64), the depth
// step that moves a pixel's projection into the FIRST source view by one pixel along its epipolar line, the MEDIAN of
// |step| over the valid pixels, then 8 hypothesis planes depth + k * median, k = -4..3.
// Three launches: (1) per-pixel |step| in fp64 -> 64-bit keys (bit patterns of non-negative doubles order like the
// values; invalid pixels get the all-ones key), (2) one workgroup per batch item finds the lower median by an 8-pass
// MSB-first radix select with LDS integer histograms (exact, order-independent), (3) the hypothesis planes.
#include "pscv_common.h"

namespace pscv {

constexpr int HC = 39;   // doubles per batch item: Kref^-1 [9], T = E_src E_ref^-1 rows 0..2 [12], K_src [9], A [9]

__device__ __forceinline__ void mat3(const double* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = m[0] * x + m[1] * y + m[2] * z;
    oy = m[3] * x + m[4] * y + m[5] * z;
    oz = m[6] * x + m[7] * y + m[8] * z;
}

__global__ __launch_bounds__(256) void hypo_keys_kernel(const float* __restrict__ depth, const double* __restrict__ cams,
                                                        unsigned long long* __restrict__ keys, int H, int W) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const double* c = cams + (long)b * HC;
    const double* Kinv = c; const double* T = c + 9; const double* Ks = c + 21; const double* A = c + 30;
    const double x = (double)(pix % W), y = (double)(pix / W);
    const double d1 = (double)depth[(long)b * H * W + pix];
    auto project = [&](double d, double& u, double& v, double& z) {
        double cx, cy, cz;
        mat3(Kinv, x * d, y * d, d, cx, cy, cz);                                   // K_ref^-1 (X d)
        const double sx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];               // into the source camera frame
        const double sy = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
        const double sz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
        double px, py, pz;
        mat3(Ks, sx, sy, sz, px, py, pz);
        u = px / pz; v = py / pz; z = pz;
    };
    double u1, v1, z1, u2, v2, z2;
    project(d1, u1, v1, z1);
    project(d1 + 1.0, u2, v2, z2);
    const double dx = u2 - u1, dy = v2 - v1;
    const double nrm = sqrt(dx * dx + dy * dy);                                    // (the homogeneous third component is 1 - 1 = 0)
    const double inv = 1.0 / fmax(nrm, 1e-8);
    const double u3 = u1 + dx * inv, v3 = v1 + dy * inv;                           // one pixel along the epipolar line
    double r0, r1, r2, c0, c1, c2;
    mat3(A, u1, v1, 1.0, r0, r1, r2);
    r1 *= z1; r2 *= z1;                                                            // rows 1..2 of z1 A x1
    mat3(A, u3, v3, 1.0, c0, c1, c2);
    // rows 1..2 of [X | A x3] (delta, .)^T = rows 1..2 of z1 A x1  ->  Cramer's rule for delta
    const double det = y * c2 - c1;
    const bool valid = nrm > 1e-8 && z1 > 1e-8 && z2 > 1e-8 && fabs(det) > 1e-8;
    unsigned long long key = ~0ull;
    if (valid) {
        const double delta = fabs((c2 * r1 - c1 * r2) / det);
        key = (delta == delta) ? (unsigned long long)__double_as_longlong(delta) : ~0ull;
    }
    keys[(long)b * H * W + pix] = key;
}

// one workgroup per batch item: lower median (rank (n-1)/2) of the keys != ~0 by radix select, 8 bits per pass
__global__ __launch_bounds__(1024) void hypo_median_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ fallback,
                                                           double* __restrict__ steps, int n) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ unsigned rank_s;
    __shared__ int empty_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* k = keys + (long)b * n;
    unsigned long long prefix = 0;
    unsigned rank = 0;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long v = k[i];
            if (v == ~0ull) continue;
            if (pass == 0 || (v >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(v >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            if (pass == 0) {
                unsigned total = 0;
                for (int j = 0; j < 256; ++j) total += hist[j];
                empty_s = total == 0;
                rank = total ? (total - 1) / 2 : 0;
            }
            unsigned acc = 0;
            int j = 0;
            for (; j < 255; ++j) {
                if (acc + hist[j] > rank) break;
                acc += hist[j];
            }
            prefix_s = prefix | ((unsigned long long)j << shift);
            rank_s = rank - acc;
        }
        __syncthreads();
        prefix = prefix_s;
        rank = rank_s;
        if (empty_s) break;
    }
    if (tid == 0) steps[b] = empty_s ? (double)fallback[b] : __longlong_as_double((long long)prefix);
}

__global__ __launch_bounds__(256) void hypo_planes_kernel(const float* __restrict__ depth, const double* __restrict__ steps,
                                                          float* __restrict__ hypos, int hw) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const double d = (double)depth[(long)b * hw + pix], s = steps[b];
#pragma unroll
    for (int k = 0; k < 8; ++k) hypos[((long)b * 8 + k) * hw + pix] = (float)(d + (double)(k - 4) * s);
}

}  // namespace pscv

using namespace pscv;

extern "C" int pscv_cvp_depth_hypos(const float* depth, const double* cams, const float* fallback, unsigned long long* keys,
                                    double* steps, float* hypos, int B, int H, int W, void* stream) {
    PSCV_CHECK_ARG(depth && cams && fallback && keys && steps && hypos, "pscv_cvp_depth_hypos: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && H > 0 && W > 0 && (long)H * W < (1L << 30), "pscv_cvp_depth_hypos: bad sizes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hw = H * W;
    hipLaunchKernelGGL(hypo_keys_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, st, depth, cams, keys, H, W);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(keys)");
    hipLaunchKernelGGL(hypo_median_kernel, dim3(B), dim3(1024), 0, st, keys, fallback, steps, hw);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(median)");
    hipLaunchKernelGGL(hypo_planes_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, st, depth, steps, hypos, hw);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(planes)");
    return 0;
}
