// Camera blocks for the warp kernel, computed on the device in one launch (fp64 inside, fp32 out) so that the
// hot path never waits on dozens of tiny host-launched tensor ops.
//
// PROJ geometry (reference models/MVSNet/module.py:128-130, models/CVP_MVSNet/models/modules.py:89-98):
//   proj = P_src * P_ref^-1 with P = [[A, b], [0 0 0 1]]  =>  rot = A_s A_r^-1,  trans = b_s - rot b_r.
#include "pscv_common.h"

namespace pscv {

__device__ __forceinline__ void inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double c0 = e * i - f * h, c3 = f * g - d * i, c6 = d * h - e * g;
    const double inv_det = 1.0 / (a * c0 + b * c3 + c * c6);
    o[0] = c0 * inv_det; o[1] = (c * h - b * i) * inv_det; o[2] = (b * f - c * e) * inv_det;
    o[3] = c3 * inv_det; o[4] = (a * i - c * g) * inv_det; o[5] = (c * d - a * f) * inv_det;
    o[6] = c6 * inv_det; o[7] = (b * g - a * h) * inv_det; o[8] = (a * e - b * d) * inv_det;
}

__global__ void proj_cams_kernel(const float* __restrict__ proj, int B, int V, int ref, float* __restrict__ cams) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (V - 1)) return;
    const int b = t / (V - 1), j = t % (V - 1);
    const int v = j < ref ? j : j + 1;
    const float* Pr = proj + ((long)b * V + ref) * 16;
    const float* Ps = proj + ((long)b * V + v) * 16;
    double Ar[9], Ai[9], As[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { Ar[r * 3 + c] = Pr[r * 4 + c]; As[r * 3 + c] = Ps[r * 4 + c]; }
    inv3(Ar, Ai);
    float* o = cams + ((long)j * B + b) * PSCV_CAM_FLOATS;
    double rot[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            rot[r * 3 + c] = As[r * 3] * Ai[c] + As[r * 3 + 1] * Ai[3 + c] + As[r * 3 + 2] * Ai[6 + c];
            o[r * 3 + c] = (float)rot[r * 3 + c];
        }
    for (int r = 0; r < 3; ++r) {
        const double tr = (double)Ps[r * 4 + 3] - (rot[r * 3] * Pr[3] + rot[r * 3 + 1] * Pr[7] + rot[r * 3 + 2] * Pr[11]);
        o[9 + r] = (float)tr;
    }
    for (int k = 12; k < PSCV_CAM_FLOATS; ++k) o[k] = 0.0f;
}

}  // namespace pscv

extern "C" int pscv_proj_cams(const float* proj, int B, int V, int reference_frame, float* cams, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(proj && cams, "pscv_proj_cams: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && V >= 2 && reference_frame >= 0 && reference_frame < V, "pscv_proj_cams: bad sizes B=%d V=%d ref=%d", B, V, reference_frame);
    const int n = B * (V - 1);
    hipLaunchKernelGGL(proj_cams_kernel, dim3((n + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), proj, B, V,
                       reference_frame, cams);
    PSCV_CHECK_LAUNCH("pscv_proj_cams");
    return 0;
}
