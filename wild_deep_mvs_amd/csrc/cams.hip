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

// ---- HOMOG geometry (Vis-MVSNet) ------------------------------------------------------------------
// hom(d) = A p - Bm p / (d + 1e-9) with A = K_s R_s R_r^T K_r^-1 and Bm = K_s R_s (c_s - c_r) n_r^T R_r^T K_r^-1,
// n_r = third row of R_r, c = -R^T t  (reference models/VisMVSNet/homography.py:23-74); the intrinsics are first
// scaled like scale_camera(cam, scale) (models/VisMVSNet/preproc.py:63-92, call model_cas.py:177).
namespace pscv {

__device__ __forceinline__ void mat3mul(const double* a, const double* b, double* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}

__global__ void homog_cams_kernel(const float* __restrict__ ref_cam, const float* __restrict__ src_cams, int B, int n_src,
                                  float scale, float* __restrict__ cams) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_src) return;
    const int b = t / n_src, j = t % n_src;
    const float* L = ref_cam + (long)b * 32;                         // [2,4,4]: [R|t], [K; ...]
    const float* Rr = src_cams + ((long)j * B + b) * 32;
    double Rl[9], Rs[9], Kl[9], Ks[9], tl[3], ts[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            Rl[r * 3 + c] = L[r * 4 + c]; Rs[r * 3 + c] = Rr[r * 4 + c];
            Kl[r * 3 + c] = L[16 + r * 4 + c]; Ks[r * 3 + c] = Rr[16 + r * 4 + c];
        }
        tl[r] = L[r * 4 + 3]; ts[r] = Rr[r * 4 + 3];
    }
    const double s = scale;
    Kl[0] *= s; Kl[4] *= s; Kl[2] *= s; Kl[5] *= s;
    Ks[0] *= s; Ks[4] *= s; Ks[2] *= s; Ks[5] *= s;
    double Kli[9], RlT[9], M1[9], KR[9];
    inv3(Kl, Kli);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) RlT[r * 3 + c] = Rl[c * 3 + r];
    mat3mul(RlT, Kli, M1);          // R_r^T K_r^-1
    mat3mul(Ks, Rs, KR);            // K_s R_s
    double cl[3], cs[3], crel[3];
    for (int r = 0; r < 3; ++r) {
        cl[r] = -(Rl[0 * 3 + r] * tl[0] + Rl[1 * 3 + r] * tl[1] + Rl[2 * 3 + r] * tl[2]);
        cs[r] = -(Rs[0 * 3 + r] * ts[0] + Rs[1 * 3 + r] * ts[1] + Rs[2 * 3 + r] * ts[2]);
        crel[r] = cs[r] - cl[r];
    }
    double T[9], TM[9], A[9], Bm[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[r * 3 + c] = crel[r] * Rl[2 * 3 + c];   // c_rel n_r^T
    mat3mul(T, M1, TM);
    mat3mul(KR, M1, A);
    mat3mul(KR, TM, Bm);
    float* o = cams + ((long)j * B + b) * PSCV_CAM_FLOATS;
    for (int k = 0; k < 9; ++k) { o[k] = (float)A[k]; o[9 + k] = (float)Bm[k]; }
}

}  // namespace pscv

extern "C" int pscv_homog_cams(const float* ref_cam, const float* src_cams, int B, int n_src, float scale, float* cams,
                               void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(ref_cam && src_cams && cams, "pscv_homog_cams: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && n_src > 0 && scale > 0.f, "pscv_homog_cams: bad sizes B=%d n_src=%d scale=%g", B, n_src, (double)scale);
    const int n = B * n_src;
    hipLaunchKernelGGL(homog_cams_kernel, dim3((n + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ref_cam,
                       src_cams, B, n_src, scale, cams);
    PSCV_CHECK_LAUNCH("pscv_homog_cams");
    return 0;
}

// ---- CVP-MVSNet: every camera block of a forward pass in one launch ----------------------------------------------------
// Per pyramid level l (intrinsics rows 0-1 scaled by 1 / level_scale[l] in fp32 like conditionIntrinsics, reference
// models/CVP_MVSNet/models/modules.py:31-50) and batch item:
//   warp  [L][N][B][18] : rot | trans of P_src P_ref^-1 with P = [[K E[:3]], [0 0 0 1]] built in fp32 (modules.py:89-98), the block
//                         pscv_proj_cams makes from the stacked projections;
//   hypo  [L][B][39] f64: K_ref^-1, rows 0..2 of E_src0 E_ref^-1, K_src0, (K_ref R_ref)(K_src0 R_src0)^-1 -- the constants of the
//                         one-pixel depth step of calDepthHypo (modules.py:131-226), first source view.
// The tensor-level path spent ~700 launches of 4 us on these 3x3 products per forward (a quarter of configuration 4).
namespace pscv {

struct CvpCamArgs {
    const float *ref_in, *src_in, *ref_ex, *src_ex;
    float inv_scale[8];
    int B, N, L;
    float* warp;
    double* hypo;
};

__device__ __forceinline__ void cvp_level_k(const float* K, float inv_s, float* o) {
    for (int c = 0; c < 3; ++c) { o[c] = K[c] * inv_s; o[3 + c] = K[3 + c] * inv_s; o[6 + c] = K[6 + c]; }
}
// rows 0..2 of [[K E[:3]], [0 0 0 1]] in fp32
__device__ __forceinline__ void cvp_projection(const float* K, const float* E, float* P) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) P[r * 4 + c] = fmaf(K[r * 3 + 2], E[8 + c], fmaf(K[r * 3 + 1], E[4 + c], K[r * 3] * E[c]));
}

__global__ void cvp_cams_kernel(const CvpCamArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.L * a.B * a.N) return;
    const int j = t % a.N, b = (t / a.N) % a.B, l = t / (a.N * a.B);
    const float inv_s = a.inv_scale[l];
    const float* Er = a.ref_ex + (long)b * 16;
    const float* Es = a.src_ex + ((long)b * a.N + j) * 16;
    float Kr[9], Ks[9], Pr[12], Ps[12];
    cvp_level_k(a.ref_in + (long)b * 9, inv_s, Kr);
    cvp_level_k(a.src_in + ((long)b * a.N + j) * 9, inv_s, Ks);
    cvp_projection(Kr, Er, Pr);
    cvp_projection(Ks, Es, Ps);
    {   // warp block (same arithmetic as proj_cams_kernel)
        double Ar[9], Ai[9], As[9], rot[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) { Ar[r * 3 + c] = Pr[r * 4 + c]; As[r * 3 + c] = Ps[r * 4 + c]; }
        inv3(Ar, Ai);
        float* o = a.warp + (((long)l * a.N + j) * a.B + b) * PSCV_CAM_FLOATS;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                rot[r * 3 + c] = As[r * 3] * Ai[c] + As[r * 3 + 1] * Ai[3 + c] + As[r * 3 + 2] * Ai[6 + c];
                o[r * 3 + c] = (float)rot[r * 3 + c];
            }
        for (int r = 0; r < 3; ++r)
            o[9 + r] = (float)((double)Ps[r * 4 + 3] - (rot[r * 3] * Pr[3] + rot[r * 3 + 1] * Pr[7] + rot[r * 3 + 2] * Pr[11]));
        for (int k = 12; k < PSCV_CAM_FLOATS; ++k) o[k] = 0.0f;
    }
    if (j == 0 && a.hypo) {
        double Ki[9], Kd[9], Ksd[9], Ri[9], Rs[9], Rii[9], ti[3], ts[3];
        for (int k = 0; k < 9; ++k) { Kd[k] = Kr[k]; Ksd[k] = Ks[k]; }
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) { Ri[r * 3 + c] = Er[r * 4 + c]; Rs[r * 3 + c] = Es[r * 4 + c]; }
            ti[r] = Er[r * 4 + 3]; ts[r] = Es[r * 4 + 3];
        }
        inv3(Kd, Ki);
        inv3(Ri, Rii);
        double* o = a.hypo + ((long)l * a.B + b) * 39;
        for (int k = 0; k < 9; ++k) o[k] = Ki[k];
        // E_src E_ref^-1 with E_ref^-1 = [[R^-1, -R^-1 t], [0 0 0 1]]
        double ti_inv[3];
        for (int r = 0; r < 3; ++r) ti_inv[r] = -(Rii[r * 3] * ti[0] + Rii[r * 3 + 1] * ti[1] + Rii[r * 3 + 2] * ti[2]);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) o[9 + r * 4 + c] = Rs[r * 3] * Rii[c] + Rs[r * 3 + 1] * Rii[3 + c] + Rs[r * 3 + 2] * Rii[6 + c];
            o[9 + r * 4 + 3] = Rs[r * 3] * ti_inv[0] + Rs[r * 3 + 1] * ti_inv[1] + Rs[r * 3 + 2] * ti_inv[2] + ts[r];
        }
        for (int k = 0; k < 9; ++k) o[21 + k] = Ksd[k];
        double KR[9], KsRs[9], KsRsi[9], A[9];
        mat3mul(Kd, Ri, KR);
        mat3mul(Ksd, Rs, KsRs);
        inv3(KsRs, KsRsi);
        mat3mul(KR, KsRsi, A);
        for (int k = 0; k < 9; ++k) o[30 + k] = A[k];
    }
}

}  // namespace pscv

extern "C" int pscv_cvp_cams(const float* ref_in, const float* src_in, const float* ref_ex, const float* src_ex,
                             const float* level_scale, int B, int N, int L, float* warp_cams, double* hypo_cams, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(ref_in && src_in && ref_ex && src_ex && level_scale && warp_cams, "pscv_cvp_cams: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && N > 0 && L > 0 && L <= 8, "pscv_cvp_cams: bad sizes B=%d N=%d L=%d (at most 8 levels)", B, N, L);
    CvpCamArgs a;
    a.ref_in = ref_in; a.src_in = src_in; a.ref_ex = ref_ex; a.src_ex = src_ex;
    for (int l = 0; l < 8; ++l) a.inv_scale[l] = 1.0f;
    for (int l = 0; l < L; ++l) {
        PSCV_CHECK_ARG(level_scale[l] > 0.f, "pscv_cvp_cams: level_scale[%d] = %g", l, (double)level_scale[l]);
        a.inv_scale[l] = 1.0f / level_scale[l];     // (the tensor division by a Python scalar multiplies by the fp32 reciprocal)
    }
    a.B = B; a.N = N; a.L = L; a.warp = warp_cams; a.hypo = hypo_cams;
    const int n = L * B * N;
    hipLaunchKernelGGL(cvp_cams_kernel, dim3((n + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a);
    PSCV_CHECK_LAUNCH("pscv_cvp_cams");
    return 0;
}
