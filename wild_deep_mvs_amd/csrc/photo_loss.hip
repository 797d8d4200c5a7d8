// Unsupervised photometric loss of models/trainer.py:209-278: depth map -> flows -> warped source
// images (+ d warped / d depth), and the 11x11 Gaussian SSIM of utils/ssimLoss.py:27-60 (+ its gradient to the second image).
//
// HBM-bound, tiny next to the cost volume (3-channel images at the depth map's resolution): one thread per reference pixel for
// the warp (its backward needs no atomics -- the gradient flows to the DEPTH through the sampling position, every pixel owns its
// own depth), LDS-tiled separable blur for SSIM.
#include "pscv_common.h"

namespace pscv {

struct FlowGeom {
    float gx, gy;     // normalised grid coordinate after the z <= 0 rule and the +-10 clamp     trainer.py:212-217
    float z;          // depth in the source view                                                  utils_3D.py:204
    float dgx, dgy;   // d grid / d depth (0 where the assignment / clamp cuts the graph)
};

// utils_3D.py:201-206 (flows_from_single_depthmap) + normalize :243-272 + trainer.py:212-217, fp32 in the reference's order:
//   P3 = inv(P_ref) (x d, y d, d, 1);  q = P_src P3;  f = q_xy / max(q_z, 1e-6);  g = 2 f / (size - 1) - 1
__device__ __forceinline__ FlowGeom flow_geom(const float* __restrict__ inv_ref, const float* __restrict__ proj, float x, float y,
                                              float d, int h, int w, bool want_grad) {
    const float p[4] = {x * d, y * d, d, 1.0f};
    float P3[4], q[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) P3[j] = p[0] * inv_ref[j * 4 + 0] + p[1] * inv_ref[j * 4 + 1] + p[2] * inv_ref[j * 4 + 2] + p[3] * inv_ref[j * 4 + 3];
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = P3[0] * proj[j * 4 + 0] + P3[1] * proj[j * 4 + 1] + P3[2] * proj[j * 4 + 2] + P3[3] * proj[j * 4 + 3];
    FlowGeom g;
    g.z = q[2];
    const float zc = fmaxf(q[2], 1e-6f);
    const float fx = q[0] / zc, fy = q[1] / zc;
    float gx = 2.0f * fx / (float)(w - 1) - 1.0f, gy = 2.0f * fy / (float)(h - 1) - 1.0f;
    const bool behind = q[2] <= 0.0f;
    if (behind) gx = gy = -10.0f;
    g.gx = fminf(fmaxf(gx, -10.0f), 10.0f);
    g.gy = fminf(fmaxf(gy, -10.0f), 10.0f);
    g.dgx = g.dgy = 0.0f;
    if (want_grad && !behind) {
        const float dp[3] = {x, y, 1.0f};
        float dP3[4], dq[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) dP3[j] = dp[0] * inv_ref[j * 4 + 0] + dp[1] * inv_ref[j * 4 + 1] + dp[2] * inv_ref[j * 4 + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) dq[j] = dP3[0] * proj[j * 4 + 0] + dP3[1] * proj[j * 4 + 1] + dP3[2] * proj[j * 4 + 2] + dP3[3] * proj[j * 4 + 3];
        float dfx, dfy;
        if (q[2] > 1e-6f) { dfx = (dq[0] - fx * dq[2]) / zc; dfy = (dq[1] - fy * dq[2]) / zc; }
        else { dfx = dq[0] / zc; dfy = dq[1] / zc; }                       // clamp(min=1e-6) cuts d/dz
        if (gx >= -10.0f && gx <= 10.0f) g.dgx = 2.0f * dfx / (float)(w - 1);
        if (gy >= -10.0f && gy <= 10.0f) g.dgy = 2.0f * dfy / (float)(h - 1);
    }
    return g;
}

struct Taps {
    int x0, y0;
    float ax, ay;                 // fractional parts
    bool in00, in01, in10, in11;  // (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1) inside the image
};
// F.grid_sample(bilinear, zeros, align_corners=False): index = ((g + 1) size - 1) / 2
__device__ __forceinline__ Taps grid_taps(float gx, float gy, int hs, int ws) {
    const float ix = ((gx + 1.0f) * (float)ws - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)hs - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.ax = ix - fx; t.ay = iy - fy;
    const bool xa = t.x0 >= 0 && t.x0 < ws, xb = t.x0 + 1 >= 0 && t.x0 + 1 < ws;
    const bool ya = t.y0 >= 0 && t.y0 < hs, yb = t.y0 + 1 >= 0 && t.y0 + 1 < hs;
    t.in00 = ya && xa; t.in01 = ya && xb; t.in10 = yb && xa; t.in11 = yb && xb;
    return t;
}

// one thread per (b, s, y, x).  src [B,S,C,h,w]; depth [B,h,w]; inv_ref [B,16]; proj [B,S,16];
// optional second sampled plane src_depth [B,S,h,w] -> warped_depth [B,S,h,w] (occlusion masking, trainer.py:264)
__global__ __launch_bounds__(256) void photo_warp_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                         const float* __restrict__ inv_ref, const float* __restrict__ proj,
                                                         const float* __restrict__ src_depth, float* __restrict__ warped,
                                                         float* __restrict__ mask, float* __restrict__ z_src,
                                                         float* __restrict__ flows, float* __restrict__ warped_depth, int B, int S,
                                                         int C, int h, int w) {
    const long hw = (long)h * w;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * S * hw) return;
    const long bs = i / hw, pix = i - bs * hw;
    const int b = (int)(bs / S);
    const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    const FlowGeom g = flow_geom(inv_ref + b * 16, proj + bs * 16, (float)x, (float)y, depth[(long)b * hw + pix], h, w, false);
    if (mask) mask[i] = (g.gx < 1.0f && g.gy < 1.0f && g.gx > -1.0f && g.gy > -1.0f) ? 1.0f : 0.0f;     // trainer.py:227
    if (z_src) z_src[i] = g.z;
    if (flows) { flows[2 * i] = g.gx; flows[2 * i + 1] = g.gy; }
    const Taps t = grid_taps(g.gx, g.gy, h, w);
    const float w00 = (1.f - t.ax) * (1.f - t.ay), w01 = t.ax * (1.f - t.ay), w10 = (1.f - t.ax) * t.ay, w11 = t.ax * t.ay;
    const long o00 = (long)t.y0 * w + t.x0;
    if (warped)
        for (int c = 0; c < C; ++c) {
            const float* sp = src + (bs * C + c) * hw;
            float v = 0.f;
            if (t.in00) v += sp[o00] * w00;
            if (t.in01) v += sp[o00 + 1] * w01;
            if (t.in10) v += sp[o00 + w] * w10;
            if (t.in11) v += sp[o00 + w + 1] * w11;
            warped[(bs * C + c) * hw + pix] = v;
        }
    if (warped_depth) {
        const float* sp = src_depth + bs * hw;
        float v = 0.f;
        if (t.in00) v += sp[o00] * w00;
        if (t.in01) v += sp[o00 + 1] * w01;
        if (t.in10) v += sp[o00 + w] * w10;
        if (t.in11) v += sp[o00 + w + 1] * w11;
        warped_depth[i] = v;
    }
}

// one thread per reference pixel (b, y, x): grad_depth = sum_s sum_c gw * (dV/dix dix/dd + dV/diy diy/dd)
__global__ __launch_bounds__(256) void photo_warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                             const float* __restrict__ inv_ref, const float* __restrict__ proj,
                                                             const float* __restrict__ grad_warped, float* __restrict__ grad_depth,
                                                             int B, int S, int C, int h, int w) {
    const long hw = (long)h * w;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * hw) return;
    const int b = (int)(i / hw);
    const long pix = i - (long)b * hw;
    const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
    const float d = depth[i];
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        const long bs = (long)b * S + s;
        const FlowGeom g = flow_geom(inv_ref + b * 16, proj + bs * 16, (float)x, (float)y, d, h, w, true);
        if (g.dgx == 0.0f && g.dgy == 0.0f) continue;
        const Taps t = grid_taps(g.gx, g.gy, h, w);
        const long o00 = (long)t.y0 * w + t.x0;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* sp = src + (bs * C + c) * hw;
            const float v00 = t.in00 ? sp[o00] : 0.f, v01 = t.in01 ? sp[o00 + 1] : 0.f;
            const float v10 = t.in10 ? sp[o00 + w] : 0.f, v11 = t.in11 ? sp[o00 + w + 1] : 0.f;
            const float gw = grad_warped[(bs * C + c) * hw + pix];
            gix += gw * ((v01 - v00) * (1.f - t.ay) + (v11 - v10) * t.ay);
            giy += gw * ((v10 - v00) * (1.f - t.ax) + (v11 - v01) * t.ax);
        }
        acc += gix * (0.5f * (float)w) * g.dgx + giy * (0.5f * (float)h) * g.dgy;
    }
    grad_depth[i] = acc;
}

// ---- SSIM (utils/ssimLoss.py) ---------------------------------------------------------------------------------------------
constexpr int SS_R = 5;            // window 11
constexpr int SS_TW = 32, SS_TH = 8;
constexpr int SS_LW = SS_TW + 2 * SS_R, SS_LH = SS_TH + 2 * SS_R;

struct GaussWin { float g[2 * SS_R + 1]; };

// NQ planes staged with halo in in_[NQ][SS_LH][SS_LW] -> each thread's NQ blurred values at (ty, tx); zero padding is in the
// staged data.  hb_[NQ][SS_LH][SS_TW] is scratch.
template <int NQ>
__device__ __forceinline__ void blur_tile(float (*in_)[SS_LH][SS_LW], float (*hb_)[SS_LH][SS_TW], const GaussWin& gw, int tid,
                                          float* out) {
    for (int e = tid; e < NQ * SS_LH * SS_TW; e += 256) {
        const int q = e / (SS_LH * SS_TW), r = (e / SS_TW) % SS_LH, c = e % SS_TW;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * SS_R; ++k) s = fmaf(gw.g[k], in_[q][r][c + k], s);
        hb_[q][r][c] = s;
    }
    __syncthreads();
    const int ty = tid / SS_TW, tx = tid % SS_TW;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * SS_R; ++k) s = fmaf(gw.g[k], hb_[q][ty + k][tx], s);
        out[q] = s;
    }
}

struct SsimTerms { float S, dmu2, de22, de12; };
__device__ __forceinline__ SsimTerms ssim_terms(const float* m) {
    const float mu1 = m[0], mu2 = m[1];
    const float s1 = m[2] - mu1 * mu1, s2 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;      // ssimLoss.py:31-37
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
    SsimTerms t;
    const float inv = 1.0f / (B1 * B2);
    t.S = A1 * A2 * inv;                                                                 // ssimLoss.py:42
    t.dmu2 = 2.f * mu1 * (A2 - A1) * inv - 2.f * mu2 * t.S / B1 + 2.f * mu2 * t.S / B2;
    t.de22 = -t.S / B2;
    t.de12 = 2.f * A1 * inv;
    return t;
}

// grid: (tiles_x, tiles_y, N*C) with N = n1 * rep second-image items; img1 item = n / rep.
// MODE 0: out = 1 - SSIM.   MODE 1: abc[n,c,3,h,w] = -grad_out * (dS/dmu2, dS/dE22, dS/dE12).
template <int MODE>
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                   const float* __restrict__ grad_out, float* __restrict__ out, GaussWin gw, int rep,
                                                   int C, int h, int w) {
    __shared__ float in_[5][SS_LH][SS_LW];
    __shared__ float hb_[5][SS_LH][SS_TW];
    const int tid = threadIdx.x;
    const long hw = (long)h * w;
    const int nc = blockIdx.z, n = nc / C, c = nc % C;
    const float* p1 = img1 + ((long)(n / rep) * C + c) * hw;
    const float* p2 = img2 + (long)nc * hw;
    const int x0 = blockIdx.x * SS_TW - SS_R, y0 = blockIdx.y * SS_TH - SS_R;
    for (int e = tid; e < SS_LH * SS_LW; e += 256) {
        const int r = e / SS_LW, cc = e % SS_LW;
        const int y = y0 + r, x = x0 + cc;
        float a = 0.f, b = 0.f;
        if (y >= 0 && y < h && x >= 0 && x < w) { a = p1[(long)y * w + x]; b = p2[(long)y * w + x]; }
        in_[0][r][cc] = a; in_[1][r][cc] = b; in_[2][r][cc] = a * a; in_[3][r][cc] = b * b; in_[4][r][cc] = a * b;
    }
    __syncthreads();
    float m[5];
    blur_tile<5>(in_, hb_, gw, tid, m);
    const int y = blockIdx.y * SS_TH + tid / SS_TW, x = blockIdx.x * SS_TW + tid % SS_TW;
    if (y >= h || x >= w) return;
    const SsimTerms t = ssim_terms(m);
    const long o = (long)y * w + x;
    if (MODE == 0) {
        out[(long)nc * hw + o] = 1.0f - t.S;                                             // ssimLoss.py:60
    } else {
        const float go = -grad_out[(long)nc * hw + o];
        float* ap = out + (long)nc * 3 * hw + o;
        ap[0] = go * t.dmu2; ap[hw] = go * t.de22; ap[2 * hw] = go * t.de12;
    }
}

// grad_img2 = blur(a) + 2 img2 blur(b) + img1 blur(c)   (the window is symmetric: the adjoint of the zero-padded blur is itself)
__global__ __launch_bounds__(256) void ssim_bwd_finish_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                              const float* __restrict__ abc, float* __restrict__ grad_img2,
                                                              GaussWin gw, int rep, int C, int h, int w) {
    __shared__ float in_[3][SS_LH][SS_LW];
    __shared__ float hb_[3][SS_LH][SS_TW];
    const int tid = threadIdx.x;
    const long hw = (long)h * w;
    const int nc = blockIdx.z, n = nc / C, c = nc % C;
    const float* ap = abc + (long)nc * 3 * hw;
    const int x0 = blockIdx.x * SS_TW - SS_R, y0 = blockIdx.y * SS_TH - SS_R;
    for (int e = tid; e < SS_LH * SS_LW; e += 256) {
        const int r = e / SS_LW, cc = e % SS_LW;
        const int y = y0 + r, x = x0 + cc;
        const bool in = y >= 0 && y < h && x >= 0 && x < w;
        const long o = (long)y * w + x;
        in_[0][r][cc] = in ? ap[o] : 0.f; in_[1][r][cc] = in ? ap[hw + o] : 0.f; in_[2][r][cc] = in ? ap[2 * hw + o] : 0.f;
    }
    __syncthreads();
    float m[3];
    blur_tile<3>(in_, hb_, gw, tid, m);
    const int y = blockIdx.y * SS_TH + tid / SS_TW, x = blockIdx.x * SS_TW + tid % SS_TW;
    if (y >= h || x >= w) return;
    const long o = (long)y * w + x;
    const float a = img1[((long)(n / rep) * C + c) * hw + o], b = img2[(long)nc * hw + o];
    grad_img2[(long)nc * hw + o] = m[0] + 2.f * b * m[1] + a * m[2];
}

static GaussWin make_window() {          // ssimLoss.py:22-24: sigma 1.5, normalised
    GaussWin g;
    double v[2 * SS_R + 1], s = 0.0;
    for (int k = 0; k <= 2 * SS_R; ++k) { v[k] = exp(-(double)((k - SS_R) * (k - SS_R)) / (2.0 * 1.5 * 1.5)); s += v[k]; }
    for (int k = 0; k <= 2 * SS_R; ++k) g.g[k] = (float)((float)v[k] / (float)s);
    return g;
}

}  // namespace pscv

extern "C" int pscv_photo_warp(const float* src_imgs, const float* depth, const float* inv_ref, const float* proj_src,
                               const float* src_depth, float* warped, float* mask, float* z_src, float* flows, float* warped_depth,
                               int B, int S, int C, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(depth && inv_ref && proj_src, "pscv_photo_warp: null pointer argument");
    PSCV_CHECK_ARG(!warped || src_imgs, "pscv_photo_warp: warped requested without source images");
    PSCV_CHECK_ARG(!warped_depth || src_depth, "pscv_photo_warp: warped_depth requested without source depth maps");
    PSCV_CHECK_ARG(B > 0 && S > 0 && C >= 0 && h > 1 && w > 1, "pscv_photo_warp: bad sizes");
    const long n = (long)B * S * h * w;
    hipLaunchKernelGGL(photo_warp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src_imgs, depth, inv_ref, proj_src, src_depth, warped, mask, z_src, flows, warped_depth, B, S, C, h, w);
    PSCV_CHECK_LAUNCH("pscv_photo_warp");
    return 0;
}

extern "C" int pscv_photo_warp_bwd(const float* src_imgs, const float* depth, const float* inv_ref, const float* proj_src,
                                   const float* grad_warped, float* grad_depth, int B, int S, int C, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(src_imgs && depth && inv_ref && proj_src && grad_warped && grad_depth, "pscv_photo_warp_bwd: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && S > 0 && C > 0 && h > 1 && w > 1, "pscv_photo_warp_bwd: bad sizes");
    const long n = (long)B * h * w;
    hipLaunchKernelGGL(photo_warp_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src_imgs, depth, inv_ref, proj_src, grad_warped, grad_depth, B, S, C, h, w);
    PSCV_CHECK_LAUNCH("pscv_photo_warp_bwd");
    return 0;
}

extern "C" int pscv_ssim(const float* img1, const float* img2, float* out, int n1, int rep, int C, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(img1 && img2 && out, "pscv_ssim: null pointer argument");
    PSCV_CHECK_ARG(n1 > 0 && rep > 0 && C > 0 && h > 0 && w > 0 && (long)n1 * rep * C <= 65535, "pscv_ssim: bad sizes");
    const dim3 grid((w + SS_TW - 1) / SS_TW, (h + SS_TH - 1) / SS_TH, n1 * rep * C);
    hipLaunchKernelGGL(ssim_kernel<0>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img1, img2, (const float*)nullptr, out,
                       make_window(), rep, C, h, w);
    PSCV_CHECK_LAUNCH("pscv_ssim");
    return 0;
}

extern "C" int pscv_ssim_bwd(const float* img1, const float* img2, const float* grad_out, float* workspace, float* grad_img2, int n1,
                             int rep, int C, int h, int w, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(img1 && img2 && grad_out && workspace && grad_img2, "pscv_ssim_bwd: null pointer argument");
    PSCV_CHECK_ARG(n1 > 0 && rep > 0 && C > 0 && h > 0 && w > 0 && (long)n1 * rep * C <= 65535, "pscv_ssim_bwd: bad sizes");
    const dim3 grid((w + SS_TW - 1) / SS_TW, (h + SS_TH - 1) / SS_TH, n1 * rep * C);
    const GaussWin gw = make_window();
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ssim_kernel<1>, grid, dim3(256), 0, st, img1, img2, grad_out, workspace, gw, rep, C, h, w);
    hipLaunchKernelGGL(ssim_bwd_finish_kernel, grid, dim3(256), 0, st, img1, img2, (const float*)workspace, grad_img2, gw, rep, C, h, w);
    PSCV_CHECK_LAUNCH("pscv_ssim_bwd");
    return 0;
}
