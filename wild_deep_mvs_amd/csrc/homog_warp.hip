// Function-level homography warp with one 3x3 matrix per batch item OR per reference pixel (gfx950).
//
// Utility kernel behind the drop-in of `homography_warping(input, H)`: inside the model the per-plane / per-pixel homographies
// never exist (the fused sweep builds them from A, Bm and the depth planes, warp_common.h), but callers of the function itself may
// hand over arbitrary matrices [m,h,w,3,3].  Same sampling rule as the sweep: pixel centres at +0.5, points with z <= 0 go to
// (-10, -10), divisor clamped at 1e-9, normalise -> clamp(+-1.1) -> align_corners=True bilinear with zero padding.
//
// Replaces (fdarmon/wild_deep_mvs): models/VisMVSNet/homography.py:107-120 (homography_warping) + 84-104 (interpolate).
#include "pscv_common.h"

namespace pscv {

__global__ __launch_bounds__(256) void homography_warp_kernel(const float* __restrict__ img, const float* __restrict__ Hm, int per_pixel,
                                                              float* __restrict__ out, long total, int c, int h, int w, int hs, int ws,
                                                              float sx, float sy, float xlo, float xhi, float ylo, float yhi) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int ch = (int)(t % c);
    const long pix = t / c;
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const long b = pix / ((long)w * h);
    const float* H = Hm + (per_pixel ? pix : b) * 9;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float hx = fmaf(H[1], py, H[0] * px) + H[2];
    const float hy = fmaf(H[4], py, H[3] * px) + H[5];
    const float hz = fmaf(H[7], py, H[6] * px) + H[8];
    const bool front = hz > 0.0f;
    const float inv_z = 1.0f / fmaxf(hz, 1e-9f);
    const float u = front ? hx * inv_z : -10.0f, v = front ? hy * inv_z : -10.0f;
    const float ix = fminf(fmaxf(u * sx, xlo), xhi), iy = fminf(fmaxf(v * sy, ylo), yhi);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float* ib = img + b * (long)hs * ws * c + ch;
    auto tap = [&](int yy, int xx) -> float {
        return ((unsigned)xx < (unsigned)ws && (unsigned)yy < (unsigned)hs) ? ib[((long)yy * ws + xx) * c] : 0.0f;
    };
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    out[t] = tap(y0, x0) * (gx * gy) + tap(y0, x0 + 1) * (fx * gy) + tap(y0 + 1, x0) * (gx * fy) + tap(y0 + 1, x0 + 1) * (fx * fy);
}

// Adjoint of the kernel above with respect to the IMAGE (the reference computes the sample positions under no_grad,
// homography.py:110-118, so `input` is the only differentiable argument of homography_warping): every reference pixel adds
// grad_out x its four bilinear weights to the taps it read; taps outside the source image received weight on zeros and add nothing.
__global__ __launch_bounds__(256) void homography_warp_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ Hm, int per_pixel,
                                                                  float* __restrict__ gimg, long total, int c, int h, int w, int hs, int ws,
                                                                  float sx, float sy, float xlo, float xhi, float ylo, float yhi) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int ch = (int)(t % c);
    const long pix = t / c;
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const long b = pix / ((long)w * h);
    const float* H = Hm + (per_pixel ? pix : b) * 9;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float hx = fmaf(H[1], py, H[0] * px) + H[2];
    const float hy = fmaf(H[4], py, H[3] * px) + H[5];
    const float hz = fmaf(H[7], py, H[6] * px) + H[8];
    const bool front = hz > 0.0f;
    const float inv_z = 1.0f / fmaxf(hz, 1e-9f);
    const float u = front ? hx * inv_z : -10.0f, v = front ? hy * inv_z : -10.0f;
    const float ix = fminf(fmaxf(u * sx, xlo), xhi), iy = fminf(fmaxf(v * sy, ylo), yhi);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    float* gb = gimg + b * (long)hs * ws * c + ch;
    const float g = gout[t];
    auto tap = [&](int yy, int xx, float wgt) {
        if ((unsigned)xx < (unsigned)ws && (unsigned)yy < (unsigned)hs) atomicAdd(gb + ((long)yy * ws + xx) * c, g * wgt);
    };
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    tap(y0, x0, gx * gy); tap(y0, x0 + 1, fx * gy); tap(y0 + 1, x0, gx * fy); tap(y0 + 1, x0 + 1, fx * fy);
}

}  // namespace pscv

extern "C" int pscv_homography_warp_bwd(const float* grad_out, const float* H, int per_pixel, float* grad_image, int m, int c, int h, int w,
                                        int hs, int ws, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(grad_out && H && grad_image, "pscv_homography_warp_bwd: null pointer argument");
    PSCV_CHECK_ARG(m > 0 && c > 0 && h > 0 && w > 0 && hs > 0 && ws > 0, "pscv_homography_warp_bwd: bad sizes");
    const long total = (long)m * h * w * c;
    const long nblk = (total + 255) / 256;
    PSCV_CHECK_ARG(nblk <= 0x7fffffffL, "pscv_homography_warp_bwd: grid too large (%ld workgroups)", nblk);
    hipLaunchKernelGGL(homography_warp_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad_out, H,
                       per_pixel, grad_image, total, c, h, w, hs, ws, (float)(ws - 1) / (float)ws, (float)(hs - 1) / (float)hs,
                       -0.05f * (ws - 1), 1.05f * (ws - 1), -0.05f * (hs - 1), 1.05f * (hs - 1));
    PSCV_CHECK_LAUNCH("pscv_homography_warp_bwd");
    return 0;
}

extern "C" int pscv_homography_warp(const float* image, const float* H, int per_pixel, float* out, int m, int c, int h, int w, int hs,
                                    int ws, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(image && H && out, "pscv_homography_warp: null pointer argument");
    PSCV_CHECK_ARG(m > 0 && c > 0 && h > 0 && w > 0 && hs > 0 && ws > 0, "pscv_homography_warp: bad sizes");
    const long total = (long)m * h * w * c;
    const long nblk = (total + 255) / 256;
    PSCV_CHECK_ARG(nblk <= 0x7fffffffL, "pscv_homography_warp: grid too large (%ld workgroups)", nblk);
    hipLaunchKernelGGL(homography_warp_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), image, H,
                       per_pixel, out, total, c, h, w, hs, ws, (float)(ws - 1) / (float)ws, (float)(hs - 1) / (float)hs,
                       -0.05f * (ws - 1), 1.05f * (ws - 1), -0.05f * (hs - 1), 1.05f * (hs - 1));
    PSCV_CHECK_LAUNCH("pscv_homography_warp");
    return 0;
}
