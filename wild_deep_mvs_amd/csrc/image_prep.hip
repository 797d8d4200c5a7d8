// fp32 NCHW images -> the 2-D extractors' input layout, and CVP's image pyramid (gfx950).
//
// Every extractor's first layer reads [B,H,W,8] 16-bit channels-last pixels (3 colour channels + 5 zero channels: one 16-byte
// chunk per pixel, the MFMA k-padding of csrc/conv2d.hip).  On torch ops that was a zero fill + a strided copy per image
// batch, and CVP's pyramid (FeaturePyramid.forward, models/CVP_MVSNet/models/net.py:34-47: `F.interpolate(img, scale_factor=0.5,
// mode='bilinear')` between the towers) one more launch per level.  Here: one launch per level that reads the fp32 image
// once and writes (optionally) its 8-channel 16-bit form, the half-resolution fp32 image and that one's 8-channel form.
// The half-scale bilinear resample (align_corners = False, source index 2 x + 0.5) is the 2 x 2 mean in ATen's own operation
// order 0.5 (0.5 a + 0.5 b) + 0.5 (0.5 c + 0.5 d): the products are exact, so the result has ATen's bits.
#include "pscv_common.h"

namespace pscv {

template <typename H>
__global__ __launch_bounds__(256) void image_prep_kernel(const float* __restrict__ img, int C, int Hh, int W, uint4* __restrict__ out_cl8,
                                                         float* __restrict__ half_img, uint4* __restrict__ half_cl8) {
    const int b = blockIdx.y;
    const long plane = (long)Hh * W;
    const float* __restrict__ src = img + (long)b * C * plane;
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (out_cl8 && pix < plane) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) v[c] = src[c * plane + pix];
        out_cl8[(long)b * plane + pix] = make_uint4(Half16<H>::pack(v[0], v[1]), Half16<H>::pack(v[2], v[3]), Half16<H>::pack(v[4], v[5]),
                                                    Half16<H>::pack(v[6], v[7]));
    }
    const int H2 = Hh >> 1, W2 = W >> 1;
    const long plane2 = (long)H2 * W2;
    if ((half_img || half_cl8) && pix < plane2) {
        const int y = (int)(pix / W2), x = (int)(pix - (long)y * W2);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) {
                const float* p = src + c * plane + (long)(2 * y) * W + 2 * x;
                const float2 r0 = *reinterpret_cast<const float2*>(p), r1 = *reinterpret_cast<const float2*>(p + W);
                v[c] = 0.5f * (0.5f * r0.x + 0.5f * r0.y) + 0.5f * (0.5f * r1.x + 0.5f * r1.y);
                if (half_img) half_img[((long)b * C + c) * plane2 + pix] = v[c];
            }
        if (half_cl8)
            half_cl8[(long)b * plane2 + pix] = make_uint4(Half16<H>::pack(v[0], v[1]), Half16<H>::pack(v[2], v[3]), Half16<H>::pack(v[4], v[5]),
                                                          Half16<H>::pack(v[6], v[7]));
    }
}

}  // namespace pscv

using namespace pscv;

extern "C" int pscv_image_prep(const float* img, int B, int C, int H, int W, int dtype, void* out_cl8, float* half_img, void* half_cl8,
                               void* stream) {
    PSCV_CHECK_ARG(img && (out_cl8 || half_img || half_cl8), "pscv_image_prep: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && C <= 8 && H > 0 && W > 0, "pscv_image_prep: bad sizes B=%d C=%d H=%d W=%d", B, C, H, W);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_image_prep: storage dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(!(half_img || half_cl8) || (H >= 2 && W >= 2 && W % 2 == 0), "pscv_image_prep: the half-resolution outputs need an even width (8-byte row pairs) and H, W >= 2");
    const long n = out_cl8 ? (long)H * W : (long)(H / 2) * (W / 2);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PSCV_BF16)
        hipLaunchKernelGGL(image_prep_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, img, C, H, W, reinterpret_cast<uint4*>(out_cl8),
                           half_img, reinterpret_cast<uint4*>(half_cl8));
    else
        hipLaunchKernelGGL(image_prep_kernel<f16_t>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, img, C, H, W, reinterpret_cast<uint4*>(out_cl8),
                           half_img, reinterpret_cast<uint4*>(half_cl8));
    PSCV_CHECK_LAUNCH("pscv_image_prep");
    return 0;
}
