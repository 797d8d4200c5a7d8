// Vis-MVSNet's UncertNet on the entropy maps of all pairs, ONE launch (gfx950).
//
// Reference: UncertNet, models/VisMVSNet/model_cas.py:77-98 -- [n,1,h,w] entropy -> Conv2d(1,8,3,p1) + BN + ReLU ->
// Conv2d(8,8,3,p1) + BN + ReLU -> `out += x` (the 1-channel input broadcast over the 8 channels) -> head Conv2d(8,1,3,p1):
// the log-uncertainty map that weights a pair's volume in the fusion (model_cas.py:349-357).  Eval mode (BatchNorm folded to a
// per-channel scale and bias, like every other layer of the eval path).
//
// On PyTorch-ROCm this was 3 fp32 MIOpen convolutions + 2 BatchNorms + 2 ReLUs + the add per stage: 1.4 ms of configuration
// 5's 19.4 ms for 1.2 % of its arithmetic (the 8-channel intermediates of eight 576 x 800 maps make four round trips through
// HBM).  Here a workgroup owns a 16 x 32 output tile and keeps everything in LDS: the 22 x 38 input patch, the 8-channel
// first layer on the 20 x 36 halo-2 grid, the second layer (+ x) on the 18 x 34 halo-1 grid; each convolution pads ITS input
// with zeros, so grid points outside the image hold 0, not values computed from a padded input.  fp32 FMAs on the VALU (the
// reference computes in fp32; 861 FMAs per output pixel incl. 20 % halo recompute), the weights are wave-uniform: scalar loads
// from the 752-float parameter block.  HBM traffic: 4 B read + 4 B written per pixel.
#include "pscv_common.h"

namespace pscv {

constexpr int UN_TH = 16, UN_TW = 32;
constexpr int UN_XH = UN_TH + 6, UN_XW = UN_TW + 6;     // input patch
constexpr int UN_1H = UN_TH + 4, UN_1W = UN_TW + 4;     // layer-1 grid
constexpr int UN_2H = UN_TH + 2, UN_2W = UN_TW + 2;     // layer-2 grid
// parameter block (floats): w1 [tap][co] 72 | s1 8 | b1 8 | w2 [ci][tap][co] 576 | s2 8 | b2 8 | head [ci][tap] 72
constexpr int UN_W1 = 0, UN_S1 = 72, UN_B1 = 80, UN_W2 = 88, UN_S2 = 664, UN_B2 = 672, UN_WH = 680, UN_PARAMS = 752;

typedef float un_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void uncert_net_kernel(const float* __restrict__ ent, const float* __restrict__ prm,
                                                         float* __restrict__ out, int H, int W, int ntw) {
    __shared__ float xs[UN_XH * UN_XW];
    __shared__ float t1[8][UN_1H * UN_1W];
    __shared__ float t2[8][UN_2H * UN_2W];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int th_i = blockIdx.x / ntw, tw_i = blockIdx.x - th_i * ntw;
    const int y0 = th_i * UN_TH, x0 = tw_i * UN_TW;
    const float* __restrict__ src = ent + (long)b * H * W;

    for (int p = tid; p < UN_XH * UN_XW; p += 256) {
        const int r = p / UN_XW, c = p - r * UN_XW;
        const int gy = y0 - 3 + r, gx = x0 - 3 + c;
        xs[p] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? src[(long)gy * W + gx] : 0.0f;
    }
    __syncthreads();

    // layer 1: 1 -> 8 on the halo-2 grid
    for (int p = tid; p < UN_1H * UN_1W; p += 256) {
        const int r = p / UN_1W, c = p - r * UN_1W;
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        un_f2 acc[4] = {};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = xs[(r + t / 3) * UN_XW + c + t % 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const un_f2 w = {prm[UN_W1 + t * 8 + 2 * j], prm[UN_W1 + t * 8 + 2 * j + 1]};
                    acc[j] = __builtin_elementwise_fma(un_f2{a, a}, w, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j].x = fmaxf(fmaf(acc[j].x, prm[UN_S1 + 2 * j], prm[UN_B1 + 2 * j]), 0.0f);
                acc[j].y = fmaxf(fmaf(acc[j].y, prm[UN_S1 + 2 * j + 1], prm[UN_B1 + 2 * j + 1]), 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t1[2 * j][p] = acc[j].x; t1[2 * j + 1][p] = acc[j].y; }
    }
    __syncthreads();

    // layer 2: 8 -> 8 on the halo-1 grid, + x
    for (int p = tid; p < UN_2H * UN_2W; p += 256) {
        const int r = p / UN_2W, c = p - r * UN_2W;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        un_f2 acc[4] = {};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float a = t1[ci][(r + t / 3) * UN_1W + c + t % 3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const un_f2 w = {prm[UN_W2 + (ci * 9 + t) * 8 + 2 * j], prm[UN_W2 + (ci * 9 + t) * 8 + 2 * j + 1]};
                        acc[j] = __builtin_elementwise_fma(un_f2{a, a}, w, acc[j]);
                    }
                }
            const float xc = xs[(r + 2) * UN_XW + c + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j].x = fmaxf(fmaf(acc[j].x, prm[UN_S2 + 2 * j], prm[UN_B2 + 2 * j]), 0.0f) + xc;
                acc[j].y = fmaxf(fmaf(acc[j].y, prm[UN_S2 + 2 * j + 1], prm[UN_B2 + 2 * j + 1]), 0.0f) + xc;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t2[2 * j][p] = acc[j].x; t2[2 * j + 1][p] = acc[j].y; }
    }
    __syncthreads();

    // head: 8 -> 1 on the tile
    for (int p = tid; p < UN_TH * UN_TW; p += 256) {
        const int r = p / UN_TW, c = p - r * UN_TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        float acc = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(t2[ci][(r + t / 3) * UN_2W + c + t % 3], prm[UN_WH + ci * 9 + t], acc);
        out[((long)b * H + gy) * W + gx] = acc;
    }
}

}  // namespace pscv

using namespace pscv;

extern "C" int pscv_uncert_net(const float* entropy, const float* params, float* out, int N, int H, int W, void* stream) {
    PSCV_CHECK_ARG(entropy && params && out, "pscv_uncert_net: null pointer argument");
    PSCV_CHECK_ARG(N > 0 && H > 0 && W > 0 && N <= 65535, "pscv_uncert_net: bad sizes N=%d H=%d W=%d", N, H, W);
    const int nth = (H + UN_TH - 1) / UN_TH, ntw = (W + UN_TW - 1) / UN_TW;
    hipLaunchKernelGGL(uncert_net_kernel, dim3(nth * ntw, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), entropy, params,
                       out, H, W, ntw);
    PSCV_CHECK_LAUNCH("pscv_uncert_net");
    return 0;
}
