// 3x3x3 convolution with ONE output channel (the `prob` heads: MVSNet 8->1, CVP 16->1, Vis final_conv 8->1).
//
// A 1-channel output would waste 15 of the 16 MFMA rows, and at 216 (432) MACs per voxel the layer is HBM-bound
// anyway, so it runs on the vector ALU: one lane per output voxel, `v_dot2_f32_f16` / `v_dot2_f32_bf16` (two MACs
// per instruction, fp32 accumulate), input planes swept through a 4-slot LDS ring (each input plane is fetched
// once per sweep, prefetched into registers one iteration ahead), weights as wave-uniform scalar operands.
// A workgroup owns an 8 x 32 pixel tile; a wave reads 2 rows x 32 consecutive voxels = conflict-free LDS rows.
//
// Replaces (fdarmon/wild_deep_mvs): CostRegNet.prob models/MVSNet/model.py:72,82; prob0 models/CVP_MVSNet/models/
// net.py:76,83; RegPair / RegFuse final_conv models/VisMVSNet/model_cas.py:55,68.
#include "pscv_common.h"

namespace pscv {

typedef _Float16 c1_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 c1_b2 __attribute__((ext_vector_type(2)));

template <typename H> __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c);
template <> __device__ __forceinline__ float dot2<f16_t>(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(c1_h2, a), __builtin_bit_cast(c1_h2, b), c, false);
}
template <> __device__ __forceinline__ float dot2<bf16_t>(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(c1_b2, a), __builtin_bit_cast(c1_b2, b), c, false);
}

struct C1Args {
    const uint16_t* in;
    const uint4* wpk;        // [27 taps][CIN/8] x 8 halves, tap = kd*9 + kh*3 + kw
    const uint16_t* skip;
    void* out;
    const float* scale;      // device, [1] each (may be null)
    const float* bias;
    const float* floor;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, D, Hh, W;
    int epi;
    int nth, ntw, ndc, dc;
};

constexpr int C1_TH = 8, C1_TW = 32, C1_BH = C1_TH + 2, C1_BW = C1_TW + 2, C1_PV = C1_BH * C1_BW, C1_NSLOT = 4;

template <typename H, int CIN>
__global__ __launch_bounds__(256) void conv3d_c1_kernel(const C1Args a) {
    constexpr int CCH = CIN / 8;             // 16-byte chunks per voxel
    constexpr int VB = CIN * 2;              // bytes per voxel
    constexpr int PB = C1_PV * VB;           // bytes per plane slot
    constexpr int NCH = C1_PV * CCH;         // chunks per plane
    constexpr int NLD = (NCH + 255) / 256;   // chunks per thread per plane
    __shared__ __attribute__((aligned(16))) unsigned char smem[C1_NSLOT * PB];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot_;
    const int dci = wg % a.ndc; wg /= a.ndc;
    const int twi = wg % a.ntw; wg /= a.ntw;
    const int thi = wg % a.nth; wg /= a.nth;
    const int b = wg;
    const int h0 = thi * C1_TH, w0 = twi * C1_TW;
    const int dbeg = dci * a.dc, dend = min(a.D, dbeg + a.dc);

    const int tid = threadIdx.x;
    const int row = tid / C1_TW, col = tid % C1_TW;

    // staging descriptors
    int goff[NLD], loff[NLD];
    bool gval[NLD], lval[NLD];
    const long plane_stride = (long)a.Hh * a.W * a.in_cs;
    const uint16_t* inb = a.in + (long)b * a.D * plane_stride + a.in_co;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i;
        const int v = id / CCH, c = id - v * CCH;
        const int bh = v / C1_BW, bw = v - bh * C1_BW;
        const int gh = h0 - 1 + bh, gw = w0 - 1 + bw;
        lval[i] = id < NCH;
        gval[i] = lval[i] && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
        goff[i] = gval[i] ? (gh * a.W + gw) * a.in_cs + c * 8 : 0;
        loff[i] = v * VB + c * 16;
    }
    const int plane_hi = min(a.D - 1, dend);
    auto fetch = [&](int plane, uint4 (&reg)[NLD]) {
        const bool pv = plane >= 0 && plane <= plane_hi;
        const uint16_t* pp = inb + (long)(pv ? plane : 0) * plane_stride;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            reg[i] = make_uint4(0u, 0u, 0u, 0u);
            if (gval[i] && pv) reg[i] = *reinterpret_cast<const uint4*>(pp + goff[i]);
        }
    };
    auto stash = [&](int ring, const uint4 (&reg)[NLD]) {
        unsigned char* sp = smem + ring * PB;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lval[i]) *reinterpret_cast<uint4*>(sp + loff[i]) = reg[i];
    };

    {   // prologue: planes dbeg-1, dbeg, dbeg+1 -> slots 0, 1, 2
        uint4 r0[NLD], r1[NLD], r2[NLD];
        fetch(dbeg - 1, r0); fetch(dbeg, r1); fetch(dbeg + 1, r2);
        stash(0, r0); stash(1, r1); stash(2, r2);
    }
    __syncthreads();

    // C_in = 8: the 27 weight quads live in VGPRs (same value in every lane) for the whole sweep; re-reading them
    // through the scalar cache every plane exposed its latency once per iteration.  C_in = 16 would need 216
    // registers and keeps the scalar-operand path.
    constexpr bool WREG = CIN == 8;
    uint4 wreg[WREG ? 27 * CCH : 1];
    if (WREG) {
#pragma unroll
        for (int t = 0; t < 27 * CCH; ++t) wreg[t] = a.wpk[t];
    }
    const float e_scale = a.scale ? a.scale[0] : 1.0f, e_bias = a.bias ? a.bias[0] : 0.0f,
                e_floor = a.floor ? a.floor[0] : 0.0f;
    const int lane_off = (row * C1_BW + col) * VB;   // this lane's voxel at tap (kh=0, kw=0)
    const int oh = h0 + row, ow = w0 + col;
    const bool inside = oh < a.Hh && ow < a.W;
    int ring = 0;   // slot of plane d-1
    for (int d = dbeg; d < dend; ++d) {
        uint4 nxt[NLD];
        fetch(d + 2, nxt);

        float acc = 0.0f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const unsigned char* sp = smem + ((ring + kd) & (C1_NSLOT - 1)) * PB + lane_off;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int c = 0; c < CCH; ++c) {
                        const uint4 x = *reinterpret_cast<const uint4*>(sp + (kh * C1_BW + kw) * VB + c * 16);
                        const uint4 w = WREG ? wreg[((kd * 3 + kh) * 3 + kw) * CCH + c]
                                             : a.wpk[((kd * 3 + kh) * 3 + kw) * CCH + c];   // wave-uniform -> scalar loads
                        acc = dot2<H>(x.x, w.x, acc);
                        acc = dot2<H>(x.y, w.y, acc);
                        acc = dot2<H>(x.z, w.z, acc);
                        acc = dot2<H>(x.w, w.w, acc);
                    }
        }

        if (inside) {
            const long vox = (((long)b * a.D + d) * a.Hh + oh) * a.W + ow;
            float y = fmaf(acc, e_scale, e_bias);
            if (a.epi & PSCV_EPI_RELU_PRE) y = fmaxf(y, e_floor);
            if (a.skip) y += Half16<H>::one(a.skip[vox * a.skip_cs + a.skip_co]);
            if (a.epi & PSCV_EPI_RELU_POST) y = fmaxf(y, 0.0f);
            if (a.out_f32) reinterpret_cast<float*>(a.out)[vox * a.out_cs + a.out_co] = y;
            else reinterpret_cast<uint16_t*>(a.out)[vox * a.out_cs + a.out_co] = Half16<H>::bits(y);
        }

        stash((ring + 3) & (C1_NSLOT - 1), nxt);   // plane d+2 replaces plane d-2 (last read one iteration ago)
        ring = (ring + 1) & (C1_NSLOT - 1);
        __syncthreads();
    }
}

}  // namespace pscv

int pscv_conv3d_c1_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed,
                          const float* scale, const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                          int out_cstride, int out_coff, int out_dtype, int B, int D, int Hh, int W, int c_in,
                          int epi_flags, hipStream_t st) {
    using namespace pscv;
    C1Args a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = reinterpret_cast<const uint4*>(packed);
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.scale = scale; a.bias = bias; a.floor = floor;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.out_cs = out_cstride; a.out_co = out_coff; a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.D = D; a.Hh = Hh; a.W = W; a.epi = epi_flags;
    a.nth = (Hh + C1_TH - 1) / C1_TH;
    a.ntw = (W + C1_TW - 1) / C1_TW;
    const long tiles = (long)B * a.nth * a.ntw;
    // one resident round of workgroups (about 4 per CU at this kernel's register / LDS footprint), fewest chunk seams
    const long slots = 1024;
    const long ndc_want = tiles >= slots ? 1 : slots / tiles;
    int dc = (int)((D + ndc_want - 1) / ndc_want);
    dc = dc < 4 ? 4 : dc;
    dc = dc > D ? D : dc;
    a.dc = dc;
    a.ndc = (D + dc - 1) / dc;
    const long nblk = tiles * a.ndc;
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(c1): bad grid %ld", nblk); return -1; }
    const dim3 grid((unsigned)nblk), block(256);
    if (dtype == PSCV_BF16 && c_in == 8) hipLaunchKernelGGL((conv3d_c1_kernel<bf16_t, 8>), grid, block, 0, st, a);
    else if (dtype == PSCV_BF16 && c_in == 16) hipLaunchKernelGGL((conv3d_c1_kernel<bf16_t, 16>), grid, block, 0, st, a);
    else if (dtype == PSCV_F16 && c_in == 8) hipLaunchKernelGGL((conv3d_c1_kernel<f16_t, 8>), grid, block, 0, st, a);
    else if (dtype == PSCV_F16 && c_in == 16) hipLaunchKernelGGL((conv3d_c1_kernel<f16_t, 16>), grid, block, 0, st, a);
    else { set_error("pscv_conv3d(c1): c_in=%d dtype=%d not supported (c_in 8 or 16)", c_in, dtype); return -1; }
    return 0;
}
