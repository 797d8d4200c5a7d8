// 3x3x3 convolution with ONE output channel (the `prob` heads: MVSNet 8->1, CVP 16->1, Vis final_conv 8->1) on the
// matrix cores, with the DEPTH axis packed into the MFMA rows.  gfx950.
//
// A 1-channel output uses one of the 16 MFMA rows.  Here the rows are six consecutive OUTPUT PLANES instead: for a
// block of output planes d0..d0+5 the reduction runs over the eight input planes d0-1..d0+6, and row m of the A operand
// holds kernel slice kd = p - m for input plane p (zero where kd falls outside 0..2) -- a banded (Toeplitz) weight
// matrix built once on the host (pscv_pack_conv3d_weights, kind S1C1).  K = 8 planes x 9 taps x C_in = 576 (1152) is
// exactly 18 (36) k-steps of 32, so one 16-pixel x 6-plane output tile costs 18 (36) MFMAs and 18 (36) ds_read_b128,
// 0.19 MFMA and 192 LDS bytes per voxel, and no vector-ALU work beyond the epilogue.  (The previous vector-ALU dot2
// sweep needed 108 dependent v_dot2 and 27 LDS reads per voxel behind a per-plane barrier: 38 us at the headline size,
// a quarter of its HBM roofline.)
//
// Workgroup: 4 x 32 output pixels x NB blocks of 6 planes; the (6 NB + 2) x 6 x 34 input brick is staged into LDS once
// (all loads in flight together, one barrier); wave w owns tile row w (two 16-pixel MFMA column tiles, sharing A).
// k-step operand order (what lane group g = lane >> 4 holds) is chosen so that every ds_read_b128 is bank-conflict
// free and its address is lane base + immediate: step s = (q, tap); C_in = 8: plane 4 (g >> 1) + 2 q + (g & 1), with a
// plane stride of 208 voxels (= 0 mod 16 chunks); C_in = 16: plane 4 (g >> 1) + q, g & 1 = channel half.
//
// Replaces (fdarmon/wild_deep_mvs): CostRegNet.prob models/MVSNet/model.py:72,82; prob0 models/CVP_MVSNet/models/
// net.py:76,83; RegPair / RegFuse final_conv models/VisMVSNet/model_cas.py:55,68.
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 c1_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 c1_f16x8;
typedef __attribute__((ext_vector_type(4))) float c1_f32x4;

template <typename H> struct C1Mfma;
template <> struct C1Mfma<bf16_t> {
    __device__ static __forceinline__ c1_f32x4 run(const uint4& a, const uint4& b, const c1_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c1_bf16x8, a), __builtin_bit_cast(c1_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct C1Mfma<f16_t> {
    __device__ static __forceinline__ c1_f32x4 run(const uint4& a, const uint4& b, const c1_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_f16x8, a), __builtin_bit_cast(c1_f16x8, b), c, 0, 0, 0);
    }
};

struct C1Args {
    const uint16_t* in;
    const uint4* wpk;        // [NSTEPS][64 lanes] x 8 halves (A fragments)
    const uint16_t* skip;
    void* out;
    const float* scale;      // device, [1] each (may be null)
    const float* bias;
    const float* floor;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, D, Hh, W;
    int epi;
    int nth, ntw, ndc, nb;   // tiles, depth chunks, 6-plane blocks per workgroup
    unsigned mg_th, mg_tw, mg_dc;
    // fused softmax partials (depth-sweep variant only; null = off): per (batch, depth chunk, pixel) the running max, sum of
    // exponentials, sum exp * depth plane and sum exp * plane index of the chunk's logits -- merged by softargmin_merge_kernel
    const float* depth;      // [B][D] planes, row stride depth_bstride
    long depth_bstride;
    float* part;             // [B][ndc][4][Hh][W]
    // FUSE = 2 (Vis pair head, pscv_head_index_entropy): the statistics are (max, sum e, sum e * (logit - max), sum e * index); with one
    // depth chunk the kernel writes the expected index and the entropy itself, otherwise the partials above + the merge launch
    float* o_index;          // [B][Hh][W]
    float* o_entropy;
};

constexpr int C1_P = 6;                        // output planes per block (MFMA rows 0..5)
constexpr int C1_TH = 4, C1_TW = 32, C1_BH = C1_TH + 2, C1_BW = C1_TW + 2;
constexpr int C1_PS = 208;                     // plane stride in voxels: 6 x 34 = 204 padded to 0 mod 16
constexpr int C1_NB_MAX = 2;
PSCV_PROF_BUFFER(c1)

template <typename H, int CIN>
__global__ __launch_bounds__(256) void conv3d_c1_kernel(const C1Args a) {
    constexpr int CCH = CIN / 8;                       // 16-byte chunks per voxel
    constexpr int VB = CIN * 2;                        // bytes per voxel
    constexpr int NSTEPS = 8 * 9 * CIN / 32;           // 18 / 36
    constexpr int PV = C1_BH * C1_BW;                  // 204 voxels per brick plane
    constexpr int MAXP = C1_NB_MAX * C1_P + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [(6 nb + 2)][C1_PS][VB]

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * C1_TH, w0 = twi * C1_TW;
    const int brick_planes = a.nb * C1_P;              // output planes of this workgroup
    const int nplanes = brick_planes + 2;              // staged planes dbeg-1 .. dbeg + brick_planes
    const int dbeg = dci * brick_planes;
    const bool interior = dbeg >= 1 && dbeg + brick_planes < a.D && a.nb == C1_NB_MAX;   // all staged planes exist

    const int tid = threadIdx.x;
    PSCV_PROF_BEGIN
    {   // ---- stage the brick: every load of the workgroup is in flight before the first LDS write.  Thread t < 204
        // owns brick voxel t of EVERY plane: its offset inside a plane and its LDS offset are computed once, the plane
        // advances through an immediate / scalar offset -- no per-load vector arithmetic, and for an interior brick
        // one predicate for the whole batch (a flat chunk-id decomposition made this kernel vector-ALU bound).
        const unsigned long plane_bytes = (unsigned long)a.Hh * a.W * a.in_cs * 2;
        const char* inb = reinterpret_cast<const char*>(a.in) + ((unsigned long)b * a.D * plane_bytes + (unsigned long)a.in_co * 2);
        const int sbh = tid / C1_BW, sbw = tid - sbh * C1_BW;
        const int sgh = h0 - 1 + sbh, sgw = w0 - 1 + sbw;
        const bool vox_ok = tid < PV && (unsigned)sgh < (unsigned)a.Hh && (unsigned)sgw < (unsigned)a.W;
        const unsigned goff = vox_ok ? (unsigned)(sgh * a.W + sgw) * (unsigned)(a.in_cs * 2) : 0u;
        uint4 reg[MAXP][CCH];
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
#pragma unroll
            for (int c = 0; c < CCH; ++c) reg[p][c] = make_uint4(0u, 0u, 0u, 0u);
        if (interior) {
            const char* pp = inb + (unsigned long)(dbeg - 1) * plane_bytes + goff;
            if (vox_ok) {
#pragma unroll
                for (int p = 0; p < MAXP; ++p)
#pragma unroll
                    for (int c = 0; c < CCH; ++c) reg[p][c] = *reinterpret_cast<const uint4*>(pp + p * plane_bytes + c * 16);
            }
        } else {
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                const int gd = dbeg - 1 + p;                       // wave-uniform
                const bool pv = p < nplanes && (unsigned)gd < (unsigned)a.D;
                const char* pp = inb + (unsigned long)(pv ? gd : 0) * plane_bytes;
#pragma unroll
                for (int c = 0; c < CCH; ++c)
                    if (pv && vox_ok) reg[p][c] = *reinterpret_cast<const uint4*>(pp + goff + c * 16);
            }
        }
        PSCV_STAMP(0)
        PSCV_STAMP_WAIT(1)
        unsigned char* sp = smem + tid * VB;
        if (tid < PV) {
#pragma unroll
            for (int p = 0; p < MAXP; ++p)
                if (p < nplanes) {
#pragma unroll
                    for (int c = 0; c < CCH; ++c) *reinterpret_cast<uint4*>(sp + p * (C1_PS * VB) + c * 16) = reg[p][c];
                }
        }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    uint4 wf[NSTEPS];
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s)
        wf[s] = a.wpk[s * 64 + lane];
    const float e_scale = a.scale ? a.scale[0] : 1.0f, e_bias = a.bias ? a.bias[0] : 0.0f,
                e_floor = a.floor ? a.floor[0] : 0.0f;
    // ReLU switches as clamps against -inf (no branches in the epilogue)
    const float lo_pre = (a.epi & PSCV_EPI_RELU_PRE) ? e_floor : -__builtin_inff();
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    __syncthreads();
    PSCV_STAMP(2)

    // byte offset of this lane's B operand at step (q = 0, tap 0) of column tile 0: voxel (row = wave, col = n) of the
    // lane group's first plane
    const int lane_off = (CIN == 8) ? ((4 * (g >> 1) + (g & 1)) * C1_PS + wave * C1_BW + n) * VB
                                    : (4 * (g >> 1) * C1_PS + wave * C1_BW + n) * VB + (g & 1) * 16;
    const int oh = h0 + wave;
    // output addressing: wave-uniform 64-bit base per (block, plane), 32-bit lane offsets computed once
    const int OB = a.out_f32 ? 4 : 2;
    const unsigned long out_plane = (unsigned long)a.Hh * a.W * a.out_cs * OB;
    const unsigned long skip_plane = (unsigned long)a.Hh * a.W * a.skip_cs * 2;
    const unsigned pix0 = (unsigned)(oh * a.W + w0 + n);
    const bool row_ok = g < 2 && oh < a.Hh;
    const unsigned ostep = (unsigned)a.out_cs * (unsigned)OB;                           // bytes between x-adjacent outputs
    // fast-path lane offset: pixel (oh, w0 + n) of plane 4 g of the block (6 planes of output stay below 4 GiB)
    const unsigned ooff_g = (pix0 * (unsigned)a.out_cs + (unsigned)a.out_co) * (unsigned)OB + (unsigned)(4 * g) * (unsigned)out_plane;
    const bool full = w0 + C1_TW <= a.W && dbeg + brick_planes <= a.D && !a.skip && oh < a.Hh;

    for (int blk = 0; blk < a.nb; ++blk) {
        const int d0 = dbeg + blk * C1_P;
        if (d0 >= a.D) break;
        const unsigned char* sp = smem + blk * (C1_P * C1_PS * VB) + lane_off;
        c1_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            // step s = (q, tap): lane group g reads plane 4 (g >> 1) + 2 q + (g & 1) (C_in 8) or 4 (g >> 1) + q
            // (C_in 16); the lane-group part sits in lane_off, the rest is a compile-time immediate
            const int q = s / 9, t = s % 9;
            const int off = (((CIN == 8) ? 2 * q : q) * C1_PS + (t / 3) * C1_BW + (t % 3)) * VB;
            const uint4 x0 = *reinterpret_cast<const uint4*>(sp + off);
            const uint4 x1 = *reinterpret_cast<const uint4*>(sp + off + 16 * VB);
            acc0 = C1Mfma<H>::run(wf[s], x0, acc0);
            acc1 = C1Mfma<H>::run(wf[s], x1, acc1);
        }
        PSCV_STAMP(3)
        // ---- epilogue: lane (n, g) holds rows 4g..4g+3 = output planes d0 + 4g + r of pixel n ----
        if (full) {
            // interior brick without a skip tensor (the common case): straight-line code, two predicates in total
            // (the generic path below spends ~30 scalar instructions and four branches per 4-byte store)
            char* ob = reinterpret_cast<char*>(a.out) + ((unsigned long)b * a.D + d0) * out_plane;
            float y[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[0][r] = relu_floor(relu_floor(fmaf(acc0[r], e_scale, e_bias), lo_pre), lo_post);
                y[1][r] = relu_floor(relu_floor(fmaf(acc1[r], e_scale, e_bias), lo_pre), lo_post);
            }
            if (g < 2) {          // rows 0,1 (g = 0) and 4,5 (g = 1)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        char* o = ob + (unsigned long)r * out_plane + (ooff_g + ct * 16 * ostep);
                        if (a.out_f32) *reinterpret_cast<float*>(o) = y[ct][r];
                        else *reinterpret_cast<uint16_t*>(o) = Half16<H>::bits(y[ct][r]);
                    }
            }
            if (g == 0) {         // rows 2,3
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 2; r < 4; ++r) {
                        char* o = ob + (unsigned long)r * out_plane + (ooff_g + ct * 16 * ostep);
                        if (a.out_f32) *reinterpret_cast<float*>(o) = y[ct][r];
                        else *reinterpret_cast<uint16_t*>(o) = Half16<H>::bits(y[ct][r]);
                    }
            }
        } else if (row_ok) {
            char* ob = reinterpret_cast<char*>(a.out) + ((unsigned long)b * a.D + d0 + 4 * g) * out_plane;
            const char* sb = reinterpret_cast<const char*>(a.skip) + ((unsigned long)b * a.D + d0 + 4 * g) * skip_plane;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if (w0 + ct * 16 + n >= a.W) continue;
                const unsigned pix = pix0 + ct * 16;
                const unsigned ooff = (pix * (unsigned)a.out_cs + (unsigned)a.out_co) * (unsigned)OB;
                const unsigned soff = (pix * (unsigned)a.skip_cs + (unsigned)a.skip_co) * 2u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * g + r;
                    if (m >= C1_P || d0 + m >= a.D) continue;
                    float y = relu_floor(fmaf(ct ? acc1[r] : acc0[r], e_scale, e_bias), lo_pre);
                    if (a.skip) y += Half16<H>::one(*reinterpret_cast<const uint16_t*>(sb + r * skip_plane + soff));
                    y = relu_floor(y, lo_post);
                    if (a.out_f32) *reinterpret_cast<float*>(ob + r * out_plane + ooff) = y;
                    else *reinterpret_cast<uint16_t*>(ob + r * out_plane + ooff) = Half16<H>::bits(y);
                }
            }
        }
        PSCV_STAMP(4)
    }
    PSCV_STAMP_WAIT(5)
    PSCV_PROF_END(c1, blockIdx.x)
}

// ---- depth-sweep variant (C_in = 8, long depth axes: MVSNet's prob head at D = 192) ---------------------------------------
// The kernel above stages 14 planes, waits, computes 12, stores, and leaves: its workgroup lifetime (13 K cycles, three resident per
// CU) is a chain of exposed phases.  Here a workgroup keeps its 4 x 32 pixel tile and walks a depth chunk in 6-plane blocks over a
// 16-slot LDS plane ring (slot = plane & 15, 53 KB): block k reads planes 6k .. 6k+7 (relative to the chunk's first halo plane) while
// the six planes of block k+1 -- requested one block earlier with raw buffer loads (hardware zero fill outside the image / volume)
// -- are written into slots nobody reads; every input plane is fetched once per chunk (14/12 before), the A fragments once per chunk.
// A lane group reads two planes per block (4 (g >> 1) + (g & 1) and + 2): two ring addresses per block, taps as immediates.
constexpr int C1S_NSLOT = 16;

template <typename H, int FUSE>      // FUSE 1 / 2: also keep the softmax statistics of a fused tail (its registers cost the plain head a wave per SIMD)
__global__ __launch_bounds__(256, 3) void conv3d_c1_sweep_kernel(const C1Args a) {
    constexpr int VB = 16, NSTEPS = 18, PV = C1_BH * C1_BW, PSB = C1_PS * VB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [16 slots][C1_PS][16 B]

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q_ = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * C1_TH, w0 = twi * C1_TW;
    const int nbk = a.nb;                                // 6-plane blocks per chunk
    const int dbeg = dci * nbk * C1_P;

    const int tid = threadIdx.x;
    PSCV_PROF_BEGIN   // (profile builds: slots = prologue | MFMA + LDS reads | epilogue | stash + fetch | barrier)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;

    // ---- staging: thread t < 204 owns voxel t of every plane ----
    const unsigned long plane_stride_b = (unsigned long)a.Hh * a.W * a.in_cs * 2;
    const unsigned plane_bytes = (unsigned)(plane_stride_b - (unsigned long)a.in_co * 2);
    const char* inb = reinterpret_cast<const char*>(a.in) + ((unsigned long)b * a.D * plane_stride_b + (unsigned long)a.in_co * 2);
    const int sbh = tid / C1_BW, sbw = tid - sbh * C1_BW;
    const int sgh = h0 - 1 + sbh, sgw = w0 - 1 + sbw;
    const bool vox_ok = tid < PV && (unsigned)sgh < (unsigned)a.Hh && (unsigned)sgw < (unsigned)a.W;
    const unsigned goff = vox_ok ? (unsigned)(sgh * a.W + sgw) * (unsigned)(a.in_cs * 2) : 0x7ffffff0u;
    auto fetch = [&](int plane) -> uint4 {
        const bool pv = plane >= 0 && plane < a.D;                                               // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(inb + (unsigned long)(pv ? plane : 0) * plane_stride_b), (short)0, pv ? (int)plane_bytes : 0, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff, 0, 0));
    };
    auto stash = [&](int prel, const uint4& v) {          // prel = plane - (dbeg - 1)
        if (tid < PV) *reinterpret_cast<uint4*>(smem + (prel & (C1S_NSLOT - 1)) * PSB + tid * VB) = v;
    };

    uint4 wf[NSTEPS];
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) wf[s] = a.wpk[s * 64 + lane];
    {
        uint4 r8[C1_P + 2];
#pragma unroll
        for (int p = 0; p < C1_P + 2; ++p) r8[p] = fetch(dbeg - 1 + p);
#pragma unroll
        for (int p = 0; p < C1_P + 2; ++p) stash(p, r8[p]);
    }
    uint4 nx[C1_P];
#pragma unroll
    for (int p = 0; p < C1_P; ++p) nx[p] = fetch(dbeg + C1_P + 1 + p);        // planes of block 1 (prel 8 .. 13)

    const float e_scale = a.scale ? a.scale[0] : 1.0f, e_bias = a.bias ? a.bias[0] : 0.0f, e_floor = a.floor ? a.floor[0] : 0.0f;
    const float lo_pre = (a.epi & PSCV_EPI_RELU_PRE) ? e_floor : -__builtin_inff();
    const float lo_post = (a.epi & PSCV_EPI_RELU_POST) ? 0.0f : -__builtin_inff();
    // fused tail: the chunk's depth planes in LDS (a global load per block sat in the block's dependency chain: 59 us fused
    // against 40 us for the two separate launches)
    __shared__ float sdep[256];
    if (FUSE == 1 && tid < 256) sdep[tid] = a.depth[(long)b * a.depth_bstride + min(dbeg + tid, a.D - 1)];
    __syncthreads();
    PSCV_STAMP(0)

    const int inplane = (wave * C1_BW + n) * VB;
    const int pA = 4 * (g >> 1) + (g & 1);
    const int oh = h0 + wave;
    const int OB = a.out_f32 ? 4 : 2;
    const unsigned long out_plane = (unsigned long)a.Hh * a.W * a.out_cs * OB;
    const unsigned long skip_plane = (unsigned long)a.Hh * a.W * a.skip_cs * 2;
    const unsigned pix0 = (unsigned)(oh * a.W + w0 + n);
    const bool store = FUSE != 2 || a.out != nullptr;        // (the Vis pair head may keep its logits to itself)
    const bool row_ok = g < 2 && oh < a.Hh && store;
    const unsigned ostep = (unsigned)a.out_cs * (unsigned)OB;
    const unsigned ooff_g = (pix0 * (unsigned)a.out_cs + (unsigned)a.out_co) * (unsigned)OB + (unsigned)(4 * g) * (unsigned)out_plane;
    const bool tile_full = w0 + C1_TW <= a.W && !a.skip && oh < a.Hh && store;
    // running softmax statistics of this lane's own logits (planes d0 + 4 g + r of pixels (oh, w0 + ct * 16 + n)), merged across the two
    // lane groups once per chunk
    float sM[2] = {-__builtin_inff(), -__builtin_inff()}, sZ[2] = {0.f, 0.f}, sD[2] = {0.f, 0.f}, sI[2] = {0.f, 0.f};

    for (int k = 0; k < nbk; ++k) {
        const int d0 = dbeg + k * C1_P;
        if (d0 >= a.D) break;
        const unsigned char* spA = smem + ((C1_P * k + pA) & (C1S_NSLOT - 1)) * PSB + inplane;
        const unsigned char* spB = smem + ((C1_P * k + pA + 2) & (C1S_NSLOT - 1)) * PSB + inplane;
        // four accumulation chains of 9 dependent MFMAs (two column tiles x the two plane sets) instead of two of 18: the block is a
        // latency chain (one workgroup per CU still needs ~3.5 K cycles per block), so the chain length counts
        c1_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc0b = {0.f, 0.f, 0.f, 0.f}, acc1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = ((t / 3) * C1_BW + (t % 3)) * VB;
            const uint4 x0 = *reinterpret_cast<const uint4*>(spA + off);
            const uint4 x1 = *reinterpret_cast<const uint4*>(spA + off + 16 * VB);
            const uint4 x2 = *reinterpret_cast<const uint4*>(spB + off);
            const uint4 x3 = *reinterpret_cast<const uint4*>(spB + off + 16 * VB);
            acc0 = C1Mfma<H>::run(wf[t], x0, acc0);
            acc1 = C1Mfma<H>::run(wf[t], x1, acc1);
            acc0b = C1Mfma<H>::run(wf[9 + t], x2, acc0b);
            acc1b = C1Mfma<H>::run(wf[9 + t], x3, acc1b);
        }
        acc0 += acc0b;
        acc1 += acc1b;
        PSCV_STAMP(1)
        // ---- epilogue: lane (n, g) holds rows 4g..4g+3 = output planes d0 + 4g + r of pixel n ----
        if (tile_full && d0 + C1_P <= a.D) {
            char* ob = reinterpret_cast<char*>(a.out) + ((unsigned long)b * a.D + d0) * out_plane;
            float y[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[0][r] = clamp_lo(clamp_lo(fmaf(acc0[r], e_scale, e_bias), lo_pre), lo_post);
                y[1][r] = clamp_lo(clamp_lo(fmaf(acc1[r], e_scale, e_bias), lo_pre), lo_post);
            }
            if (g < 2) {          // rows 0,1 (g = 0) and 4,5 (g = 1)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        char* o = ob + (unsigned long)r * out_plane + (ooff_g + ct * 16 * ostep);
                        if (a.out_f32) *reinterpret_cast<float*>(o) = y[ct][r];
                        else *reinterpret_cast<uint16_t*>(o) = Half16<H>::bits(y[ct][r]);
                    }
            }
            if (g == 0) {         // rows 2,3
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 2; r < 4; ++r) {
                        char* o = ob + (unsigned long)r * out_plane + (ooff_g + ct * 16 * ostep);
                        if (a.out_f32) *reinterpret_cast<float*>(o) = y[ct][r];
                        else *reinterpret_cast<uint16_t*>(o) = Half16<H>::bits(y[ct][r]);
                    }
            }
        } else if (row_ok) {
            char* ob = reinterpret_cast<char*>(a.out) + ((unsigned long)b * a.D + d0 + 4 * g) * out_plane;
            const char* sb = reinterpret_cast<const char*>(a.skip) + ((unsigned long)b * a.D + d0 + 4 * g) * skip_plane;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if (w0 + ct * 16 + n >= a.W) continue;
                const unsigned pix = pix0 + ct * 16;
                const unsigned ooff = (pix * (unsigned)a.out_cs + (unsigned)a.out_co) * (unsigned)OB;
                const unsigned soff = (pix * (unsigned)a.skip_cs + (unsigned)a.skip_co) * 2u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * g + r;
                    if (m >= C1_P || d0 + m >= a.D) continue;
                    float y = relu_floor(fmaf(ct ? acc1[r] : acc0[r], e_scale, e_bias), lo_pre);
                    if (a.skip) y += Half16<H>::one(*reinterpret_cast<const uint16_t*>(sb + r * skip_plane + soff));
                    y = relu_floor(y, lo_post);
                    if (a.out_f32) *reinterpret_cast<float*>(ob + r * out_plane + ooff) = y;
                    else *reinterpret_cast<uint16_t*>(ob + r * out_plane + ooff) = Half16<H>::bits(y);
                }
            }
        }
        if (FUSE && g < 2) {
            // fused tail: fold the block's logits (fp32, before any store rounding) into the lane's softmax statistics
            float dpl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dpl[r] = FUSE == 1 ? sdep[min(k * C1_P + 4 * g + r, 255)] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float yv[4];
                bool ok[4];
                float lmax = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * g + r;
                    ok[r] = m < C1_P && d0 + m < a.D;
                    yv[r] = clamp_lo(clamp_lo(fmaf(ct ? acc1[r] : acc0[r], e_scale, e_bias), lo_pre), lo_post);
                    if (ok[r]) lmax = fmaxf(lmax, yv[r]);
                }
                if (lmax > -__builtin_inff()) {
                    const float mn = fmaxf(sM[ct], lmax);
                    const float sc = sM[ct] > -__builtin_inff() ? __expf(sM[ct] - mn) : 0.0f;
                    float z = sZ[ct] * sc, si = sI[ct] * sc;
                    // FUSE 1: sum e * depth plane.  FUSE 2: sum e * (logit - max), re-based when the max moves (entropy = log Z - that / Z:
                    // both terms of the entropy's size, where max + log Z - E[logit] would cancel two large numbers)
                    float sd = FUSE == 1 ? sD[ct] * sc : (sM[ct] > -__builtin_inff() ? sc * fmaf(sM[ct] - mn, sZ[ct], sD[ct]) : 0.0f);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ok[r]) {
                            const float e = __expf(yv[r] - mn);
                            z += e;
                            sd = fmaf(e, FUSE == 1 ? dpl[r] : yv[r] - mn, sd);
                            si = fmaf(e, (float)(d0 + 4 * g + r), si);
                        }
                    }
                    sM[ct] = mn; sZ[ct] = z; sD[ct] = sd; sI[ct] = si;
                }
            }
        }
        PSCV_STAMP(2)
        // the planes of block k + 1 into the slots nobody reads now; their registers are refilled with the planes of block k + 2
        if (k + 1 < nbk) {
#pragma unroll
            for (int p = 0; p < C1_P; ++p) stash(C1_P * k + C1_P + 2 + p, nx[p]);
#pragma unroll
            for (int p = 0; p < C1_P; ++p) nx[p] = fetch(dbeg + C1_P * (k + 2) + 1 + p);
            __builtin_amdgcn_sched_barrier(0);
        }
        PSCV_STAMP(3)
        __syncthreads();
        PSCV_STAMP(4)
    }
    if (FUSE) {
        // merge the lane groups g = 0 (rows 0..3) and g = 1 (rows 4, 5) of every pixel: lane n + 16 -> lane n; lanes g = 0 write
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float m1 = __shfl_down(sM[ct], 16, 64), z1 = __shfl_down(sZ[ct], 16, 64);
            const float d1 = __shfl_down(sD[ct], 16, 64), i1 = __shfl_down(sI[ct], 16, 64);
            const float mn = fmaxf(sM[ct], m1);
            const float f0 = sM[ct] > -__builtin_inff() ? __expf(sM[ct] - mn) : 0.0f, f1 = m1 > -__builtin_inff() ? __expf(m1 - mn) : 0.0f;
            const int ow = w0 + ct * 16 + n;
            if (g == 0 && oh < a.Hh && ow < a.W) {
                const long hw = (long)a.Hh * a.W;
                const float Z = sZ[ct] * f0 + z1 * f1, SI = sI[ct] * f0 + i1 * f1;
                float SD;
                if (FUSE == 1) SD = sD[ct] * f0 + d1 * f1;
                else SD = (f0 > 0.0f ? f0 * fmaf(sM[ct] - mn, sZ[ct], sD[ct]) : 0.0f) + (f1 > 0.0f ? f1 * fmaf(m1 - mn, z1, d1) : 0.0f);
                if (FUSE == 2 && a.ndc == 1) {
                    const long o = (long)b * hw + (long)oh * a.W + ow;
                    const float inv = 1.0f / Z;
                    a.o_index[o] = SI * inv;
                    a.o_entropy[o] = __logf(Z) - SD * inv;
                } else {
                    float* pp = a.part + (((long)b * a.ndc + dci) * 4) * hw + (long)oh * a.W + ow;
                    pp[0] = mn;
                    pp[hw] = Z;
                    pp[2 * hw] = SD;
                    pp[3 * hw] = SI;
                }
            }
        }
    }
    PSCV_PROF_END(c1, blockIdx.x)
}

// Merge of the per-chunk softmax partials written by the sweep above + the 4-plane photometric confidence from the fp32 logits:
// depth = sum p_d depth_d, confidence = sum of p over planes i-1 .. i+2 around i = trunc(E[index]) (zero padded).
// Replaces (fdarmon/wild_deep_mvs): F.softmax + depth_regression + confidence, models/MVSNet/model.py:207-215.
__global__ __launch_bounds__(256) void softargmin_merge_kernel(const float* __restrict__ part, const float* __restrict__ logits, int ndc, int B,
                                                                int D, long hw, float* __restrict__ o_depth, float* __restrict__ o_conf) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)B * hw) return;
    const int b = (int)(p / hw);
    const long pf = p - (long)b * hw;
    const float* pp = part + ((long)b * ndc * 4) * hw + pf;
    float M = -__builtin_inff();
    for (int c = 0; c < ndc; ++c) M = fmaxf(M, pp[(long)c * 4 * hw]);
    float Z = 0.f, SD = 0.f, SI = 0.f;
    for (int c = 0; c < ndc; ++c) {
        const float* q = pp + (long)c * 4 * hw;
        const float f = expf(q[0] - M);
        Z = fmaf(q[hw], f, Z); SD = fmaf(q[2 * hw], f, SD); SI = fmaf(q[3 * hw], f, SI);
    }
    const float inv = 1.0f / Z;
    if (o_depth) o_depth[p] = SD * inv;
    if (o_conf) {
        const int i = (int)(SI * inv);
        const float* lp = logits + (long)b * D * hw + pf;
        float c = 0.f;
        for (int k = -1; k <= 2; ++k) {
            const int d = i + k;
            if (d >= 0 && d < D) c += expf(lp[(long)d * hw] - M) * inv;
        }
        o_conf[p] = c;
    }
}


// Merge of the per-chunk partials (max, Z, sum e (logit - max), sum e index) of the Vis pair head: expected index and entropy.
__global__ __launch_bounds__(256) void index_entropy_merge_kernel(const float* __restrict__ part, int ndc, int B, long hw,
                                                                  float* __restrict__ o_index, float* __restrict__ o_entropy) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)B * hw) return;
    const int b = (int)(p / hw);
    const long pf = p - (long)b * hw;
    const float* pp = part + ((long)b * ndc * 4) * hw + pf;
    float M = -__builtin_inff();
    for (int c = 0; c < ndc; ++c) M = fmaxf(M, pp[(long)c * 4 * hw]);
    float Z = 0.f, S = 0.f, SI = 0.f;
    for (int c = 0; c < ndc; ++c) {
        const float* q = pp + (long)c * 4 * hw;
        if (q[0] == -__builtin_inff()) continue;
        const float f = expf(q[0] - M);
        Z = fmaf(q[hw], f, Z);
        S += f * fmaf(q[0] - M, q[hw], q[2 * hw]);
        SI = fmaf(q[3 * hw], f, SI);
    }
    const float inv = 1.0f / Z;
    o_index[p] = SI * inv;
    o_entropy[p] = logf(Z) - S * inv;
}

template <typename H, int CIN>
static int c1_launch(const C1Args& a, long nblk, hipStream_t st) {
    auto kern = conv3d_c1_kernel<H, CIN>;
    const size_t lds = (size_t)(a.nb * C1_P + 2) * C1_PS * CIN * 2;
    if (lds > 60000) {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) { set_error("pscv_conv3d(c1): hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
    return 0;
}

}  // namespace pscv

pscv::Knob g_c1_nb = {0, pscv::KNOB_C1_NB};
pscv::Knob g_c1_sweep = {1, pscv::KNOB_C1_SWEEP};   // pscv_set_tuning("c1_sweep", 0): always the brick variant; 2: the depth sweep at any depth
PSCV_PROF_EXPORT(c1)

// depth / part / merged outputs non-null: the fused tail (pscv_prob_softargmin); returns 1 when the layer does not get the depth-sweep
// variant (the caller then runs the two separate entry points), 0 on success, < 0 on error
static int c1_dispatch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed,
                       const float* scale, const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                       int out_cstride, int out_coff, int out_dtype, int B, int D, int Hh, int W, int c_in,
                       int epi_flags, hipStream_t st, const float* depth, long depth_bstride, float* part, long part_floats,
                       float* o_depth, float* o_conf, float* o_index = nullptr, float* o_entropy = nullptr) {
    using namespace pscv;
    C1Args a;
    a.depth = depth; a.depth_bstride = depth_bstride; a.part = nullptr;
    a.o_index = o_index; a.o_entropy = o_entropy;
    const bool ie = o_index != nullptr;       // the Vis pair head: index + entropy (FUSE = 2)
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
    const int nblocks = (D + C1_P - 1) / C1_P;
    // two 6-plane blocks per workgroup (the 2-plane halo of the brick costs 1/6 instead of 1/3) when that still leaves
    // the chip a few rounds of workgroups; C_in = 16 keeps one (LDS: 53 KB per workgroup instead of 93)
    a.nb = (c_in == 8 && tiles * ((nblocks + 1) / 2) >= 1024) ? 2 : 1;
    if (g_c1_nb) a.nb = g_c1_nb;
    // long depth axes with 8 input channels: the depth-sweep variant, chunks sized for about one resident round (3 per CU)
    bool sweep = false;
    if (c_in == 8 && g_c1_sweep && (long)Hh * W * in_cstride * 2 < 0x7fffffffL) {
        const long want = tiles >= 768 ? 1 : 768 / tiles;
        const int ndc = (int)(want < nblocks ? want : nblocks);
        int nbk = (nblocks + ndc - 1) / ndc;
        if (g_c1_nb > 2) nbk = g_c1_nb < nblocks ? g_c1_nb : nblocks;     // (measurement override: blocks per chunk)
        if (nbk >= 3 || g_c1_sweep == 2) { sweep = true; a.nb = nbk; }
    }
    a.ndc = (nblocks + a.nb - 1) / a.nb;
    const long nblk = tiles * a.ndc;
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(c1): bad grid %ld", nblk); return -1; }
    if (part) {
        if (!sweep || skip || out_dtype != PSCV_F32 || out_cstride != 1 || (!ie && a.nb * C1_P > 256)) return 1;
        if ((long)B * a.ndc * 4 * Hh * W > part_floats) { set_error("pscv_prob_softargmin: workspace of %ld floats is too small", part_floats); return -1; }
        a.part = part;
    }
    if (sweep) {
        const size_t lds = (size_t)C1S_NSLOT * C1_PS * 16;
        if (part && ie) {
            if (dtype == PSCV_BF16) hipLaunchKernelGGL((conv3d_c1_sweep_kernel<bf16_t, 2>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else hipLaunchKernelGGL((conv3d_c1_sweep_kernel<f16_t, 2>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            if (a.ndc > 1) {
                const long npix = (long)B * Hh * W;
                hipLaunchKernelGGL(index_entropy_merge_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, part, a.ndc, B, (long)Hh * W,
                                   o_index, o_entropy);
            }
            return 0;
        }
        if (part) {
            if (dtype == PSCV_BF16) hipLaunchKernelGGL((conv3d_c1_sweep_kernel<bf16_t, 1>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else hipLaunchKernelGGL((conv3d_c1_sweep_kernel<f16_t, 1>), dim3((unsigned)nblk), dim3(256), lds, st, a);
        } else {
            if (dtype == PSCV_BF16) hipLaunchKernelGGL((conv3d_c1_sweep_kernel<bf16_t, 0>), dim3((unsigned)nblk), dim3(256), lds, st, a);
            else hipLaunchKernelGGL((conv3d_c1_sweep_kernel<f16_t, 0>), dim3((unsigned)nblk), dim3(256), lds, st, a);
        }
        if (part) {
            const long npix = (long)B * Hh * W;
            hipLaunchKernelGGL(softargmin_merge_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, part, reinterpret_cast<const float*>(out),
                               a.ndc, B, D, (long)Hh * W, o_depth, o_conf);
        }
        return 0;
    }
    if (dtype == PSCV_BF16 && c_in == 8) return c1_launch<bf16_t, 8>(a, nblk, st);
    if (dtype == PSCV_BF16 && c_in == 16) return c1_launch<bf16_t, 16>(a, nblk, st);
    if (dtype == PSCV_F16 && c_in == 8) return c1_launch<f16_t, 8>(a, nblk, st);
    if (dtype == PSCV_F16 && c_in == 16) return c1_launch<f16_t, 16>(a, nblk, st);
    set_error("pscv_conv3d(c1): c_in=%d dtype=%d not supported (c_in 8 or 16)", c_in, dtype);
    return -1;
}

// merge launch for other producers of the same partials (conv3d_tail.hip)
void pscv_softargmin_merge_launch(const float* part, const float* logits, int ndc, int B, int D, long hw, float* o_depth, float* o_conf, hipStream_t st) {
    const long npix = (long)B * hw;
    hipLaunchKernelGGL(pscv::softargmin_merge_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, part, logits, ndc, B, D, hw, o_depth, o_conf);
}

int pscv_conv3d_c1_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed,
                          const float* scale, const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                          int out_cstride, int out_coff, int out_dtype, int B, int D, int Hh, int W, int c_in,
                          int epi_flags, hipStream_t st) {
    return c1_dispatch(in, dtype, in_cstride, in_coff, packed, scale, bias, floor, skip, skip_cstride, skip_coff, out, out_cstride, out_coff,
                       out_dtype, B, D, Hh, W, c_in, epi_flags, st, nullptr, 0, nullptr, 0, nullptr, nullptr);
}

extern "C" long pscv_prob_softargmin_workspace(int B, int D, int H, int W) {
    return (long)B * ((D + pscv::C1_P - 1) / pscv::C1_P) * 4 * H * W;     // floats: one partial set per 6-plane block at most
}

extern "C" int pscv_prob_softargmin(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                                    const float* bias, const float* floor, int c_in, int epi_flags, const float* depth, long depth_bstride,
                                    float* logits, float* workspace, long workspace_floats, float* out_depth, float* out_conf, int B, int D,
                                    int H, int W, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(in && packed && depth && logits && workspace && out_depth, "pscv_prob_softargmin: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pscv_prob_softargmin: bad sizes");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_prob_softargmin: storage dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(in_cstride % 8 == 0 && in_coff % 8 == 0 && in_coff + c_in <= in_cstride, "pscv_prob_softargmin: input channel slice must be 8-aligned");
    const int rc = c1_dispatch(in, dtype, in_cstride, in_coff, packed, scale, bias, floor, nullptr, 0, 0, logits, 1, 0, PSCV_F32, B, D, H, W, c_in,
                               epi_flags, reinterpret_cast<hipStream_t>(stream), depth, depth_bstride, workspace, workspace_floats, out_depth,
                               out_conf);
    if (rc < 0) return rc;
    if (rc == 1) { set_error("pscv_prob_softargmin: this layer does not run the depth-sweep head (needs c_in = 8 and a depth axis of >= 3 six-plane blocks per chunk); use pscv_conv3d + pscv_softargmin"); return -3; }
    PSCV_CHECK_LAUNCH("pscv_prob_softargmin");
    return 0;
}

extern "C" int pscv_head_index_entropy(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                                       const float* bias, const float* floor, int c_in, int epi_flags, float* logits, float* workspace,
                                       long workspace_floats, float* out_index, float* out_entropy, int B, int D, int H, int W, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(in && packed && workspace && out_index && out_entropy, "pscv_head_index_entropy: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "pscv_head_index_entropy: bad sizes");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_head_index_entropy: storage dtype %d must be bf16 or fp16", dtype);
    PSCV_CHECK_ARG(in_cstride % 8 == 0 && in_coff % 8 == 0 && in_coff + c_in <= in_cstride, "pscv_head_index_entropy: input channel slice must be 8-aligned");
    const int rc = c1_dispatch(in, dtype, in_cstride, in_coff, packed, scale, bias, floor, nullptr, 0, 0, logits, 1, 0, PSCV_F32, B, D, H, W, c_in,
                               epi_flags, reinterpret_cast<hipStream_t>(stream), nullptr, 0, workspace, workspace_floats, nullptr, nullptr,
                               out_index, out_entropy);
    if (rc < 0) return rc;
    if (rc == 1) { set_error("pscv_head_index_entropy: this layer does not run the depth-sweep head (needs c_in = 8 and a depth axis of >= 3 six-plane blocks per chunk); use pscv_conv3d + pscv_softargmin"); return -3; }
    PSCV_CHECK_LAUNCH("pscv_head_index_entropy");
    return 0;
}
