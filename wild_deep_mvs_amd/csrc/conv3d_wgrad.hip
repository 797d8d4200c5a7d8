// Weight gradient of the 3x3x3 convolutions as an MFMA contraction over VOXELS (gfx950, wave64).
//
//     G[a][b][t] = sum_{n, o}  P[n, o, a] * Q[n, s*o + t - 1, b]          t = (tz, ty, tx) in 0..2^3, zero outside Q
//
//   Conv3d (stride s = 1 or 2), weight [C_out, C_in, 27]:          P = grad of the output (a = c_out), Q = the layer input
//   ConvTranspose3d (stride s, padding 1), weight [C_in, C_out, 27]: P = the layer input (a = c_in),  Q = grad of the output
// i.e. the backward-to-weights of every block of the reference's 3-D U-Nets (models/MVSNet/model.py:43-84,
// models/CVP_MVSNet/models/net.py:50-85, models/VisMVSNet/nn_utils.py:194-278) under loss.backward().
//
// GEMM view per tap: D[16 a x 16 b] += A[16 a x 32 k] * B[32 k x 16 b] with k = 32 voxels of the tile
// (v_mfma_f32_16x16x32: a lane group holds 8 CONSECUTIVE k = 8 x-adjacent voxels of one channel).  The tensors are
// channels-last, so both operands are transposed on their way into LDS ([channel][voxel], 16-bit); Q is staged once per
// x-tap, already shifted (and, for s = 2, decimated), so every fragment is one aligned ds_read_b128 and the k-loop is
// identical for both strides.  Channel strides are padded to an odd number of 16-byte slots: conflict-free reads.
// A workgroup owns one 16-channel a-tile and NB 16-channel b-tiles (blockIdx.y) and walks voxel tiles persistently
// (blockIdx.x, grid-stride); its 4 waves split the 27 taps, accumulators stay in registers across tiles.  Partial sums go
// to a workspace [nblk][27][CA16][CB16] and a finishing kernel adds them in a fixed order: bit-reproducible, no atomics.
#include <stdlib.h>
#include "pscv_common.h"

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 wg_f16x8;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;

template <typename H> struct WMfma;
template <> struct WMfma<bf16_t> {
    __device__ static __forceinline__ wg_f32x4 run(const uint4& a, const uint4& b, const wg_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, a), __builtin_bit_cast(wg_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WMfma<f16_t> {
    __device__ static __forceinline__ wg_f32x4 run(const uint4& a, const uint4& b, const wg_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wg_f16x8, a), __builtin_bit_cast(wg_f16x8, b), c, 0, 0, 0);
    }
};

#ifdef PSCV_ABLATE
extern Knob g_fuse_c0;      // warp_cost.hip
#endif
struct WgradArgs {
    const uint16_t* p;
    const uint16_t* q;
    float* part;        // [gridDim.x][27][ca16][cb16]
    int p_cs, p_co, q_cs, q_co;
    int ca, cb, ca16, cb16;
    int B, Dp, Hp, Wp, Dq, Hq, Wq;
    int ntz, nty, ntx, ntiles;
    int abl;            // -DPSCV_ABLATE builds: measurement flags from pscv_set_tuning("fuse_c0", bits): 1 no P staging, 2 no Q loads, 4 no Q LDS writes, 8 no MFMA loop
};
#ifdef PSCV_ABLATE
#define WG_ABL(bit) (a.abl & (bit))
#else
#define WG_ABL(bit) false
#endif

// P2D (TZ = 1): single-plane volumes, i.e. the 2-D layers of the feature extractors run as D = 1: the z taps 0 and 2 only ever meet
// zero padding, so one Q plane is staged, the waves split the 9 (ty, tx) taps of kernel slice tz = 1, and the tile is 8 rows of
// one plane instead of 4 rows of two (with D = 1 half of a two-plane tile is padding): 6x fewer MFMAs per pixel.
template <int S, int TZ, int TY, int NB, int NA = 1> struct WgGeom {
    static constexpr bool P2D = TZ == 1;
    static constexpr int NV = TZ * TY * 16;                 // P voxels per tile
    static constexpr int QZ = P2D ? 1 : S * (TZ - 1) + 3, QY = S * (TY - 1) + 3, QX = S * 15 + 3;
    static constexpr int PAS = NV * 2 + 16;                 // bytes per P channel row (odd number of 16-byte slots)
    // stride 1 (round 3): ONE copy of Q, rows of QXP = 24 elements (18 used); the fragment of x-tap tx = elements 8 x8 + tx .. + 7 of a
    // row is cut out of two aligned 16-byte reads with byte-align ops (tx = 1) or is a register renaming (tx = 2).  Stride 2 keeps the
    // three pre-shifted, decimated copies.
    static constexpr bool ONE = S == 1;
    static constexpr int QXP = 24;
    static constexpr int QBS = ONE ? QZ * QY * QXP * 2 + 16 : QZ * QY * 32 + 16;   // bytes per Q channel block
    static constexpr int QTS = 16 * NB * QBS;               // bytes per copy of Q
    static constexpr int P_BYTES = 16 * NA * PAS;           // NA 16-channel a-tiles per workgroup (single-plane mode: 4 when c_a = 64)
    static constexpr int LDS = P_BYTES + (ONE ? 1 : 3) * QTS;
    static constexpr int KSTEPS = TZ * TY * 2 / 4;
    static_assert((TZ * TY * 2) % 4 == 0, "tile must hold whole k-steps");
    static_assert((PAS / 16) % 2 == 1 && (QBS / 16) % 2 == 1, "odd slot strides");
};

template <typename H, int S, int TZ, int TY, int NB, int NA = 1>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    using G = WgGeom<S, TZ, TY, NB, NA>;
    static_assert(NA == 1 || (G::ONE && G::P2D), "several a-tiles per workgroup: single-plane stride-1 mode only (accumulator budget)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* p_lds = smem;
    unsigned char* q_lds = smem + G::P_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nbg = a.cb16 / (16 * NB);
    const int at = blockIdx.y / nbg, bg = blockIdx.y % nbg;
    const int a0 = at * 16 * NA, b0 = bg * 16 * NB;

    wg_f32x4 acc[7][NB];
#pragma unroll
    for (int ti = 0; ti < 7; ++ti)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[ti][nb] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    // stride 1: a wave owns whole (tz, ty, b-tile) units -- unit u = wave + 4 i -> pair u / NB, b-tile u % NB -- and runs their three
    // x-taps off ONE pair of aligned reads (two ds_read_b128 + four byte-align ops per three MFMAs)
    constexpr int NPAIR = G::P2D ? 3 : 9, NU = NPAIR * NB, NUW = (NU + 3) / 4;
    // (single-plane mode with 64 a-channels: NA = 4 a-tiles share ONE staging of Q -- as (a-tile, b-group) workgroups every Q tile was
    //  transposed four times)
    wg_f32x4 accu[NA][G::ONE ? NUW : 1][3];
#pragma unroll
    for (int m = 0; m < NA; ++m)
#pragma unroll
        for (int i = 0; i < (G::ONE ? NUW : 1); ++i)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) accu[m][i][tx] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx_i = t % a.ntx; t /= a.ntx;
        const int ty_i = t % a.nty; t /= a.nty;
        const int tz_i = t % a.ntz; t /= a.ntz;
        const int b = t;
        const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * 16;

        // ---- stage P, transposed: p_lds[channel][voxel] ----
        {
            const uint16_t* pb = a.p + (long)b * a.Dp * a.Hp * a.Wp * a.p_cs + a.p_co;
            constexpr int NCH = G::NV * 2 * NA;
            for (int c = tid; c < (WG_ABL(1) ? 0 : NCH); c += 256) {
                const int vox = c % G::NV, c8 = c / G::NV;
                const int xl = vox & 15, yl = (vox >> 4) % TY, zl = (vox >> 4) / TY;
                const int gz = z0 + zl, gy = y0 + yl, gx = x0 + xl;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (gz < a.Dp && gy < a.Hp && gx < a.Wp && a0 + c8 * 8 < a.ca)
                    v = *reinterpret_cast<const uint4*>(pb + (((long)gz * a.Hp + gy) * a.Wp + gx) * a.p_cs + a0 + c8 * 8);
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                unsigned char* dst = p_lds + (c8 * 8) * G::PAS + vox * 2;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<uint16_t*>(dst + j * G::PAS) = (uint16_t)(w4[j >> 1] >> ((j & 1) * 16));
            }
        }
        if constexpr (G::ONE) {
            // ---- stage Q (stride 1), transposed, ONE copy: q_lds[channel][qz][qy][xe], xe <-> gx = x0 - 1 + xe.  A thread takes FOUR
            // x-adjacent voxels of one 8-channel chunk (four 16-byte loads), transposes the 4 x 8 block in registers (16 v_perm_b32) and
            // writes eight 8-byte runs: 2 LDS write instructions per 16 bytes of Q (the per-tap 2-byte stores of rounds 1-2 took 24, and
            // were 45 % of the kernel: scripts/dev/wgrad_ablate.py) ----
            const uint16_t* qb = a.q + (long)b * a.Dq * a.Hq * a.Wq * a.q_cs + a.q_co;
            constexpr int CCH = 2 * NB, GPR = 5;                 // 8-channel chunks of this block's b slice; 4-voxel groups per row (20 >= 18)
            constexpr int NG = G::QZ * G::QY * GPR * CCH;
            constexpr int GB = 2;                                // groups in flight per thread
            const int oz = G::P2D ? z0 : z0 - 1, oy = y0 - 1, ox0 = x0 - 1;
            for (int g0 = 0; g0 < NG; g0 += 256 * GB) {
                uint4 val[GB][4];
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const int gi = g0 + k * 256 + tid;
                    const int cc = gi % CCH, r = gi / CCH;
                    const int xg = r % GPR, row = r / GPR;
                    const int yl = row % G::QY, zl = row / G::QY;
                    const int gz = oz + zl, gy = oy + yl;
                    const bool rok = gi < NG && (unsigned)gz < (unsigned)a.Dq && (unsigned)gy < (unsigned)a.Hq && b0 + cc * 8 < a.cb && !WG_ABL(2);
                    const uint16_t* rp = qb + (((long)gz * a.Hq + gy) * a.Wq) * a.q_cs + b0 + cc * 8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gx = ox0 + xg * 4 + e;
                        val[k][e] = make_uint4(0u, 0u, 0u, 0u);
                        if (rok && (unsigned)gx < (unsigned)a.Wq) val[k][e] = *reinterpret_cast<const uint4*>(rp + (long)gx * a.q_cs);
                    }
                }
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const int gi = g0 + k * 256 + tid;
                    if (gi >= NG || WG_ABL(4)) continue;
                    const int cc = gi % CCH, r = gi / CCH;
                    const int xg = r % GPR, row = r / GPR;
                    unsigned char* dst = q_lds + (cc * 8) * G::QBS + (row * G::QXP + xg * 4) * 2;
                    const uint32_t w0[4] = {val[k][0].x, val[k][0].y, val[k][0].z, val[k][0].w};
                    const uint32_t w1[4] = {val[k][1].x, val[k][1].y, val[k][1].z, val[k][1].w};
                    const uint32_t w2[4] = {val[k][2].x, val[k][2].y, val[k][2].z, val[k][2].w};
                    const uint32_t w3[4] = {val[k][3].x, val[k][3].y, val[k][3].z, val[k][3].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {                 // channel j = word j >> 1, half j & 1 of every voxel
                        const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
                        const uint32_t lo = __builtin_amdgcn_perm(w1[j >> 1], w0[j >> 1], sel);
                        const uint32_t hi = __builtin_amdgcn_perm(w3[j >> 1], w2[j >> 1], sel);
                        *reinterpret_cast<uint2*>(dst + j * G::QBS) = make_uint2(lo, hi);
                    }
                }
            }
        } else {
        // ---- stage Q (stride 2), transposed, once per x-tap: q_lds[tx][channel][qz][qy][ox] = Q[.., S*(x0+ox) + tx - 1] ----
        {
            const uint16_t* qb = a.q + (long)b * a.Dq * a.Hq * a.Wq * a.q_cs + a.q_co;
            constexpr int CCH = 2 * NB;                         // 8-channel chunks of this block's b slice
            constexpr int NCH = G::QZ * G::QY * G::QX * CCH;
            constexpr int BATCH = 4;
            const int oz = G::P2D ? S * z0 : S * z0 - 1, oy = S * y0 - 1, ox0 = S * x0 - 1;
            for (int c0 = 0; c0 < NCH; c0 += 256 * BATCH) {
                uint4 val[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int c = c0 + k * 256 + tid;
                    const int cc = c % CCH, v = c / CCH;
                    const int xl = v % G::QX, r = v / G::QX;
                    const int yl = r % G::QY, zl = r / G::QY;
                    const int gz = oz + zl, gy = oy + yl, gx = ox0 + xl;
                    val[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (c < NCH && (unsigned)gz < (unsigned)a.Dq && (unsigned)gy < (unsigned)a.Hq && (unsigned)gx < (unsigned)a.Wq &&
                        b0 + cc * 8 < a.cb && !WG_ABL(2))
                        val[k] = *reinterpret_cast<const uint4*>(qb + (((long)gz * a.Hq + gy) * a.Wq + gx) * a.q_cs + b0 + cc * 8);
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int c = c0 + k * 256 + tid;
                    if (c >= NCH || WG_ABL(4)) continue;
                    const int cc = c % CCH, v = c / CCH;
                    const int xl = v % G::QX, r = v / G::QX;
                    const int yl = r % G::QY, zl = r / G::QY;
                    const uint32_t w4[4] = {val[k].x, val[k].y, val[k].z, val[k].w};
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const int d = xl - tx;
                        if (d < 0 || (S == 2 && (d & 1))) continue;
                        const int ox = d / S;
                        if (ox >= 16) continue;
                        unsigned char* dst = q_lds + tx * G::QTS + (cc * 8) * G::QBS + ((zl * G::QY + yl) * 16 + ox) * 2;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            *reinterpret_cast<uint16_t*>(dst + j * G::QBS) = (uint16_t)(w4[j >> 1] >> ((j & 1) * 16));
                    }
                }
            }
        }
        }
        __syncthreads();

        // ---- contraction: this wave's taps x all k-steps ----
#pragma unroll
        for (int ks = 0; ks < (WG_ABL(8) ? 0 : G::KSTEPS); ++ks) {
            const int chunk = ks * 4 + g;
            const int x8 = chunk & 1, yl = (chunk >> 1) % TY, zl = (chunk >> 1) / TY;
            const uint4 af = *reinterpret_cast<const uint4*>(p_lds + n * G::PAS + ((zl * TY + yl) * 16 + x8 * 8) * 2);
            if constexpr (G::ONE) {
#pragma unroll
                for (int i = 0; i < NUW; ++i) {
                    const int u = wave + 4 * i;
                    if (u < NU) {
                        const int pair = u / NB, nb = u % NB;
                        const int tz = G::P2D ? 1 : pair / 3, ty = pair % 3;
                        const int qz = G::P2D ? 0 : zl + tz;
                        const unsigned char* qrow = q_lds + (nb * 16 + n) * G::QBS + ((qz * G::QY + (yl + ty)) * G::QXP + x8 * 8) * 2;
                        const uint4 c0 = *reinterpret_cast<const uint4*>(qrow);
                        const uint4 c1 = *reinterpret_cast<const uint4*>(qrow + 16);
                        const uint4 b1 = make_uint4(__builtin_amdgcn_alignbyte(c0.y, c0.x, 2), __builtin_amdgcn_alignbyte(c0.z, c0.y, 2),
                                                    __builtin_amdgcn_alignbyte(c0.w, c0.z, 2), __builtin_amdgcn_alignbyte(c1.x, c0.w, 2));
                        const uint4 b2 = make_uint4(c0.y, c0.z, c0.w, c1.x);
#pragma unroll
                        for (int m = 0; m < NA; ++m) {
                            const uint4 afm = m == 0 ? af : *reinterpret_cast<const uint4*>(p_lds + (m * 16 + n) * G::PAS + ((zl * TY + yl) * 16 + x8 * 8) * 2);
                            accu[m][i][0] = WMfma<H>::run(afm, c0, accu[m][i][0]);
                            accu[m][i][1] = WMfma<H>::run(afm, b1, accu[m][i][1]);
                            accu[m][i][2] = WMfma<H>::run(afm, b2, accu[m][i][2]);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int ti = 0; ti < 7; ++ti) {
                const int tap = wave + 4 * ti;
                if (tap < 27) {
                    const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
                    const unsigned char* qrow = q_lds + tx * G::QTS + (((S * zl + tz) * G::QY + (S * yl + ty)) * 16 + x8 * 8) * 2;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const uint4 bf = *reinterpret_cast<const uint4*>(qrow + (nb * 16 + n) * G::QBS);
                        acc[ti][nb] = WMfma<H>::run(af, bf, acc[ti][nb]);
                    }
                }
            }
            }
        }
        __syncthreads();
    }

    // ---- partial sums of this workgroup: part[blockIdx.x][tap][a][b] (lane (n, g) holds a = 4g..4g+3, b = n) ----
    float* part = a.part + (long)blockIdx.x * 27 * a.ca16 * a.cb16;
    if constexpr (G::ONE) {
#pragma unroll
        for (int i = 0; i < NUW; ++i) {
            const int u = wave + 4 * i;
            if (u < NU) {
                const int pair = u / NB, nb = u % NB;
                const int tz = G::P2D ? 1 : pair / 3, ty = pair % 3;
#pragma unroll
                for (int m = 0; m < NA; ++m)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (a0 + m * 16 + g * 4 + k < a.ca && b0 + nb * 16 + n < a.cb)      // (the finishing kernel reads the valid [ca][cb] block only:
                                part[((long)((tz * 3 + ty) * 3 + tx) * a.ca16 + a0 + m * 16 + g * 4 + k) * a.cb16 + b0 + nb * 16 + n] = accu[m][i][tx][k];   // 8 x 8 layers wrote 4x that)
            }
        }
    } else {
#pragma unroll
        for (int ti = 0; ti < 7; ++ti) {
            const int tap = wave + 4 * ti;
            if (tap < 27) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (a0 + g * 4 + i < a.ca && b0 + nb * 16 + n < a.cb)
                            part[((long)tap * a.ca16 + a0 + g * 4 + i) * a.cb16 + b0 + nb * 16 + n] = acc[ti][nb][i];
            }
        }
    }
}

// dw[a][b][t] = sum_blk part[blk][t][a][b], fixed order.  Block = 16 outputs x 16 workgroup slices (each thread walks
// nblk / 16 partials, then the slices are combined through LDS in slice order): the per-output chain of dependent L2 reads
// is 16x shorter than one thread per output.
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int nblk, int ca, int cb, int ca16, int cb16,
                                                           float* __restrict__ dw, int accumulate, int plane2d) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    const int n = 27 * ca * cb;
    float s0 = 0.f, s1 = 0.f;
    // outputs are walked in the WORKSPACE's order (t, a, b with b fastest): the 16 lanes of a slice read consecutive floats
    // (tap-fastest order put every lane on its own cache line: 35 us per layer, 4.3 ms of a Vis-MVSNet training step)
    const int b = i % cb, r_ = i / cb;
    const int aa = r_ % ca, t = r_ / ca;
    const int o = (aa * cb + b) * 27 + t;         // dw[a][b][t]
    if (i < n && !(plane2d && t / 9 != 1)) {        // (single-plane volumes: only kernel slice tz = 1 was computed, the others are zero)
        const long off = ((long)t * ca16 + aa) * cb16 + b;
        const long stride = 27L * ca16 * cb16;
        int k = g;
        for (; k + 16 < nblk; k += 32) { s0 += part[off + k * stride]; s1 += part[off + (k + 16) * stride]; }
        if (k < nblk) s0 += part[off + k * stride];
    }
    red[g][c] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][c];
        dw[o] = accumulate ? dw[o] + v : v;
    }
}

static inline int wg_ceil(int a, int b) { return (a + b - 1) / b; }

struct WgPlan { int tz, ty, nb, na, ntz, nty, ntx, ntiles, nblk, ny, ca16, cb16; };

static WgPlan wgrad_plan(int B, int Dp, int Hp, int Wp, int ca, int cb, int stride) {
    WgPlan p;
    p.tz = 2; p.ty = stride == 1 ? 4 : 2;
    if (Dp == 1 && stride == 1) { p.tz = 1; p.ty = 8; }          // 2-D layers (P2D)
    p.ca16 = wg_ceil(ca, 16) * 16; p.cb16 = wg_ceil(cb, 16) * 16;
    p.nb = (p.cb16 % 32 == 0) ? 2 : 1;
    p.na = (p.tz == 1 && p.ca16 % 64 == 0) ? 4 : 1;          // single-plane mode: four a-tiles share one staging of Q
    p.ntz = wg_ceil(Dp, p.tz); p.nty = wg_ceil(Hp, p.ty); p.ntx = wg_ceil(Wp, 16);
    p.ntiles = B * p.ntz * p.nty * p.ntx;
    p.ny = (p.ca16 / (16 * p.na)) * (p.cb16 / (16 * p.nb));
    // ONE resident round of persistent workgroups: as many as the LDS tile lets a CU hold (at most 4), times 256 CUs.  (A fixed 1024
    // left the 41 KB tiles of the two-b-tile stride-1 kernel -- three per CU -- with a third of a second round: conv0's gradient
    // 380 -> 314 us at 768.)
    const int lds = stride == 1 ? (p.tz == 1 ? (p.na == 4 ? (p.nb == 2 ? WgGeom<1, 1, 8, 2, 4>::LDS : WgGeom<1, 1, 8, 1, 4>::LDS)
                                                            : (p.nb == 2 ? WgGeom<1, 1, 8, 2>::LDS : WgGeom<1, 1, 8, 1>::LDS))
                                             : (p.nb == 2 ? WgGeom<1, 2, 4, 2>::LDS : WgGeom<1, 2, 4, 1>::LDS))
                                : (p.nb == 2 ? WgGeom<2, 2, 2, 2>::LDS : WgGeom<2, 2, 2, 1>::LDS);
    int per_cu = (160 * 1024) / (lds + 512);
    per_cu = per_cu > 4 ? 4 : per_cu < 1 ? 1 : per_cu;
    int want = 256 * per_cu / p.ny;
#ifdef PSCV_ABLATE
    if (const char* e = getenv("PSCV_WG_WANT")) want = atoi(e) / p.ny;
#endif
    if (want < 1) want = 1;
    p.nblk = p.ntiles < want ? p.ntiles : want;
    return p;
}

template <typename H, int S, int TZ, int TY, int NB, int NA = 1>
static int wgrad_launch(const WgradArgs& a, const WgPlan& p, hipStream_t st) {
    using G = WgGeom<S, TZ, TY, NB, NA>;
    static_assert(G::LDS <= 160 * 1024, "wgrad tile does not fit the LDS");
    auto kern = wgrad_kernel<H, S, TZ, TY, NB, NA>;
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), G::LDS);
        if (e != hipSuccess) { set_error("pscv_conv3d_wgrad: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL(kern, dim3(p.nblk, p.ny), dim3(256), G::LDS, st, a);
    return 0;
}

}  // namespace pscv

using namespace pscv;

extern "C" long pscv_conv3d_wgrad_workspace(int B, int Dp, int Hp, int Wp, int ca, int cb, int stride) {
    PSCV_CHECK_ARG(B > 0 && Dp > 0 && Hp > 0 && Wp > 0 && ca > 0 && cb > 0 && (stride == 1 || stride == 2),
                   "pscv_conv3d_wgrad_workspace: bad arguments");
    const WgPlan p = wgrad_plan(B, Dp, Hp, Wp, ca, cb, stride);
    return (long)p.nblk * 27 * p.ca16 * p.cb16;
}

extern "C" int pscv_conv3d_wgrad(const void* p, int p_cstride, int p_coff, int ca, const void* q, int q_cstride, int q_coff,
                                 int cb, int dtype, int B, int Dp, int Hp, int Wp, int stride, float* workspace, float* dw,
                                 int accumulate, void* stream) {
    PSCV_CHECK_ARG(p && q && workspace && dw, "pscv_conv3d_wgrad: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && Dp > 0 && Hp > 0 && Wp > 0, "pscv_conv3d_wgrad: bad sizes");
    PSCV_CHECK_ARG(stride == 1 || stride == 2, "pscv_conv3d_wgrad: stride %d must be 1 or 2", stride);
    PSCV_CHECK_ARG(ca % 8 == 0 && cb % 8 == 0 && ca >= 8 && cb >= 8 && ca <= 64 && cb <= 64,
                   "pscv_conv3d_wgrad: channel counts %d x %d must be multiples of 8 in [8,64]", ca, cb);
    PSCV_CHECK_ARG(p_cstride % 8 == 0 && p_coff % 8 == 0 && p_coff + ca <= p_cstride && q_cstride % 8 == 0 && q_coff % 8 == 0 &&
                       q_coff + cb <= q_cstride, "pscv_conv3d_wgrad: channel slices must be 8-aligned and inside their strides");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_conv3d_wgrad: dtype %d must be bf16 or fp16", dtype);
    const WgPlan pl = wgrad_plan(B, Dp, Hp, Wp, ca, cb, stride);
    WgradArgs a;
    a.p = reinterpret_cast<const uint16_t*>(p); a.q = reinterpret_cast<const uint16_t*>(q); a.part = workspace;
    a.p_cs = p_cstride; a.p_co = p_coff; a.q_cs = q_cstride; a.q_co = q_coff;
    a.ca = ca; a.cb = cb; a.ca16 = pl.ca16; a.cb16 = pl.cb16;
    a.B = B; a.Dp = Dp; a.Hp = Hp; a.Wp = Wp;
    a.Dq = stride * Dp; a.Hq = stride * Hp; a.Wq = stride * Wp;
    a.ntz = pl.ntz; a.nty = pl.nty; a.ntx = pl.ntx; a.ntiles = pl.ntiles;
    a.abl = 0;
#ifdef PSCV_ABLATE
    a.abl = pscv::g_fuse_c0;
#endif
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
#define PSCV_WG(HT)                                                                       \
    if (pl.tz == 1 && pl.na == 4) rc = pl.nb == 2 ? wgrad_launch<HT, 1, 1, 8, 2, 4>(a, pl, st) : wgrad_launch<HT, 1, 1, 8, 1, 4>(a, pl, st); \
    else if (pl.tz == 1) rc = pl.nb == 2 ? wgrad_launch<HT, 1, 1, 8, 2>(a, pl, st) : wgrad_launch<HT, 1, 1, 8, 1>(a, pl, st); \
    else if (stride == 1) rc = pl.nb == 2 ? wgrad_launch<HT, 1, 2, 4, 2>(a, pl, st) : wgrad_launch<HT, 1, 2, 4, 1>(a, pl, st); \
    else rc = pl.nb == 2 ? wgrad_launch<HT, 2, 2, 2, 2>(a, pl, st) : wgrad_launch<HT, 2, 2, 2, 1>(a, pl, st);
    if (dtype == PSCV_BF16) { PSCV_WG(bf16_t) } else { PSCV_WG(f16_t) }
#undef PSCV_WG
    if (rc) return rc;
    PSCV_CHECK_LAUNCH("pscv_conv3d_wgrad");
    const int n = 27 * ca * cb;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((n + 15) / 16), dim3(256), 0, st, workspace, pl.nblk, ca, cb, pl.ca16, pl.cb16, dw, accumulate, pl.tz == 1 ? 1 : 0);
    PSCV_CHECK_LAUNCH("pscv_conv3d_wgrad(finish)");
    return 0;
}
