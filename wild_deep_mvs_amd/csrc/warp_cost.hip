// Fused plane-sweep warp + cost aggregation for gfx950 (MI355X).
//
// One pass: for every cost-volume voxel (b, d, y, x) the source feature maps are sampled at the
// homography-warped position and folded straight into the cost statistic in fp32 registers; the
// per-view warped volumes of the reference (503 MB fp32 each at 5-view 512x640 D=192) never exist.
// Algorithmic HBM traffic = read V feature maps + write the cost volume once.
//
// Layout: feature maps are channels-last [B,h,w,C] so a bilinear tap is one contiguous C-vector;
// the cost volume is [B,D,h,w,C], the layout the MFMA conv3d kernel consumes.
// Mapping: LPV lanes share a voxel (each owns CPL = C/LPV channels, loaded 16 B at a time); a wave
// therefore covers 64/LPV x-adjacent voxels and writes one contiguous 64/LPV * C * esize byte run.
// A block owns 256/LPV pixels x PPD depth planes; blocks are remapped so that each XCD works on a
// contiguous band of reference pixels (its source footprint stays inside that XCD's 4 MiB L2).
//
// Reference semantics restated here (file:line of fdarmon/wild_deep_mvs):
//   PROJ  geometry  models/MVSNet/module.py:127-166, models/CVP_MVSNet/models/modules.py:74-128,241-281
//   HOMOG geometry  models/VisMVSNet/homography.py:23-120
//   variance        models/MVSNet/model.py:113-139   (CVP rounding order: models/CVP_MVSNet/models/net.py:148)
//   softmin         models/MVSNet/model.py:141-173
//   group corr.     models/VisMVSNet/nn_utils.py:473-490
#include <string.h>

#include "warp_common.h"

namespace pscv {

template <typename TIn, typename TOut, int C, int LPV, int GEOM, int COST>
__global__ __launch_bounds__(256, 4) void warp_cost_kernel(const WarpArgs a) {
    constexpr int CPL = C / LPV;        // channels per lane
    constexpr int PPB = 256 / LPV;      // pixels per block
    constexpr int PIXB = C * (int)sizeof(TIn);   // bytes of one source texel
    constexpr int RQ = (GEOM == PSCV_GEOM_HOMOG) ? 2 : 1;   // float4s of depth-independent ray terms per (view, pixel)
    static_assert(CPL % 8 == 0, "a lane owns whole 8-channel groups");

    // XCD-aware bijective remap: hardware places block `bid` on XCD bid % 8; give XCD k a contiguous
    // run of work ids (pixel-block major, depth-chunk minor) = a band of reference pixels.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int pb = __builtin_amdgcn_readfirstlane(wg / a.n_dchunks);
    const int dc = __builtin_amdgcn_readfirstlane(wg - pb * a.n_dchunks);

    // a block never straddles batch items, so b (and with it every camera / depth-plane address) is
    // wave-uniform and those reads become scalar loads
    // (integer division runs on the vector ALU: readfirstlane tells the compiler the results are scalars again)
    const int b = __builtin_amdgcn_readfirstlane(pb / a.npb_batch);
    const int pbb = __builtin_amdgcn_readfirstlane(pb - b * a.npb_batch);

    const int tid = threadIdx.x;
    // camera blocks of this batch item -> LDS once per workgroup.  (Read straight from global they were fetched with
    // VECTOR loads -- the compiler cannot use the scalar cache for memory the kernel might also store to -- which put a
    // second dependent memory latency and 3 extra VMEM instructions into every (plane, view) iteration.)
    __shared__ float cam_lds[PSCV_MAX_SRC * PSCV_CAM_FLOATS];
    extern __shared__ __attribute__((aligned(16))) float4 ray_lds[];   // [n_src][PPB][RQ]
    for (int i = tid; i < a.n_src * PSCV_CAM_FLOATS; i += 256) {
        const int v = i / PSCV_CAM_FLOATS, k = i - v * PSCV_CAM_FLOATS;
        cam_lds[i] = a.cams[((long)v * a.B + b) * PSCV_CAM_FLOATS + k];
    }
    __syncthreads();
    const int hw = a.h * a.w;
    const int pl = tid / LPV;           // pixel slot inside the block
    int pflat = pbb * PPB + pl;
    const bool active = pflat < hw;
    pflat = active ? pflat : hw - 1;
    const int choff = (tid % LPV) * CPL;
    const unsigned chb = (unsigned)choff * (unsigned)sizeof(TIn);
    const long pix = (long)b * hw + pflat;
    const int y = pflat / a.w;
    const int x = pflat - y * a.w;
    const float off = (GEOM == PSCV_GEOM_HOMOG) ? 0.5f : 0.0f;   // homography.py:78-79 half-pixel centres
    const float px = (float)x + off, py = (float)(y + a.ref_y0) + off;

    // Depth-independent part of the warp, once per (pixel, view) instead of once per (plane, view):
    //   PROJ   q = rot (x, y, 1) * d + trans         -> rot (x, y, 1)                        module.py:138-144
    //   HOMOG  hom = A p - (Bm p) / (d + 1e-9)       -> A p, Bm p                            homography.py:63-69
    // Every lane of a pixel computes the same values; a pixel's lanes sit in one wave and only that wave reads the
    // slot back, so program order is all the synchronisation this needs.
    for (int v = 0; v < a.n_src; ++v) {
        const float* cam = cam_lds + v * PSCV_CAM_FLOATS;
        float4* slot_p = ray_lds + (v * PPB + pl) * RQ;
        const float ax = fmaf(cam[1], py, cam[0] * px) + cam[2];
        const float ay = fmaf(cam[4], py, cam[3] * px) + cam[5];
        const float az = fmaf(cam[7], py, cam[6] * px) + cam[8];
        if (GEOM == PSCV_GEOM_PROJ) {
            slot_p[0] = make_float4(ax, ay, az, 0.0f);
        } else {
            const float bx = fmaf(cam[10], py, cam[9] * px) + cam[11];
            const float by = fmaf(cam[13], py, cam[12] * px) + cam[14];
            const float bz = fmaf(cam[16], py, cam[15] * px) + cam[17];
            slot_p[0] = make_float4(ax, ay, az, bx);
            slot_p[1] = make_float4(by, bz, 0.0f, 0.0f);
        }
    }

    const TIn* ref = reinterpret_cast<const TIn*>(a.ref);
    VecF<CPL> rf;
    if (COST == PSCV_COST_VARIANCE_PARTIAL && !a.ref) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) rf.v[j] = 0.0f;
    } else if (COST != PSCV_COST_WARP_ONLY) {
        rf = load_chan<TIn, CPL>(ref + pix * C + choff);
    }

    const int d0 = dc * a.ppd;
    const int d1 = min(a.D, d0 + a.ppd);
    const float invN = 1.0f / (float)(a.n_src + 1);
    const float invN2 = 1.0f / ((float)(a.n_src + 1) * (float)(a.n_src + 1));
    // wave-uniform byte offset of batch item b inside a source map, kept in scalar registers so that the taps use the
    // scalar-base + 32-bit vector-offset addressing form
    const unsigned long img_bytes_v = (unsigned long)b * a.hs * a.ws * PIXB;
    const unsigned long img_bytes = ((unsigned long)__builtin_amdgcn_readfirstlane((unsigned)(img_bytes_v >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)img_bytes_v);
    TOut* out = reinterpret_cast<TOut*>(a.out);

    for (int d = d0; d < d1; ++d) {
        const float dval = a.depth_per_pixel ? a.depth[(long)b * a.depth_bstride + (long)d * hw + pflat]
                                             : a.depth[(long)b * a.depth_bstride + d];
        const float inv_d = (GEOM == PSCV_GEOM_HOMOG) ? __builtin_amdgcn_rcpf(dval + 1e-9f) : 0.0f;
        const long vox = ((long)b * a.D + d) * hw + pflat;

        VecF<CPL> acc0, acc1;   // variance: sum, sum of squares; softmin: sum e*diff
        float sum_e = 0.0f;
        if (COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP || COST == PSCV_COST_VARIANCE_PARTIAL) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc0.v[j] = rf.v[j]; acc1.v[j] = rf.v[j] * rf.v[j]; }   // (rf = 0 without a reference)
        } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc0.v[j] = 0.0f; acc1.v[j] = 0.0f; }
        }

        for (int v = 0; v < a.n_src; ++v) {
            const float* cam = cam_lds + v * PSCV_CAM_FLOATS;
            const float4* ray = ray_lds + (v * PPB + pl) * RQ;
            float hx, hy, hz;
            if (GEOM == PSCV_GEOM_PROJ) {
                const float4 r0 = ray[0];
                hx = fmaf(r0.x, dval, cam[9]);
                hy = fmaf(r0.y, dval, cam[10]);
                hz = fmaf(r0.z, dval, cam[11]);
            } else {
                const float4 r0 = ray[0], r1 = ray[1];
                hx = fmaf(-r0.w, inv_d, r0.x);
                hy = fmaf(-r1.x, inv_d, r0.y);
                hz = fmaf(-r1.y, inv_d, r0.z);
            }
            // perspective divide; points at or behind the source camera go to (-10, -10)   module.py:146-150,
            // homography.py:113-117 (which also clamps the divisor at 1e-9)
            const bool front = hz > 0.0f;
            const float inv_z = __builtin_amdgcn_rcpf(GEOM == PSCV_GEOM_HOMOG ? fmaxf(hz, 1e-9f) : hz);
            float u = front ? hx * inv_z : -10.0f;
            float w_ = front ? hy * inv_z : -10.0f;
            if (GEOM == PSCV_GEOM_HOMOG) { u *= a.sx; w_ *= a.sy; }   // PROJ: index scale is exactly 1
            // normalise -> clamp -> align_corners=True un-normalise collapses to a (scaled,) clamped index
            const float ix = __builtin_amdgcn_fmed3f(u, a.xlo, a.xhi);
            const float iy = __builtin_amdgcn_fmed3f(w_, a.ylo, a.yhi);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float fx = ix - x0f, fy = iy - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;

            const char* img = reinterpret_cast<const char*>(a.src[v]) + img_bytes;
            // all four taps of every voxel of this wave inside the image?  (the common case away from the border)
            const bool interior = (unsigned)x0 < (unsigned)(a.ws - 1) && (unsigned)y0 < (unsigned)(a.hs - 1);
            Taps taps;
            VecF<CPL> wv;
            if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
                make_taps<true, PIXB>(fx, fy, x0, y0, a.hs, a.ws, chb, taps);
                wv = blend_taps<TIn, CPL, true, PIXB>(img, taps);
            } else {
                make_taps<false, PIXB>(fx, fy, x0, y0, a.hs, a.ws, chb, taps);
                wv = blend_taps<TIn, CPL, false, PIXB>(img, taps);
            }
            if (COST == PSCV_COST_VARIANCE || COST == PSCV_COST_VARIANCE_CVP || COST == PSCV_COST_VARIANCE_PARTIAL) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    acc0.v[j] += wv.v[j];
                    acc1.v[j] = fmaf(wv.v[j], wv.v[j], acc1.v[j]);
                }
            } else if (COST == PSCV_COST_SOFTMIN) {
                VecF<CPL> diff;
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const float t = rf.v[j] - wv.v[j];
                    diff.v[j] = t * t;
                    part += diff.v[j];
                }
                // sum over all C channels = over the LPV lanes that share this voxel
#pragma unroll
                for (int m = 1; m < LPV; m <<= 1) part += __shfl_xor(part, m, 64);
                const float e = __expf(-a.temp * part);
                sum_e += e;
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc0.v[j] = fmaf(e, diff.v[j], acc0.v[j]);
            } else if (COST == PSCV_COST_GROUPCORR) {
                constexpr int G = C / 4;
                TOut* o = out + (long)v * a.out_view_stride + vox * G + choff / 4;
                if (active) {
#pragma unroll
                    for (int g = 0; g < CPL / 4; g += 2) {
                        const float c0 = rf.v[4 * g] * wv.v[4 * g] + rf.v[4 * g + 1] * wv.v[4 * g + 1] +
                                         rf.v[4 * g + 2] * wv.v[4 * g + 2] + rf.v[4 * g + 3] * wv.v[4 * g + 3];
                        const float c1 = rf.v[4 * g + 4] * wv.v[4 * g + 4] + rf.v[4 * g + 5] * wv.v[4 * g + 5] +
                                         rf.v[4 * g + 6] * wv.v[4 * g + 6] + rf.v[4 * g + 7] * wv.v[4 * g + 7];
                        Elem<TOut>::store2(o + g, c0, c1);
                    }
                }
            } else {  // WARP_ONLY
                if (active) store_chan<TOut, CPL>(out + (long)v * a.out_view_stride + vox * C + choff, wv);
            }
        }

        if (COST == PSCV_COST_VARIANCE) {
            VecF<CPL> o;
#pragma unroll
            for (int j = 0; j < CPL; ++j) o.v[j] = acc1.v[j] * invN - (acc0.v[j] * acc0.v[j]) * invN2;
            if (active) store_chan<TOut, CPL>(out + vox * C + choff, o);
        } else if (COST == PSCV_COST_VARIANCE_CVP) {
            VecF<CPL> o;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float m = acc0.v[j] * invN;
                o.v[j] = acc1.v[j] * invN - m * m;
            }
            if (active) store_chan<TOut, CPL>(out + vox * C + choff, o);
        } else if (COST == PSCV_COST_VARIANCE_PARTIAL) {
            if (active) {
                store_chan<TOut, CPL>(out + vox * C + choff, acc0);
                store_chan<TOut, CPL>(out + a.out_view_stride + vox * C + choff, acc1);
            }
        } else if (COST == PSCV_COST_SOFTMIN) {
            VecF<CPL> o;
            const float inv = 1.0f / (sum_e + 1e-6f);
#pragma unroll
            for (int j = 0; j < CPL; ++j) o.v[j] = acc0.v[j] * inv;
            if (active) store_chan<TOut, CPL>(out + vox * C + choff, o);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
static Knob g_warp_lpv_override = {0, KNOB_WARP_LPV};  // 0 = default heuristic; set through pscv_set_tuning("warp_lpv", n)
static Knob g_warp_ppd_override = {0, KNOB_WARP_PPD};
static Knob g_warp_gc_lds = {1, KNOB_WARP_GC_LDS};   // 1 (default): group-correlation volumes over per-batch planes on the LDS-staged kernel (warp_gc_lv.hip); 2: per-pixel planes too; 0: quad kernel
static Knob g_warp_tiled = {1, KNOB_WARP_TILED};     // 1 (default; 2 = the same): the LDS-staged kernel (warp_cost_tiled.hip) where it applies: fp32
                                 // patches, scalar fp32 blend, same bits as the direct kernels; 0: direct kernels; 4: lane-owns-voxel kernel
extern Knob g_conv_small_tiles;   // conv3d.hip
extern Knob g_sweep_th16;         // conv3d_sweep.hip
extern Knob g_sweep_dc;
extern Knob g_sweep_kdm;
extern Knob g_sweep_kdm_pd;
extern Knob g_sweepc_slots;
extern Knob g_sweepc_pd;
}
extern pscv::Knob g_c1_nb;
extern pscv::Knob g_c1_sweep;
namespace pscv {
extern Knob g_warp_bwd_direct;    // warp_bwd.hip
extern Knob g_conv_s2_sweep;      // conv3d_sweep_s2.hip
extern Knob g_s2s_slots;
static Knob g_warp_q2 = {1, KNOB_WARP_Q2};        // 1: 32-channel 16-bit sweeps use the quad-mapped kernel (warp_cost_q2.hip)
int warp_cost_q2_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st);
int warp_cost_tiled_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st);
// the lane-owns-voxel kernel ("warp_tiled" = 4; warp_cost_lv.hip): variance costs
int warp_cost_lv_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st);
// group-wise correlation, LDS-staged (warp_gc_lv.hip)
int warp_gc_lv_try(WarpArgs& a, int C, int geom, int cost, int in_dtype, int out_dtype, int ppd_override, hipStream_t st);

// Instantiated (geometry, cost) pairs: the variance / softmin statistics belong to the PROJ models (MVSNet,
// CVP), group-wise correlation to the HOMOG model (Vis); the plain warp exists for both.
// dynamic LDS = the per-(view, pixel) ray terms; above the 64 KiB default (many views) the kernel needs the opt-in
template <typename K>
static int launch_one(K kern, const WarpArgs& a, int nblk, size_t ray_bytes, hipStream_t st) {
    if (ray_bytes > 60000) {   // rare (> 14 HOMOG views): not worth caching per kernel
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)ray_bytes);
        if (e != hipSuccess) { set_error("pscv_warp_cost: hipFuncSetAttribute(%zu B LDS): %s", ray_bytes, hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ray_bytes, st, a);
    return 0;
}

template <typename TIn, typename TOut, int C, int LPV, int GEOM>
static int launch_cost(const WarpArgs& a, int cost, int nblk, hipStream_t st) {
    const size_t ray_bytes = (size_t)a.n_src * (256 / LPV) * (GEOM == PSCV_GEOM_HOMOG ? 32 : 16);
#define PSCV_LAUNCH_COST(COSTV)                                                                              \
    case COSTV:                                                                                              \
        return launch_one(warp_cost_kernel<TIn, TOut, C, LPV, GEOM, COSTV>, a, nblk, ray_bytes, st);
    if constexpr (GEOM == PSCV_GEOM_PROJ) {
        switch (cost) {
            PSCV_LAUNCH_COST(PSCV_COST_VARIANCE)
            PSCV_LAUNCH_COST(PSCV_COST_VARIANCE_CVP)
            PSCV_LAUNCH_COST(PSCV_COST_SOFTMIN)
            PSCV_LAUNCH_COST(PSCV_COST_WARP_ONLY)
            case PSCV_COST_VARIANCE_PARTIAL:
                if constexpr (sizeof(TOut) == 4)
                    return launch_one(warp_cost_kernel<TIn, TOut, C, LPV, GEOM, PSCV_COST_VARIANCE_PARTIAL>, a, nblk, ray_bytes, st);
                break;
        }
    } else {
        switch (cost) {
            PSCV_LAUNCH_COST(PSCV_COST_GROUPCORR)
            PSCV_LAUNCH_COST(PSCV_COST_WARP_ONLY)
        }
    }
#undef PSCV_LAUNCH_COST
    set_error("pscv_warp_cost: cost mode %d is not available with geometry %d", cost, (int)GEOM);
    return -1;
}

template <typename TIn, typename TOut, int C, int LPV>
static int launch_geom(WarpArgs& a, int geom, int cost, hipStream_t st) {
    constexpr int PPB = 256 / LPV;
    a.npb_batch = (a.h * a.w + PPB - 1) / PPB;
    const long n_pixblocks = (long)a.npb_batch * a.B;
    int ppd = g_warp_ppd_override > 0 ? g_warp_ppd_override : 8;
    // keep >= ~4096 blocks in flight for 256 CUs when the problem allows it
    while (ppd > 1 && n_pixblocks * ((a.D + ppd - 1) / ppd) < 4096) ppd >>= 1;
    a.ppd = ppd;
    a.n_dchunks = (a.D + ppd - 1) / ppd;
    const long nblk = n_pixblocks * a.n_dchunks;
    if (nblk <= 0 || nblk > 0x7fffffffL) {
        set_error("pscv_warp_cost: bad grid size %ld", nblk);
        return -1;
    }
    if (geom == PSCV_GEOM_PROJ) return launch_cost<TIn, TOut, C, LPV, PSCV_GEOM_PROJ>(a, cost, (int)nblk, st);
    if (geom == PSCV_GEOM_HOMOG) return launch_cost<TIn, TOut, C, LPV, PSCV_GEOM_HOMOG>(a, cost, (int)nblk, st);
    set_error("pscv_warp_cost: unknown geometry %d", geom);
    return -1;
}

template <typename TIn, typename TOut>
static int launch_channels(WarpArgs& a, int C, int geom, int cost, hipStream_t st) {
    // default: 16 channels per lane (two 16-byte loads per tap): measured fastest on MI355X at C = 32
    // (226 us vs 284 us for 8 and 286 us for 32 channels per lane, 5-view 128x160 D=192) -- the sample
    // coordinates are computed once per 2 lanes instead of once per 4
    int lpv = g_warp_lpv_override > 0 ? g_warp_lpv_override : (C == 32 ? 2 : C / 8);
    if (cost == PSCV_COST_GROUPCORR && C / lpv < 8) lpv = C / 8;
#define PSCV_CASE(CC, LL) \
    if (C == CC && lpv == LL) return launch_geom<TIn, TOut, CC, LL>(a, geom, cost, st);
    PSCV_CASE(32, 4) PSCV_CASE(32, 2) PSCV_CASE(32, 1)
    PSCV_CASE(16, 2)
#undef PSCV_CASE
    set_error("pscv_warp_cost: unsupported channels/lanes-per-voxel C=%d lpv=%d", C, lpv);
    return -1;
}

}  // namespace pscv

namespace pscv {
Knob g_warp_tile = {0, KNOB_WARP_TILE};       // warp_cost_tiled.hip variants (measurement)
Knob g_fuse_c0 = {0, KNOB_FUSE_C0};           // reserved: fused warp -> conv0 experiment
Knob g_warp_lds_pad = {0, KNOB_WARP_LDS_PAD};       // KiB of LDS the LDS-staged warp kernel requests on top of its need (fewer workgroups per CU)
extern Knob g_conv2d_wlds;                    // conv2d.hip
extern Knob g_conv_tall64;                    // conv3d.hip
extern Knob g_conv_small_nt;                  // conv3d.hip
extern Knob g_tail_nbk;                       // conv3d_tail.hip
extern Knob g_conv_wide;                      // conv3d_wide.hip
}
extern pscv::Knob g_block8_slots;             // conv3d_block8.hip
extern pscv::Knob g_softargmin_small;         // softargmin.hip
namespace pscv {
static Knob* find_knob(const char* key) {
    static const struct { const char* name; Knob* k; } table[] = {
        {"warp_lpv", &g_warp_lpv_override}, {"warp_ppd", &g_warp_ppd_override}, {"conv_small_tiles", &g_conv_small_tiles},
        {"warp_tiled", &g_warp_tiled}, {"warp_gc_lds", &g_warp_gc_lds}, {"warp_q2", &g_warp_q2}, {"c1_nb", &g_c1_nb}, {"c1_sweep", &g_c1_sweep},
        {"sweep_th16", &g_sweep_th16}, {"sweep_dc", &g_sweep_dc}, {"sweep_kdm", &g_sweep_kdm}, {"sweep_kdm_pd", &g_sweep_kdm_pd}, {"sweepc_slots", &g_sweepc_slots}, {"sweepc_pd", &g_sweepc_pd},
        {"warp_bwd_direct", &g_warp_bwd_direct}, {"conv_s2_sweep", &g_conv_s2_sweep}, {"s2s_slots", &g_s2s_slots},
        {"warp_tile", &g_warp_tile}, {"fuse_c0", &g_fuse_c0}, {"warp_lds_pad", &g_warp_lds_pad}, {"conv2d_wlds", &g_conv2d_wlds}, {"conv_tall64", &g_conv_tall64}, {"block8_slots", &::g_block8_slots}, {"softargmin_small", &::g_softargmin_small}, {"tail_nbk", &g_tail_nbk}, {"conv_wide", &g_conv_wide}, {"conv_small_nt", &g_conv_small_nt}};
    for (const auto& e : table)
        if (!strcmp(key, e.name)) return e.k;
    return nullptr;
}
static int knob_value(const char* key, int value) { return (!strcmp(key, "warp_tiled") && value < 0) ? 1 : value; }   // -1: default
}  // namespace pscv

extern "C" int pscv_set_tuning(const char* key, int value) {
    using namespace pscv;
    PSCV_CHECK_ARG(key, "pscv_set_tuning: null key");
    Knob* k = find_knob(key);
    if (!k) { set_error("pscv_set_tuning: unknown key '%s'", key); return -1; }
    // ("warp_tiled" = 3 selected the SLP-packed diagnostic build of the LDS-staged kernel, removed in round 5: refuse it rather than
    //  silently measure the default kernel under its name)
    if (!strcmp(key, "warp_tiled") && value == 3) { set_error("pscv_set_tuning: warp_tiled = 3 (packed diagnostic build) no longer exists"); return -1; }
    k->set(knob_value(key, value));
    return 0;
}

extern "C" int pscv_set_tuning_thread(const char* key, int value, int enable) {
    using namespace pscv;
    PSCV_CHECK_ARG(key, "pscv_set_tuning_thread: null key");
    Knob* k = find_knob(key);
    if (!k) { set_error("pscv_set_tuning_thread: unknown key '%s'", key); return -1; }
    knob_thread_set(k->id, knob_value(key, value), enable != 0);
    return 0;
}

extern "C" int pscv_get_tuning(const char* key, int* value) {
    using namespace pscv;
    PSCV_CHECK_ARG(key && value, "pscv_get_tuning: null argument");
    Knob* k = find_knob(key);
    if (!k) { set_error("pscv_get_tuning: unknown key '%s'", key); return -1; }
    *value = (int)*k;
    return 0;
}

extern "C" int pscv_warp_cost(const void* ref, const void* const* srcs, int n_src, const float* cams,
                              const float* depth, long depth_bstride, int depth_per_pixel, int geom, int cost,
                              float temp, void* out, int B, int C, int h, int w, int hs, int ws, int D, int in_dtype,
                              int out_dtype, void* stream) {
    return pscv_warp_cost_rows(ref, srcs, n_src, cams, depth, depth_bstride, depth_per_pixel, geom, cost, temp, out, B, C, h, w, hs, ws, D,
                               in_dtype, out_dtype, 0, stream);
}

extern "C" int pscv_warp_cost_rows(const void* ref, const void* const* srcs, int n_src, const float* cams,
                                   const float* depth, long depth_bstride, int depth_per_pixel, int geom, int cost,
                                   float temp, void* out, int B, int C, int h, int w, int hs, int ws, int D, int in_dtype,
                                   int out_dtype, int ref_y0, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(ref_y0 >= 0 && ref_y0 < (1 << 20), "pscv_warp_cost_rows: ref_y0=%d", ref_y0);
    PSCV_CHECK_ARG(n_src >= 1 && n_src <= PSCV_MAX_SRC, "pscv_warp_cost: n_src=%d outside [1,%d]", n_src, PSCV_MAX_SRC);
    PSCV_CHECK_ARG(srcs && cams && depth && out, "pscv_warp_cost: null pointer argument");
    PSCV_CHECK_ARG(cost == PSCV_COST_WARP_ONLY || cost == PSCV_COST_VARIANCE_PARTIAL || ref, "pscv_warp_cost: ref is required for cost mode %d", cost);
    PSCV_CHECK_ARG(cost != PSCV_COST_VARIANCE_PARTIAL || out_dtype == PSCV_F32, "pscv_warp_cost: partial sums are written in fp32");
    PSCV_CHECK_ARG(B > 0 && h > 0 && w > 0 && hs > 1 && ws > 1 && D > 0, "pscv_warp_cost: bad sizes");
    PSCV_CHECK_ARG(C % 8 == 0, "pscv_warp_cost: C=%d must be a multiple of 8", C);
    // taps are addressed with 24-bit texel indices and 32-bit byte offsets inside one source image
    PSCV_CHECK_ARG((long)hs * ws < (1L << 24) && (long)hs * ws * C * 4 < (1L << 32), "pscv_warp_cost: source map %dx%dx%d too large", hs, ws, C);
    WarpArgs a;
    a.mode_hist = nullptr;
    a.ref_y0 = ref_y0;
    a.ref = ref;
    for (int i = 0; i < PSCV_MAX_SRC; ++i) a.src[i] = i < n_src ? srcs[i] : nullptr;
    for (int i = 0; i < n_src; ++i) PSCV_CHECK_ARG(srcs[i], "pscv_warp_cost: srcs[%d] is null", i);
    a.cams = cams;
    a.depth = depth;
    a.out = out;
    a.depth_bstride = depth_bstride;
    a.n_src = n_src; a.B = B; a.h = h; a.w = w; a.hs = hs; a.ws = ws; a.D = D;
    a.depth_per_pixel = depth_per_pixel;
    a.temp = temp;
    a.variant = 0;
    const long vol = (long)B * D * h * w;
    a.out_view_stride = cost == PSCV_COST_GROUPCORR ? vol * (C / 4) : vol * C;   // (PARTIAL: offset of the sum-of-squares half)
    if (geom == PSCV_GEOM_PROJ) {
        // grid = u/((W-1)/2) - 1 clamped to +-10, index = (grid+1)/2*(W-1)  ->  index = u clamped to
        // [-4.5 (W-1), 5.5 (W-1)]                                            module.py:151-155
        a.sx = 1.0f; a.sy = 1.0f;
        a.xlo = -4.5f * (ws - 1); a.xhi = 5.5f * (ws - 1);
        a.ylo = -4.5f * (hs - 1); a.yhi = 5.5f * (hs - 1);
    } else {
        // grid = u/W*2 - 1 clamped to +-1.1, index = (grid+1)/2*(W-1)       homography.py:92-96
        a.sx = (float)(ws - 1) / (float)ws; a.sy = (float)(hs - 1) / (float)hs;
        a.xlo = -0.05f * (ws - 1); a.xhi = 1.05f * (ws - 1);
        a.ylo = -0.05f * (hs - 1); a.yhi = 1.05f * (hs - 1);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (g_warp_tiled && g_warp_lpv_override == 0 && cost != PSCV_COST_VARIANCE_PARTIAL) {
        rc = 1;
        if (g_warp_tiled == 4) rc = warp_cost_lv_try(a, C, geom, cost, in_dtype, out_dtype, g_warp_ppd_override, st);
        if (rc == 1 && g_warp_gc_lds && (g_warp_gc_lds >= 2 || !depth_per_pixel)) rc = warp_gc_lv_try(a, C, geom, cost, in_dtype, out_dtype, g_warp_ppd_override, st);
        if (rc == 1)
            rc = warp_cost_tiled_try(a, C, geom, cost, in_dtype, out_dtype, g_warp_ppd_override, st);
        if (rc < 0) return rc;
        if (rc == 0) {
            PSCV_CHECK_LAUNCH("pscv_warp_cost(tiled)");
            return 0;
        }
    }
    if (g_warp_q2 && g_warp_lpv_override == 0) {
        rc = warp_cost_q2_try(a, C, geom, cost, in_dtype, out_dtype, g_warp_ppd_override, st);
        if (rc < 0) return rc;
        if (rc == 0) {
            PSCV_CHECK_LAUNCH("pscv_warp_cost(q2)");
            return 0;
        }
    }
    if (in_dtype == PSCV_BF16 && out_dtype == PSCV_BF16) rc = launch_channels<bf16_t, bf16_t>(a, C, geom, cost, st);
    else if (in_dtype == PSCV_F16 && out_dtype == PSCV_F16) rc = launch_channels<f16_t, f16_t>(a, C, geom, cost, st);
    else if (in_dtype == PSCV_BF16 && out_dtype == PSCV_F32) rc = launch_channels<bf16_t, float>(a, C, geom, cost, st);
    else if (in_dtype == PSCV_F16 && out_dtype == PSCV_F32) rc = launch_channels<f16_t, float>(a, C, geom, cost, st);
    else if (in_dtype == PSCV_F32 && out_dtype == PSCV_F32) rc = launch_channels<float, float>(a, C, geom, cost, st);
    else { set_error("pscv_warp_cost: unsupported dtype pair in=%d out=%d (out must be the input's 16-bit format or fp32)", in_dtype, out_dtype); return -1; }
    if (rc) return rc;
    PSCV_CHECK_LAUNCH("pscv_warp_cost");
    return 0;
}
