// Depth-sweep 3x3x3 convolution for the full-resolution 32 -> 8 layer (MVSNet CostRegNet.conv0: 68 % of the
// regulariser's FLOPs and the only reader of the 32-channel cost volume).  gfx950, wave64.
//
// Design (vs the generic brick kernel in conv3d.hip):
//   * A workgroup owns an 8 x 16 pixel tile and sweeps a run of depth planes.  Input planes live in a 6-slot
//     LDS ring; every input plane is fetched once per sweep (2-D halo only), two new planes per iteration,
//     prefetched into registers while the MFMAs of the current iteration run (issue early / write late).
//   * Plane-pair packing: C_out = 8 fills only half of the 16 MFMA rows, so rows 0-7 compute output plane d
//     and rows 8-15 output plane d+1.  An input plane p feeds plane d with kernel slice kd = p-d+1 and plane
//     d+1 with kd = p-d, so 4 input planes x 9 (kh,kw) taps = 36 MFMAs produce TWO output planes
//     (18 per plane instead of 27), and all 64 lanes end with a useful 8-byte store.
//   * All 36 A fragments (144 VGPRs) stay in registers for the whole sweep: no weight traffic in the loop.
//   * LDS voxels are 64 B (32 ch x 16 bit) with the XOR swizzle chunk ^= ((voxel >> 2) & 1) << 1, which makes
//     every ds_read_b128 B-fragment read conflict-free for any tap offset (brute-forced over the gfx950
//     lane-group model; unswizzled or +16 B padded rows are 2-way).
//
// Replaces (fdarmon/wild_deep_mvs): CostRegNet.conv0 = ConvBnReLU3D(32, 8) models/MVSNet/model.py:46,75
// (block definition models/MVSNet/module.py:41-48).
#include "pscv_common.h"
#include <type_traits>

namespace pscv {

typedef __attribute__((ext_vector_type(8))) __bf16 sw_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 sw_f16x8;
typedef __attribute__((ext_vector_type(4))) float sw_f32x4;

template <typename H> struct SwMfma;
template <> struct SwMfma<bf16_t> {
    __device__ static __forceinline__ sw_f32x4 run(const uint4& a, const uint4& b, const sw_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sw_bf16x8, a), __builtin_bit_cast(sw_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SwMfma<f16_t> {
    __device__ static __forceinline__ sw_f32x4 run(const uint4& a, const uint4& b, const sw_f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sw_f16x8, a), __builtin_bit_cast(sw_f16x8, b), c, 0, 0, 0);
    }
};

Knob g_sweep_dc = {0, KNOB_SWEEP_DC};     // pscv_set_tuning("sweep_dc", n): depth planes per workgroup sweep (0 = heuristic)
Knob g_sweepc_pd = {0, KNOB_SWEEPC_PD};    // pscv_set_tuning("sweepc_pd", 1..3): prefetch distance (iterations) of the narrow-input sweep (0 = 1)
Knob g_sweepc_slots = {0, KNOB_SWEEPC_SLOTS}; // pscv_set_tuning("sweepc_slots", n): resident-workgroup target of the narrow-input sweep (0 = 768)
Knob g_sweep_th16 = {0, KNOB_SWEEP_TH16};   // pscv_set_tuning("sweep_th16", 1) selects the 16-row / 512-thread tile variant (measured
                        // 116 us vs 107 us for 8-row tiles at the headline size: one workgroup per CU hides less latency)

struct SweepArgs {
    const uint16_t* in;
    const uint16_t* wpk;     // [4 p_rel][9 taps][64 lanes][8]
    const float* scale;
    const float* bias;
    const float* floor;
    const uint16_t* skip;
    void* out;
    int in_cs, in_co, skip_cs, skip_co, out_cs, out_co;
    int out_f32;
    int B, D, Hh, W;
    int epi;
    const uint16_t* in2;     // narrow sweep, C_in = 16: channels 8..15 come from this tensor (the same tensor at in_co + 8 for a plain
    int in2_cs, in2_co;      // 16-channel input; another tensor for pscv_conv3d_cat2: torch.cat([a, b], channel) never materialised)
    int nth, ntw, ndc, dc;   // tiles along h, w; depth chunks and planes per chunk (even)
    unsigned mg_th, mg_tw, mg_dc;
};

PSCV_PROF_BUFFER(sweep)
constexpr int SW_BW = 18;
constexpr int SW_VB = 64;                  // bytes per voxel (32 ch x 2 B)
constexpr int SW_NSLOT = 6;
// Tile height TH (rows) is a template parameter: TH = 8 -> 256 threads, 70 KiB ring, two workgroups per CU;
// TH = 16 -> 512 threads, 123 KiB ring, one workgroup per CU but 18/16 instead of 10/8 halo rows per tile.
template <int TH> struct SwGeom {
    static constexpr int BH = TH + 2;
    static constexpr int PV = ((BH * SW_BW + 7) / 8) * 8;   // voxels per plane slot (multiple of 8: swizzle is slot-invariant)
    static constexpr int PB = PV * SW_VB;
    static constexpr int LDS = SW_NSLOT * PB;
    static constexpr int CHUNKS = BH * SW_BW * 4;            // 16-byte chunks per plane
    static constexpr int THREADS = 32 * TH;                  // 2 rows (M-tiles) per wave
    static constexpr int NLD = (CHUNKS + THREADS - 1) / THREADS;
};

__device__ __forceinline__ int sw_lds_off(int v, int chunk) {   // v = in-plane voxel index
    return v * SW_VB + ((chunk ^ (((v >> 2) & 1) << 1)) << 4);
}

template <typename H, int SW_TH>
__global__ __launch_bounds__(32 * SW_TH, 2) void conv3d_sweep8_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = SwGeom<SW_TH>;
    constexpr int SW_PB = G::PB, SW_CHUNKS = G::CHUNKS, NTHR = G::THREADS, NLD = G::NLD;
    constexpr int R = 2;   // rows (M-tiles) per wave

    // ---- work decode (XCD-aware: each XCD gets a contiguous run of (tile, depth-chunk) ids) ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * SW_TH, w0 = twi * 16;
    const int dbeg = dci * a.dc, dend = min(a.D, dbeg + a.dc);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    PSCV_PROF_BEGIN   // (profile builds: slots = prologue | fetch issue | MFMA loop | epilogue | stash incl. the wait for the planes | barrier)

    // ---- A fragments: all weights of the layer, resident for the whole sweep ----
    uint4 wf[4][9];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.wpk);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int t = 0; t < 9; ++t) wf[p][t] = wp[(p * 9 + t) * 64 + lane];
    }

    // ---- per-lane B-fragment offsets inside a plane slot: rows row0 .. row0+R+1, kw 0..2 ----
    const int row0 = wave * R;
    int boff[R + 2][3];
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) boff[rr][kw] = sw_lds_off((row0 + rr) * SW_BW + n + kw, g);

    // ---- staging descriptors: 3 chunks per thread per plane ----
    // Planes are fetched with raw buffer loads: the descriptor covers one input plane, a chunk outside the image carries an
    // out-of-range offset and a plane outside the volume an empty descriptor -- the hardware returns zeros (= the conv's
    // padding): NLD load instructions per plane, no predicate, zero fill or 64-bit address arithmetic per chunk.
    unsigned goff[NLD];
    int loff[NLD];
    bool lval[NLD];
    const long plane_stride = (long)a.Hh * a.W * a.in_cs;
    const unsigned plane_bytes = (unsigned)(plane_stride * 2 - a.in_co * 2);
    const uint16_t* inb = a.in + (long)b * a.D * plane_stride + a.in_co;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + NTHR * i;
        const int v = id >> 2, c = id & 3;
        const int bh = v / SW_BW, bw = v - bh * SW_BW;
        const int gh = h0 - 1 + bh, gw = w0 - 1 + bw;
        lval[i] = id < SW_CHUNKS;
        const bool gval = lval[i] && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
        goff[i] = gval ? ((unsigned)(gh * a.W + gw) * (unsigned)a.in_cs + (unsigned)(c * 8)) * 2u : 0x7ffffff0u;
        loff[i] = sw_lds_off(v, c);
    }
    const int plane_hi = min(a.D - 1, dend);   // last input plane this sweep can use
    auto fetch = [&](int plane, uint4 (&reg)[NLD]) {
        const bool pv = plane >= 0 && plane <= plane_hi;                                         // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(inb + (long)(pv ? plane : 0) * plane_stride), (short)0, pv ? (int)plane_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            reg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[i], 0, 0));
    };
    auto stash = [&](int ring, const uint4 (&reg)[NLD]) {
        unsigned char* sp = smem + ring * SW_PB;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lval[i]) *reinterpret_cast<uint4*>(sp + loff[i]) = reg[i];
    };

    // ---- epilogue constants: this lane always owns channels (g&1)*4 .. +3 of plane d + (g>>1) ----
    const int c0 = (g & 1) * 4;
    float sc[4], bi[4], fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = a.scale ? a.scale[c0 + k] : 1.0f;
        bi[k] = a.bias ? a.bias[c0 + k] : 0.0f;
        fl[k] = a.floor ? a.floor[c0 + k] : 0.0f;
    }

    // ---- prologue: planes dbeg-1 .. dbeg+2 into ring slots 0..3 ----
    {
        uint4 ra[NLD], rb[NLD];
        fetch(dbeg - 1, ra); fetch(dbeg, rb);
        stash(0, ra); stash(1, rb);
        fetch(dbeg + 1, ra); fetch(dbeg + 2, rb);
        stash(2, ra); stash(3, rb);
    }
    __syncthreads();
    PSCV_STAMP(0)

    // One plane pair; RING = the slot holding plane d - 1, a compile-time constant: the sweep below is unrolled over the ring's period
    // (6 slots, 2 per pair), so every slot base is an immediate offset of its ds_read / ds_write (as a run-time ring index each of the
    // 48 reads of a pair had its own v_add in front).
    auto pair = [&](auto ringc, const int d) {
        constexpr int RING = decltype(ringc)::value;
        // issue the next two planes (d+3, d+4) early; they land in LDS after this iteration's MFMAs
        uint4 na[NLD], nb[NLD];
        fetch(d + 3, na);
        fetch(d + 4, nb);
        __builtin_amdgcn_sched_barrier(0);      // both planes are requested HERE (the scheduler otherwise sinks the second below the MFMAs)
        PSCV_STAMP(1)

        sw_f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = sw_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            constexpr int SL = (RING + 4 * SW_NSLOT) % SW_NSLOT;
            const unsigned char* sp = smem + ((SL + p) % SW_NSLOT) * SW_PB;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint4 xf = *reinterpret_cast<const uint4*>(sp + boff[r + kh][kw]);
                        acc[r] = SwMfma<H>::run(wf[p][kh * 3 + kw], xf, acc[r]);
                    }
        }

        PSCV_STAMP(2)
        // epilogue: rows g*4.. of D = channels c0.. of plane d + (g >> 1)
        const int od = d + (g >> 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int oh = h0 + row0 + r, ow = w0 + n;
            if (od < dend && oh < a.Hh && ow < a.W) {
                const long vox = (((long)b * a.D + od) * a.Hh + oh) * a.W + ow;
                float y[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    y[k] = fmaf(acc[r][k], sc[k], bi[k]);
                    if (a.epi & PSCV_EPI_RELU_PRE) y[k] = relu_floor(y[k], fl[k]);
                }
                if (a.skip) {
                    const uint2 sv = *reinterpret_cast<const uint2*>(a.skip + vox * a.skip_cs + a.skip_co + c0);
                    y[0] += Half16<H>::lo(sv.x); y[1] += Half16<H>::hi(sv.x);
                    y[2] += Half16<H>::lo(sv.y); y[3] += Half16<H>::hi(sv.y);
                }
                if (a.epi & PSCV_EPI_RELU_POST) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = relu_floor(y[k], 0.0f);
                }
                if (a.out_f32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + vox * a.out_cs + a.out_co + c0) =
                        make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + vox * a.out_cs + a.out_co + c0) =
                        make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                }
            }
        }

        PSCV_STAMP(3)
        // planes d+3, d+4 replace d-3, d-2 (last read one iteration ago, fenced by that iteration's barrier)
        stash((RING + 4) % SW_NSLOT, na);
        stash((RING + 5) % SW_NSLOT, nb);
        PSCV_STAMP(4)
        __syncthreads();
        PSCV_STAMP(5)
    };
    for (int d = dbeg; d < dend; d += 6) {
        pair(std::integral_constant<int, 0>{}, d);
        if (d + 2 < dend) pair(std::integral_constant<int, 2>{}, d + 2);
        if (d + 4 < dend) pair(std::integral_constant<int, 4>{}, d + 4);
    }
    PSCV_PROF_END(sweep, blockIdx.x)
}

// ---- kd-in-rows variant of the 32 -> 8 sweep (round 3, `pscv_set_tuning("sweep_kdm", 1)`) ---------------------------------
// The plane-pair kernel above reads every (plane, tap) B fragment for ONE 16-row MFMA: 72 ds_read_b128 + 72 MFMAs per wave and
// plane pair, 144 VGPRs of weights, 70 KiB of ring -> two workgroups per CU whose MFMA phases (2.2 K of ~5 K cycles per
// iteration, phase stamps in profiles/README.md) cover each other only by chance.  Here the three depth taps of the kernel
// sit in the ROWS of a 32 x 32 x 16 MFMA instead: rows 8 kd + c_out (kd = 0..2; rows 24..31 carry zero weights), columns =
// the wave's 2 x 16 pixels, reduction = 16 of the 32 channels.  An input plane p is then read ONCE -- 9 (kh, kw) taps x 2
// channel halves = 18 B fragments / 18 MFMAs (the same 75 % of useful MFMA rows as the pair packing) -- and feeds output planes
// p+1 (kd 0), p (kd 1), p-1 (kd 2) at the same time: the three 8-row blocks of the accumulator slide down one block per
// plane (12 register moves), and the block that leaves is a finished output plane.  Half the LDS reads and half the weight
// registers (72) of the pair kernel, a 3-slot ring (plane p being read, p+1 landed, p+2 in flight in registers; 36 KiB):
// three to four workgroups per CU.  Lane -> pixel map: the hardware serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27} /
// {4-11, 16-19, 28-31} (+32); each group gets 16 x-ADJACENT pixels of one row, and with chunk ^= (voxel >> 2) & 3 sixteen
// consecutive 64-byte voxels fall on the sixteen 16-byte slots of the 256-byte bank row for every tap offset.
// Weights: the PSCV_CONV_S1P8 packing, gathered (rows 0-7 of p_rel = kd hold kernel slice kd).
typedef __attribute__((ext_vector_type(16))) float sw_f32x16;
template <typename H> struct SwMfma32;
template <> struct SwMfma32<bf16_t> {
    __device__ static __forceinline__ sw_f32x16 run(const uint4& a, const uint4& b, const sw_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sw_bf16x8, a), __builtin_bit_cast(sw_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SwMfma32<f16_t> {
    __device__ static __forceinline__ sw_f32x16 run(const uint4& a, const uint4& b, const sw_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sw_f16x8, a), __builtin_bit_cast(sw_f16x8, b), c, 0, 0, 0);
    }
};

extern Knob g_fuse_c0;
Knob g_sweep_kdm_pd = {0, KNOB_SWEEP_KDM_PD};    // prefetch distance (planes in flight per workgroup) of the kd-in-rows kernel: 1 (default) or 2
Knob g_sweep_kdm = {0, KNOB_SWEEP_KDM};      // pscv_set_tuning("sweep_kdm", 1): kd-in-rows kernel for the 32 -> 8 layer; 2: with four workgroups per CU as the chunking target
constexpr int KM_NSLOT = 3;
constexpr int KM_PV = 192;                    // 10 x 18 = 180 voxels per plane, padded to a multiple of 16 (swizzle is slot-invariant)
constexpr int KM_PB = KM_PV * SW_VB;
constexpr int KM_LDS = KM_NSLOT * KM_PB + 96;   // + scale / bias / floor of the 8 output channels
constexpr int KM_CHUNKS = 10 * SW_BW * 4;
constexpr int KM_NLD = (KM_CHUNKS + 255) / 256;

// -DPSCV_ABLATE builds read ablation flags from pscv_set_tuning("fuse_c0", bits) (measurement only; results are wrong with any bit set)
#ifdef PSCV_ABLATE
#define KM_ABL(bit) (a.in2_cs & (bit))
#else
#define KM_ABL(bit) false
#endif
__device__ __forceinline__ int km_lds_off(int v, int chunk) { return v * SW_VB + ((chunk ^ ((v >> 2) & 3)) << 4); }

template <typename H, bool SKIP, int PD>
__global__ __launch_bounds__(256, 3) void conv3d_sweep8_kdm_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NLD = KM_NLD;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * 8, w0 = twi * 16;
    const int dbeg = dci * a.dc, dend = min(a.D, dbeg + a.dc);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, nn = lane & 31;
    // this lane's pixel: quads {0, 3, 5, 6} of a 32-lane half form one ds_read_b128 lane group -> row 0, quads {1, 2, 4, 7} -> row 1;
    // x = 4 x (rank of the quad in its group) + lane in quad, and the rank is quad >> 1 in both groups
    const int quad = nn >> 2;
    const int rr = ((0x69 >> quad) & 1) ? 0 : 1;
    const int px = (quad >> 1) * 4 + (nn & 3);
    const int row0 = wave * 2;

    // ---- A fragments: row nn = 8 kd + c_out, k = 8 half + j -> channel 16 h + 8 half + j = lane group g = 2 h + half of the packing ----
    uint4 wf[9][2];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.wpk);
        const int kdb = nn >> 3, co = nn & 7;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                wf[t][h] = kdb < 3 ? wp[(kdb * 9 + t) * 64 + (2 * h + half) * 16 + co] : make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- B fragment offsets inside a plane slot (channel half 0; half 1 = the same address ^ 32) ----
    int boff[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) boff[kh][kw] = km_lds_off((row0 + rr + kh) * SW_BW + px + kw, half);

    // ---- staging: raw buffer loads, zero padding from the descriptor bounds (as in the pair kernel) ----
    unsigned goff[NLD];
    // chunk id = tid + 256 i -> voxel (tid >> 2) + 64 i: the swizzle term (voxel >> 2) & 3 does not depend on i, so the LDS offsets of a
    // thread's chunks are loff0 + 4096 i (immediates), and only the last chunk can lie beyond the 720 of a plane
    static_assert(NLD == 3 && KM_CHUNKS > 512 && KM_CHUNKS <= 768, "three chunks per thread, the third one partial");
    const int loff0 = km_lds_off(tid >> 2, tid & 3);
    const bool lval_last = tid + 512 < KM_CHUNKS;
    const long plane_stride = (long)a.Hh * a.W * a.in_cs;
    const unsigned plane_bytes = (unsigned)(plane_stride * 2 - a.in_co * 2);
    const uint16_t* inb = a.in + (long)b * a.D * plane_stride + a.in_co;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i;
        const int v = id >> 2, c = id & 3;
        const int bh = v / SW_BW, bw = v - bh * SW_BW;
        const int gh = h0 - 1 + bh, gw = w0 - 1 + bw;
        const bool gval = id < KM_CHUNKS && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
        goff[i] = gval ? ((unsigned)(gh * a.W + gw) * (unsigned)a.in_cs + (unsigned)(c * 8)) * 2u : 0x7ffffff0u;
    }
    const int plane_hi = min(a.D - 1, dend);
    auto fetch = [&](int plane, uint4 (&reg)[NLD]) {
        const bool pv = plane >= 0 && plane <= plane_hi;                                         // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(inb + (long)(pv ? plane : 0) * plane_stride), (short)0, pv ? (int)plane_bytes : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            reg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[i], 0, 0));
    };
    auto stash = [&](int ring, const uint4 (&reg)[NLD]) {
        unsigned char* sp = smem + ring * KM_PB + loff0;
        *reinterpret_cast<uint4*>(sp) = reg[0];
        *reinterpret_cast<uint4*>(sp + 4096) = reg[1];
        if (lval_last) *reinterpret_cast<uint4*>(sp + 8192) = reg[2];
    };

    // ---- epilogue constants: rows 16 + 4 half + k of the accumulator = channels 4 half + k of the finished plane ----
    // (scale / bias / ReLU floor of the 8 channels wait in LDS behind the ring: 12 registers the MFMA loop needs more)
    const int c0 = half * 4;
    float* const epc = reinterpret_cast<float*>(smem + KM_NSLOT * KM_PB);
    if (tid < 24) {
        const int k = tid & 7, which = tid >> 3;
        epc[tid] = which == 0 ? (a.scale ? a.scale[k] : 1.0f) : which == 1 ? (a.bias ? a.bias[k] : 0.0f) : (a.floor ? a.floor[k] : 0.0f);
    }
    const int oh = h0 + row0 + rr, ow = w0 + px;
    const bool pix_ok = oh < a.Hh && ow < a.W;
    const long vrow = ((long)b * a.D * a.Hh + oh) * a.W + ow;
    const long vplane = (long)a.Hh * a.W;
    // the residual of output plane od is requested an iteration before its epilogue (SKIP only): the store then waits for nothing
    auto fetch_skip = [&](int od) -> uint2 {
        uint2 sv = make_uint2(0u, 0u);
        if (SKIP && pix_ok && od >= dbeg && od < dend)
            sv = *reinterpret_cast<const uint2*>(a.skip + (vrow + od * vplane) * a.skip_cs + a.skip_co + c0);
        return sv;
    };
    auto epilogue = [&](int od, float d0, float d1, float d2, float d3, const uint2 sv) {
        if (!pix_ok) return;
        const long vox = vrow + od * vplane;
        float y[4] = {d0, d1, d2, d3};
        const float4 sc4 = *reinterpret_cast<const float4*>(epc + c0), bi4 = *reinterpret_cast<const float4*>(epc + 8 + c0);
        const float4 fl4 = *reinterpret_cast<const float4*>(epc + 16 + c0);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, bi[4] = {bi4.x, bi4.y, bi4.z, bi4.w}, fl[4] = {fl4.x, fl4.y, fl4.z, fl4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y[k] = fmaf(y[k], sc[k], bi[k]);
            if (a.epi & PSCV_EPI_RELU_PRE) y[k] = relu_floor(y[k], fl[k]);
        }
        if (SKIP) {
            y[0] += Half16<H>::lo(sv.x); y[1] += Half16<H>::hi(sv.x);
            y[2] += Half16<H>::lo(sv.y); y[3] += Half16<H>::hi(sv.y);
        }
        if (a.epi & PSCV_EPI_RELU_POST) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = relu_floor(y[k], 0.0f);
        }
        if (a.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + vox * a.out_cs + a.out_co + c0) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + vox * a.out_cs + a.out_co + c0) =
                make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
        }
    };

    // ---- prologue: plane dbeg-1 into slot 0, planes dbeg .. dbeg+PD-1 in flight in the register FIFO ----
    uint4 nx[PD][NLD];
    {
        uint4 ra[NLD];
        fetch(dbeg - 1, ra);
#pragma unroll
        for (int s = 0; s < PD; ++s) fetch(dbeg + s, nx[s]);
        stash(0, ra);
    }
    __syncthreads();

    sw_f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    uint2 skv = make_uint2(0u, 0u);
    int sl = 0;   // slot of plane p
    for (int p0 = dbeg - 1; p0 <= dend; p0 += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            const int p = p0 + s;
            if (p > dend) break;                                     // workgroup-uniform
            // plane p+1 (requested PD iterations ago) lands in the slot plane p-2 left (last read two iterations ago, two barriers since);
            // its registers are refilled at once with plane p+1+PD
            int sn = sl + 1;
            sn = sn >= KM_NSLOT ? 0 : sn;
            stash(sn, nx[s]);
            const uint2 skc = skv;
            skv = fetch_skip(p - 1);                                 // residual of the plane that finishes in the NEXT iteration
            if (!KM_ABL(1) || p < dbeg) fetch(p + 1 + PD, nx[s]);
            __builtin_amdgcn_sched_barrier(0);

            // rows 16..23 hold output plane p-2, complete since the previous iteration; the blocks slide down one step
            const float d0 = acc[8], d1 = acc[9], d2 = acc[10], d3 = acc[11];
            acc[8] = acc[4]; acc[9] = acc[5]; acc[10] = acc[6]; acc[11] = acc[7];
            acc[4] = acc[0]; acc[5] = acc[1]; acc[6] = acc[2]; acc[7] = acc[3];
            acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f;

            // 18 (tap, channel half) steps, B fragments three steps ahead of their MFMA (the accumulator chain is serial:
            // one 32-cycle MFMA per step; an LDS read needs two to three of those to arrive)
            const unsigned char* sp = smem + sl * KM_PB;
            if (!KM_ABL(4)) {
                constexpr int AH = 3;
                uint4 xb[AH];
#pragma unroll
                for (int t = 0; t < AH; ++t) xb[t] = *reinterpret_cast<const uint4*>(sp + (boff[(t >> 1) / 3][(t >> 1) % 3] ^ ((t & 1) << 5)));
#pragma unroll
                for (int t = 0; t < 18; ++t) {
                    acc = SwMfma32<H>::run(wf[t >> 1][t & 1], xb[t % AH], acc);
                    if (t + AH < 18) {
                        const int u = t + AH;
                        xb[t % AH] = *reinterpret_cast<const uint4*>(sp + (boff[(u >> 1) / 3][(u >> 1) % 3] ^ ((u & 1) << 5)));
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, AH, 0);
#pragma unroll
                for (int t = 0; t < 18; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (t + AH < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }

            if (p - 2 >= dbeg && (!KM_ABL(2) || d0 == 123.456f)) epilogue(p - 2, d0, d1, d2, d3, skc);
            sl = sn;
            __syncthreads();
        }
    }
    epilogue(dend - 1, acc[8], acc[9], acc[10], acc[11], skv);
}

// ---- narrow-input variant: C_in = 8 or 16, C_out = 8 (the Vis-MVSNet U-Net's full-resolution layers) ---------------------
// Same sweep (8 x 16 pixel tile, 6-slot plane ring, two new planes per iteration prefetched under the MFMAs, plane-pair packed
// rows), but with 16 / 32-byte voxels the 32-deep MFMA reduction spans PLANES instead of channels:
//   C_in = 8:  k = 4 input planes (lane group g = plane d-1+g) x 8 channels; one MFMA per (kh,kw) tap -> 9 MFMAs give two
//              output planes of a 16-pixel row (the generic brick kernel issues 7 per ONE plane with half its rows idle);
//   C_in = 16: k = 2 planes (g >> 1) x 16 channels (chunk g & 1); 18 MFMAs per plane pair.
// A lane's plane is fixed, so its ring-slot base is computed once per iteration.  ds_read_b128 lane groups pair g = 0 with 1
// and g = 2 with 3: with the plane slot a multiple of 256 B (C_in = 8) the paired lanes read the same in-plane voxels of two
// planes -> 16 distinct 16-byte slots; for C_in = 16 the pair reads the two chunks of the same voxels -> even / odd slots.
// Conflict-free without a swizzle.  All weights stay in registers (36 / 72 VGPRs), 18 / 36 KiB of LDS: 4 / 3 workgroups per CU.
// Input planes and the residual (skip) values travel through a register FIFO of PD iterations (template; 1 by default).
// HBM-bound: 16 (32) B in + 16 B out per voxel; replaces BasicBlock.conv1 / conv2 (nn_utils.py:27-37) and the decoder's
// post-concat conv (nn_utils.py:238-245, 269-272) of the reference's UNet.
template <int CIN> struct ScGeom {
    static constexpr int VB = CIN * 2, CCH = CIN / 8;
    static constexpr int BH = 10, PV = 192;                  // 10 x 18 = 180 voxels per plane, padded to a multiple of 16
    static constexpr int PB = PV * VB;
    static constexpr int LDS = SW_NSLOT * PB;
    static constexpr int CHUNKS = BH * SW_BW * CCH;
    static constexpr int NLD = (CHUNKS + 255) / 256;
    static constexpr int NM = 4 * 9 * CIN / 32;              // MFMAs per plane pair and pixel row
};

// COUT = 16 (C_in = 16 only; CVP-MVSNet's full-resolution conv0 / conv0a, MVSNet's conv2): the 16 MFMA rows are the 16 output
// channels of ONE plane, so an iteration runs the tap loop twice (output planes dd and dd+1, same A fragments); the reduction
// pairs planes (o-1, o) and (o, o+1) -- the repeated plane o carries zero weights in the second set, which keeps every read inside
// the four ring slots that are already synchronised (a plane o+2 with zero weights could still be uninitialised LDS: 0 x NaN).
template <typename H, int CIN, int PD, int COUT = 8>
__global__ __launch_bounds__(256, (CIN == 8 && PD <= 2) ? 4 : (CIN == 8 || (PD == 1 && COUT == 8)) ? 3 : 2) void conv3d_sweepc_kernel(const SweepArgs a) {
    static_assert(COUT == 8 || (COUT == 16 && CIN == 16), "narrow sweep: 8|16 -> 8 or 16 -> 16");
    constexpr int NOP = COUT == 16 ? 2 : 1;      // output planes a lane finishes per iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = ScGeom<CIN>;
    constexpr int VB = G::VB, CCH = G::CCH, PB = G::PB, CHUNKS = G::CHUNKS, NLD = G::NLD, NM = G::NM;
    constexpr int R = 2;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot_ = bid >> 3, q = nwg >> 3, r_ = nwg & 7;
    int wg = (xcd < r_ ? xcd * (q + 1) : r_ * (q + 1) + (xcd - r_) * q) + slot_;
    const int dci = fast_divmod(wg, a.ndc, a.mg_dc);
    const int twi = fast_divmod(wg, a.ntw, a.mg_tw);
    const int thi = fast_divmod(wg, a.nth, a.mg_th);
    const int b = wg;
    const int h0 = thi * 8, w0 = twi * 16;
    const int dbeg = dci * a.dc, dend = min(a.D, dbeg + a.dc);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    uint4 wf[NM];
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.wpk);
#pragma unroll
        for (int m = 0; m < NM; ++m) wf[m] = wp[m * 64 + lane];
    }

    const int row0 = wave * R;
    int boff[R + 2][3];
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) boff[rr][kw] = ((row0 + rr) * SW_BW + n + kw) * VB + (CIN == 16 ? (g & 1) * 16 : 0);
    const int pl0 = CIN == 8 ? g : (g >> 1);      // this lane's plane (relative to d-1) in MFMA set 0; set 1 (C_in = 16) adds 2

    // (raw buffer loads as in the 32 -> 8 sweep: out-of-image chunks and out-of-volume planes come back as zeros from the hardware)
    // a thread owns ONE voxel of the 10 x 18 plane and load i fetches that voxel's 16-byte chunk i: chunk 0 (channels 0..7) from
    // `in`, chunk 1 (C_in = 16: channels 8..15) from `in2` -- each load instruction has its own per-plane descriptor, so the two
    // halves of a 16-channel input may live in different tensors (the decoder's cat([deconv, enc]) is never written)
    static_assert(NLD == CCH, "one load per 16-byte chunk of a voxel");
    unsigned goff[NLD];
    int loff[NLD];
    bool lval[NLD];
    long plane_stride[NLD];
    unsigned plane_bytes[NLD];
    const uint16_t* inb[NLD];
    {
        const int v = tid;
        const int bh = v / SW_BW, bw = v - bh * SW_BW;
        const int gh = h0 - 1 + bh, gw = w0 - 1 + bw;
        const bool in_tile = v < G::BH * SW_BW;
        const bool gval = in_tile && (unsigned)gh < (unsigned)a.Hh && (unsigned)gw < (unsigned)a.W;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int cs = i == 0 ? a.in_cs : a.in2_cs, co = i == 0 ? a.in_co : a.in2_co;
            const uint16_t* base = i == 0 ? a.in : a.in2;
            lval[i] = in_tile;
            plane_stride[i] = (long)a.Hh * a.W * cs;
            plane_bytes[i] = (unsigned)(plane_stride[i] * 2 - co * 2);
            inb[i] = base + (long)b * a.D * plane_stride[i] + co;
            goff[i] = gval ? (unsigned)(gh * a.W + gw) * (unsigned)cs * 2u : 0x7ffffff0u;
            loff[i] = v * VB + i * 16;
        }
    }
    const int plane_hi = min(a.D - 1, dend);
    auto fetch = [&](int plane, uint4 (&reg)[NLD]) {
        const bool pv = plane >= 0 && plane <= plane_hi;                                         // wave-uniform
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint16_t*>(inb[i] + (long)(pv ? plane : 0) * plane_stride[i]), (short)0, pv ? (int)plane_bytes[i] : 0, 0x00020000);
            reg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff[i], 0, 0));
        }
    };
    auto stash = [&](int ring, const uint4 (&reg)[NLD]) {
        unsigned char* sp = smem + ring * PB;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (lval[i]) *reinterpret_cast<uint4*>(sp + loff[i]) = reg[i];
    };

    const int c0 = COUT == 16 ? g * 4 : (g & 1) * 4;
    float sc[4], bi[4], fl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = a.scale ? a.scale[c0 + k] : 1.0f;
        bi[k] = a.bias ? a.bias[c0 + k] : 0.0f;
        fl[k] = a.floor ? a.floor[c0 + k] : 0.0f;
    }
    // this lane's output voxels: plane dd + (g >> 1), rows row0 + r, column n
    const bool col_ok = w0 + n < a.W;
    long vrow[R];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row_ok[r] = col_ok && h0 + row0 + r < a.Hh;
        vrow[r] = ((long)b * a.D * a.Hh + (h0 + row0 + r)) * a.W + w0 + n;
    }
    const long vplane = (long)a.Hh * a.W;
    auto fetch_skip = [&](int dd, uint2 (&reg)[NOP][R]) {
#pragma unroll
        for (int op = 0; op < NOP; ++op) {
            const int od = COUT == 16 ? dd + op : dd + (g >> 1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                reg[op][r] = make_uint2(0u, 0u);
                if (a.skip && od < dend && row_ok[r])
                    reg[op][r] = *reinterpret_cast<const uint2*>(a.skip + (vrow[r] + od * vplane) * a.skip_cs + a.skip_co + c0);
            }
        }
    };

    // ---- prologue: planes dbeg-1 .. dbeg+2 into ring slots 0..3; the register FIFO holds the planes and skip values of the
    // next PD iterations (loads issued 2 PD planes ahead of their use: the HBM latency spans several iterations) ----
    {
        uint4 ra[NLD], rb[NLD];
        fetch(dbeg - 1, ra); fetch(dbeg, rb);
        stash(0, ra); stash(1, rb);
        fetch(dbeg + 1, ra); fetch(dbeg + 2, rb);
        stash(2, ra); stash(3, rb);
    }
    uint4 pfa[PD][NLD], pfb[PD][NLD];
    uint2 sk[PD][NOP][R];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        fetch(dbeg + 3 + 2 * s, pfa[s]);
        fetch(dbeg + 4 + 2 * s, pfb[s]);
        fetch_skip(dbeg + 2 * s, sk[s]);
    }
    __syncthreads();

    int ring = 0;   // slot holding plane dd-1
    for (int d = dbeg; d < dend; d += 2 * PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            const int dd = d + 2 * s;
            if (dd < dend) {                                         // workgroup-uniform
                // planes dd+3, dd+4 replace dd-3, dd-2 (last read one iteration ago, fenced by that iteration's barrier); their
                // registers are refilled at once with the planes of iteration dd + 2 PD
                int s4 = ring + 4, s5 = ring + 5;
                s4 = s4 >= SW_NSLOT ? s4 - SW_NSLOT : s4;
                s5 = s5 >= SW_NSLOT ? s5 - SW_NSLOT : s5;
                stash(s4, pfa[s]);
                stash(s5, pfb[s]);
                fetch(dd + 3 + 2 * PD, pfa[s]);
                fetch(dd + 4 + 2 * PD, pfb[s]);

                sw_f32x4 acc[NOP][R];
#pragma unroll
                for (int op = 0; op < NOP; ++op)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[op][r] = sw_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int op = 0; op < NOP; ++op)
#pragma unroll
                    for (int set = 0; set < NM / 9; ++set) {
                        // COUT 8: planes dd-1 + pl0 (+2 for the second set); COUT 16: output dd+op reads (o-1, o) then (o, o+1)
                        int sl = ring + pl0 + (COUT == 16 ? op + set : 2 * set);
                        sl = sl >= SW_NSLOT ? sl - SW_NSLOT : sl;
                        const unsigned char* sp = smem + sl * PB;
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                                for (int r = 0; r < R; ++r) {
                                    const uint4 xf = *reinterpret_cast<const uint4*>(sp + boff[r + kh][kw]);
                                    acc[op][r] = SwMfma<H>::run(wf[set * 9 + kh * 3 + kw], xf, acc[op][r]);
                                }
                            // 72 MFMAs per iteration: fence the scheduler per tap row, or it hoists the independent LDS reads of all
                            // four (plane, set) blocks and spills ~150 registers
                            if (COUT == 16) __builtin_amdgcn_sched_barrier(0);
                        }
                    }

#pragma unroll
                for (int op = 0; op < NOP; ++op) {
                    const int od = COUT == 16 ? dd + op : dd + (g >> 1);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (od < dend && row_ok[r]) {
                            const long vox = vrow[r] + od * vplane;
                            float y[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                y[k] = fmaf(acc[op][r][k], sc[k], bi[k]);
                                if (a.epi & PSCV_EPI_RELU_PRE) y[k] = relu_floor(y[k], fl[k]);
                            }
                            if (a.skip) {
                                y[0] += Half16<H>::lo(sk[s][op][r].x); y[1] += Half16<H>::hi(sk[s][op][r].x);
                                y[2] += Half16<H>::lo(sk[s][op][r].y); y[3] += Half16<H>::hi(sk[s][op][r].y);
                            }
                            if (a.epi & PSCV_EPI_RELU_POST) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) y[k] = relu_floor(y[k], 0.0f);
                            }
                            if (a.out_f32) {
                                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + vox * a.out_cs + a.out_co + c0) =
                                    make_float4(y[0], y[1], y[2], y[3]);
                            } else {
                                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.out) + vox * a.out_cs + a.out_co + c0) =
                                    make_uint2(Half16<H>::pack(y[0], y[1]), Half16<H>::pack(y[2], y[3]));
                            }
                        }
                    }
                }
                fetch_skip(dd + 2 * PD, sk[s]);
                ring += 2;
                ring = ring >= SW_NSLOT ? ring - SW_NSLOT : ring;
                __syncthreads();
            }
        }
    }
}

}  // namespace pscv

template <typename H, int CIN, int PD, int COUT = 8>
static int sweepc_launch_t(pscv::SweepArgs& a, long nblk, hipStream_t st) {
    using namespace pscv;
    constexpr int lds = ScGeom<CIN>::LDS;
    {
        hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(conv3d_sweepc_kernel<H, CIN, PD, COUT>), lds);
        if (e != hipSuccess) { set_error("pscv_conv3d(sweep): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL((conv3d_sweepc_kernel<H, CIN, PD, COUT>), dim3((unsigned)nblk), dim3(256), lds, st, a);
    return 0;
}
template <typename H, int CIN, int COUT = 8>
static int sweepc_launch_pd(pscv::SweepArgs& a, long nblk, hipStream_t st) {
    // measured on MI355X over the six Vis stage shapes of BASELINE configurations 3 and 5 (scripts/kbench.py --only vis): distance 1
    // wins or ties everywhere -- deeper FIFOs cost a wave of occupancy (C_in = 16: 156 -> 182 VGPRs), and resident workgroups hide
    // more latency than registers do
    const int pd = pscv::g_sweepc_pd > 0 ? pscv::g_sweepc_pd : (COUT == 16 ? 2 : 1);   // 16 -> 16 runs at two waves per SIMD anyway
    return pd == 1 ? sweepc_launch_t<H, CIN, 1, COUT>(a, nblk, st) : pd == 2 ? sweepc_launch_t<H, CIN, 2, COUT>(a, nblk, st)
                                                                             : sweepc_launch_t<H, CIN, 3, COUT>(a, nblk, st);
}

// entry used by pscv_conv3d (conv3d.hip) for kind == PSCV_CONV_S1P8 with (c_in, c_out) = (8, 8), (16, 8) or (16, 16)
int pscv_conv3d_sweepc_launch(const void* in, int dtype, int c_in, int c_out, int in_cstride, int in_coff, const uint16_t* packed,
                              const float* scale, const float* bias, const float* floor, const void* skip, int skip_cstride,
                              int skip_coff, void* out, int out_cstride, int out_coff, int out_dtype, int B, int D, int Hh, int W,
                              int epi_flags, hipStream_t st, const void* in2, int in2_cstride, int in2_coff) {
    using namespace pscv;
    PSCV_CHECK_ARG((long)Hh * W * in_cstride * 2 < 0x7fffffffL, "pscv_conv3d(sweep): an input plane of %d x %d x %d channels exceeds 2 GiB", Hh, W, in_cstride);
    PSCV_CHECK_ARG(!in2 || (c_in == 16 && (long)Hh * W * in2_cstride * 2 < 0x7fffffffL), "pscv_conv3d(sweep): a second input tensor needs c_in = 16 and planes below 2 GiB");
    SweepArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.in2 = in2 ? reinterpret_cast<const uint16_t*>(in2) : a.in;
    a.in2_cs = in2 ? in2_cstride : in_cstride;
    a.in2_co = in2 ? in2_coff : in_coff + 8;
    a.wpk = packed; a.scale = scale; a.bias = bias; a.floor = floor;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.out_cs = out_cstride; a.out_co = out_coff; a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.D = D; a.Hh = Hh; a.W = W; a.epi = epi_flags;
    a.nth = (Hh + 7) / 8;
    a.ntw = (W + 15) / 16;
    // one resident round of workgroups (4 per CU); each depth-chunk seam re-reads two halo planes
    const long tiles = (long)B * a.nth * a.ntw;
    const long slots = g_sweepc_slots > 0 ? g_sweepc_slots : (c_out == 16 ? 512 : 768);
    const long ndc_want = tiles >= slots ? 1 : slots / tiles;
    int dc = (int)((D + ndc_want - 1) / ndc_want);
    dc = (dc + 1) & ~1;
    dc = dc < 4 ? 4 : dc;
    if (g_sweep_dc > 0) dc = g_sweep_dc & ~1;
    dc = dc > D ? ((D + 1) & ~1) : dc;
    a.dc = dc;
    a.ndc = (D + dc - 1) / dc;
    const long nblk = tiles * a.ndc;
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(sweep): bad grid %ld", nblk); return -1; }
    if (c_out == 16) return dtype == PSCV_BF16 ? sweepc_launch_pd<bf16_t, 16, 16>(a, nblk, st) : sweepc_launch_pd<f16_t, 16, 16>(a, nblk, st);
    if (c_in == 8) return dtype == PSCV_BF16 ? sweepc_launch_pd<bf16_t, 8>(a, nblk, st) : sweepc_launch_pd<f16_t, 8>(a, nblk, st);
    return dtype == PSCV_BF16 ? sweepc_launch_pd<bf16_t, 16>(a, nblk, st) : sweepc_launch_pd<f16_t, 16>(a, nblk, st);
}

// entry used by pscv_conv3d (conv3d.hip) for kind == PSCV_CONV_S1P8
PSCV_PROF_EXPORT(sweep)

int pscv_conv3d_sweep8_launch(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed,
                              const float* scale, const float* bias, const float* floor, const void* skip,
                              int skip_cstride, int skip_coff, void* out, int out_cstride, int out_coff, int out_dtype,
                              int B, int D, int Hh, int W, int epi_flags, hipStream_t st) {
    using namespace pscv;
    SweepArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.wpk = packed; a.scale = scale; a.bias = bias; a.floor = floor;
    a.skip = reinterpret_cast<const uint16_t*>(skip);
    a.out = out;
    a.in_cs = in_cstride; a.in_co = in_coff; a.skip_cs = skip_cstride; a.skip_co = skip_coff;
    a.out_cs = out_cstride; a.out_co = out_coff; a.out_f32 = out_dtype == PSCV_F32;
    a.B = B; a.D = D; a.Hh = Hh; a.W = W; a.epi = epi_flags;
    PSCV_CHECK_ARG((long)Hh * W * in_cstride * 2 < 0x7fffffffL, "pscv_conv3d(sweep): an input plane of %d x %d x %d channels exceeds 2 GiB", Hh, W, in_cstride);
    if (g_sweep_kdm) {   // kd-in-rows kernel: 3-slot ring, three (knob 2: four) workgroups per CU in one resident round
#ifdef PSCV_ABLATE
        a.in2_cs = g_fuse_c0;   // ablation flags: 1 no plane fetch, 2 no stores, 4 no LDS reads / MFMAs (scripts/dev/kdm_bench.py --ablate)
#endif
        a.nth = (Hh + 7) / 8;
        a.ntw = (W + 15) / 16;
        const long tiles = (long)B * a.nth * a.ntw;
        const long slots = g_sweep_kdm >= 2 ? 1024 : 768;
        const long ndc_want = tiles >= slots ? 1 : slots / tiles;
        int dc = (int)((D + ndc_want - 1) / ndc_want);
        dc = dc < 4 ? 4 : dc;
        if (g_sweep_dc > 0) dc = g_sweep_dc;
        dc = dc > D ? D : dc;
        a.dc = dc;
        a.ndc = (D + dc - 1) / dc;
        const long nblk = tiles * a.ndc;
        a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
        if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(sweep): bad grid %ld", nblk); return -1; }
        const int pd = g_sweep_kdm_pd > 0 ? g_sweep_kdm_pd : 1;
        const int ti = (dtype == PSCV_BF16 ? 0 : 1) + (skip ? 2 : 0) + (pd >= 2 ? 4 : 0);
#define PSCV_KDM(I, HT, SK, PDV)                                                                                                  \
        if (ti == I) {                                                                                                            \
            {                                                                                                   \
                hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(conv3d_sweep8_kdm_kernel<HT, SK, PDV>), \
                                                   KM_LDS);                           \
                if (e != hipSuccess) { set_error("pscv_conv3d(sweep): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; } \
            }                                                                                                                     \
            hipLaunchKernelGGL((conv3d_sweep8_kdm_kernel<HT, SK, PDV>), dim3((unsigned)nblk), dim3(256), KM_LDS, st, a);          \
            return 0;                                                                                                             \
        }
        PSCV_KDM(0, bf16_t, false, 1) PSCV_KDM(1, f16_t, false, 1) PSCV_KDM(2, bf16_t, true, 1) PSCV_KDM(3, f16_t, true, 1)
        PSCV_KDM(4, bf16_t, false, 2) PSCV_KDM(5, f16_t, false, 2) PSCV_KDM(6, bf16_t, true, 2) PSCV_KDM(7, f16_t, true, 2)
#undef PSCV_KDM
        return -1;
    }
    const bool tall = g_sweep_th16 && Hh >= 16;
    const int TH = tall ? 16 : 8;
    a.nth = (Hh + TH - 1) / TH;
    a.ntw = (W + 15) / 16;
    // depth chunk: the whole grid should be ONE resident round of workgroups (256 CUs x 2 for 8-row tiles, x 1 for
    // 16-row tiles): no tail round, and the fewest chunk seams (each seam re-reads 2 halo planes).  Measured at the
    // headline size: 64 planes / 480 workgroups 90 us, 12 planes / 2560 workgroups 130 us.
    const long tiles = (long)B * a.nth * a.ntw;
    const long slots = tall ? 256 : 512;
    const long ndc_want = tiles >= slots ? 1 : slots / tiles;
    int dc = (int)((D + ndc_want - 1) / ndc_want);
    dc = (dc + 1) & ~1;
    dc = dc < 4 ? 4 : dc;
    if (g_sweep_dc > 0) dc = g_sweep_dc & ~1;
    dc = dc > D ? ((D + 1) & ~1) : dc;
    a.dc = dc;
    a.ndc = (D + dc - 1) / dc;
    const long nblk = tiles * a.ndc;
    a.mg_th = fast_div_magic(a.nth); a.mg_tw = fast_div_magic(a.ntw); a.mg_dc = fast_div_magic(a.ndc);
    if (nblk <= 0 || nblk > 0x7fffffffL) { set_error("pscv_conv3d(sweep): bad grid %ld", nblk); return -1; }
    const int ti = (dtype == PSCV_BF16 ? 0 : 1) + (tall ? 2 : 0);
    const void* kern = ti == 0 ? reinterpret_cast<const void*>(conv3d_sweep8_kernel<bf16_t, 8>)
                     : ti == 1 ? reinterpret_cast<const void*>(conv3d_sweep8_kernel<f16_t, 8>)
                     : ti == 2 ? reinterpret_cast<const void*>(conv3d_sweep8_kernel<bf16_t, 16>)
                               : reinterpret_cast<const void*>(conv3d_sweep8_kernel<f16_t, 16>);
    const int lds = tall ? SwGeom<16>::LDS : SwGeom<8>::LDS;
    {
        hipError_t e = ensure_dyn_lds(kern, lds);
        if (e != hipSuccess) { set_error("pscv_conv3d(sweep): hipFuncSetAttribute: %s", hipGetErrorString(e)); return -2; }
    }
    const dim3 grid((unsigned)nblk);
    if (ti == 0) hipLaunchKernelGGL((conv3d_sweep8_kernel<bf16_t, 8>), grid, dim3(256), lds, st, a);
    else if (ti == 1) hipLaunchKernelGGL((conv3d_sweep8_kernel<f16_t, 8>), grid, dim3(256), lds, st, a);
    else if (ti == 2) hipLaunchKernelGGL((conv3d_sweep8_kernel<bf16_t, 16>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((conv3d_sweep8_kernel<f16_t, 16>), grid, dim3(512), lds, st, a);
    return 0;
}
