// CVP-MVSNet refinement hypotheses in eval mode, on the device with no host round trip (gfx950).
//
// Reference: calDepthHypo, models/CVP_MVSNet/models/modules.py:131-226 -- per batch item (a Python loop, fp64), the depth
// step that moves a pixel's projection into the FIRST source view by one pixel along its epipolar line, the MEDIAN of
// |step| over the valid pixels, then 8 hypothesis planes depth + k * median, k = -4..3.
// Three stages: (1) per-pixel |step| in fp64 -> 64-bit keys (bit patterns of non-negative doubles order like the
// values; invalid pixels get the all-ones key), (2) the lower median by an 8-pass MSB-first radix select (one launch per
// pass over many workgroups, integer histograms: exact and order-independent; a first single-workgroup version took 4.2 ms
// at 1024x1280 -- every key shares its leading digits, so the LDS atomics serialised on one bin), (3) the planes.
#include "pscv_common.h"

namespace pscv {

constexpr int HC = 39;   // doubles per batch item: Kref^-1 [9], T = E_src E_ref^-1 rows 0..2 [12], K_src [9], A [9]

__device__ __forceinline__ void mat3(const double* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = m[0] * x + m[1] * y + m[2] * z;
    oy = m[3] * x + m[4] * y + m[5] * z;
    oz = m[6] * x + m[7] * y + m[8] * z;
}

__global__ __launch_bounds__(256) void hypo_keys_kernel(const float* __restrict__ depth, const double* __restrict__ cams,
                                                        unsigned long long* __restrict__ keys, unsigned long long* __restrict__ ws,
                                                        int ws_u64, int H, int W) {
    const int b = blockIdx.y;
    // the select's histograms and state start from zero: cleared here (this launch completes before the first select pass).
    // A hipMemsetAsync node did this before; replayed from a captured hipGraph it left the previous replay's counters in place
    // (ROCm 7.2; first replay right, later ones wrong), so the kernels no longer rely on a memset.
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < ws_u64; i += 256) ws[(long)b * ws_u64 + i] = 0ull;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const double* c = cams + (long)b * HC;
    const double* Kinv = c; const double* T = c + 9; const double* Ks = c + 21; const double* A = c + 30;
    const double x = (double)(pix % W), y = (double)(pix / W);
    const double d1 = (double)depth[(long)b * H * W + pix];
    auto project = [&](double d, double& u, double& v, double& z) {
        double cx, cy, cz;
        mat3(Kinv, x * d, y * d, d, cx, cy, cz);                                   // K_ref^-1 (X d)
        const double sx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];               // into the source camera frame
        const double sy = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
        const double sz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
        double px, py, pz;
        mat3(Ks, sx, sy, sz, px, py, pz);
        u = px / pz; v = py / pz; z = pz;
    };
    double u1, v1, z1, u2, v2, z2;
    project(d1, u1, v1, z1);
    project(d1 + 1.0, u2, v2, z2);
    const double dx = u2 - u1, dy = v2 - v1;
    const double nrm = sqrt(dx * dx + dy * dy);                                    // (the homogeneous third component is 1 - 1 = 0)
    const double inv = 1.0 / fmax(nrm, 1e-8);
    const double u3 = u1 + dx * inv, v3 = v1 + dy * inv;                           // one pixel along the epipolar line
    double r0, r1, r2, c0, c1, c2;
    mat3(A, u1, v1, 1.0, r0, r1, r2);
    r1 *= z1; r2 *= z1;                                                            // rows 1..2 of z1 A x1
    mat3(A, u3, v3, 1.0, c0, c1, c2);
    // rows 1..2 of [X | A x3] (delta, .)^T = rows 1..2 of z1 A x1  ->  Cramer's rule for delta
    const double det = y * c2 - c1;
    const bool valid = nrm > 1e-8 && z1 > 1e-8 && z2 > 1e-8 && fabs(det) > 1e-8;
    unsigned long long key = ~0ull;
    if (valid) {
        const double delta = fabs((c2 * r1 - c1 * r2) / det);
        key = (delta == delta) ? (unsigned long long)__double_as_longlong(delta) : ~0ull;
    }
    keys[(long)b * H * W + pix] = key;
}

// Lower median (rank (n_valid - 1) / 2) of the keys != ~0 by an MSB-first radix select, 8 bits per pass, one launch per
// pass over many workgroups.  Per batch item the workspace holds hist[8][256] (u32, zeroed by the key kernel's launch
// companion) and state[8] = (prefix, rank) after each pass.  Launch p: every workgroup derives state[p-1] from state[p-2]
// and hist[p-1] (256-entry scan, redundantly: cheaper than another launch), workgroup 0 publishes it, then each workgroup
// histograms its slice of the keys that match the prefix into hist[p].  Lanes aggregate runs of equal digits before the
// LDS atomic (in the first passes nearly all keys share the digit: sign / exponent bits), the LDS histogram goes to
// global memory with one integer atomic per non-empty bin.
struct SelState { unsigned long long prefix; unsigned long long rank; };   // rank == ~0: no valid key

constexpr int SEL_WS_U64 = 8 * 256 / 2 + 8 * 2;   // u64 words per batch item: hist[8][256] u32 + state[8]

__device__ __forceinline__ unsigned* sel_hist(unsigned long long* ws, int b, int pass) {
    return reinterpret_cast<unsigned*>(ws + (long)b * SEL_WS_U64) + pass * 256;
}
__device__ __forceinline__ SelState* sel_state(unsigned long long* ws, int b) {
    return reinterpret_cast<SelState*>(ws + (long)b * SEL_WS_U64 + 8 * 256 / 2);
}

// state after pass `done` (0-based) from the state before it and that pass's histogram, by all 256 threads of the block (bin
// tid each): wave prefix sums + a 4-entry cross-wave table, then the ONE thread whose bin holds the rank publishes the state in
// LDS (bins [excl, incl) are disjoint, and bin 255 takes whatever is left -- the serial form's fall-through).  A first version
// walked the 256 bins from one thread: ~10 us of dependent loads per pass, 8 passes per call.
__device__ __forceinline__ void sel_advance(const unsigned* __restrict__ hist, SelState prev, int done, SelState* out, unsigned* wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int shift = 56 - 8 * done;
    const unsigned h = hist[tid];
    unsigned incl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned t = wave_tot[i];
        if (i < wv) base += t;
        total += t;
    }
    incl += base;
    const unsigned excl = incl - h;
    unsigned long long rank = prev.rank;
    if (done == 0) rank = total ? (unsigned long long)((total - 1) / 2) : ~0ull;
    if (rank == ~0ull) {
        if (tid == 0) { out->prefix = 0; out->rank = ~0ull; }
    } else if ((unsigned long long)excl <= rank && (rank < (unsigned long long)incl || tid == 255)) {
        out->prefix = prev.prefix | ((unsigned long long)tid << shift);
        out->rank = rank - excl;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void hypo_select_pass_kernel(const unsigned long long* __restrict__ keys, unsigned long long* __restrict__ ws,
                                                               int n, int pass) {
    __shared__ unsigned hist[256];
    __shared__ SelState st;
    const int b = blockIdx.y, tid = threadIdx.x;
    __shared__ unsigned wave_tot[4];
    hist[tid] = 0;
    {
        SelState prev{0, 0};
        if (pass >= 2) prev = sel_state(ws, b)[pass - 2];
        if (pass >= 1) {
            sel_advance(sel_hist(ws, b, pass - 1), prev, pass - 1, &st, wave_tot);
            if (blockIdx.x == 0 && tid == 0) sel_state(ws, b)[pass - 1] = st;
        } else {
            if (tid == 0) st = prev;
            __syncthreads();
        }
    }
    if (st.rank == ~0ull) return;
    const int shift = 56 - 8 * pass;
    const unsigned long long prefix = st.prefix;
    const unsigned long long* k = keys + (long)b * n;
    int run_digit = -1;
    unsigned run = 0;
    for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256) {
        const unsigned long long v = k[i];
        if (v == ~0ull) continue;
        if (pass > 0 && (v >> (shift + 8)) != (prefix >> (shift + 8))) continue;
        const int dgt = (int)((v >> shift) & 255);
        if (dgt != run_digit) {
            if (run) atomicAdd(&hist[run_digit], run);
            run_digit = dgt; run = 0;
        }
        ++run;
    }
    if (run) atomicAdd(&hist[run_digit], run);
    __syncthreads();
    if (hist[tid]) atomicAdd(sel_hist(ws, b, pass) + tid, hist[tid]);
}

// the last advance (redundantly per block, like the passes) + the 8 planes depth + k * median
__global__ __launch_bounds__(256) void hypo_planes_kernel(const float* __restrict__ depth, unsigned long long* __restrict__ ws,
                                                          const float* __restrict__ fallback, double* __restrict__ steps,
                                                          float* __restrict__ hypos, int hw) {
    __shared__ SelState st;
    __shared__ unsigned wave_tot[4];
    const int b = blockIdx.y;
    sel_advance(sel_hist(ws, b, 7), sel_state(ws, b)[6], 7, &st, wave_tot);
    const double s = st.rank == ~0ull ? (double)fallback[b] : __longlong_as_double((long long)st.prefix);
    if (blockIdx.x == 0 && threadIdx.x == 0) steps[b] = s;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const double d = (double)depth[(long)b * hw + pix];
#pragma unroll
    for (int k = 0; k < 8; ++k) hypos[((long)b * 8 + k) * hw + pix] = (float)(d + (double)(k - 4) * s);
}

}  // namespace pscv

using namespace pscv;

extern "C" int pscv_cvp_depth_hypos(const float* depth, const double* cams, const float* fallback, unsigned long long* keys,
                                    double* steps, float* hypos, int B, int H, int W, void* stream) {
    PSCV_CHECK_ARG(depth && cams && fallback && keys && steps && hypos, "pscv_cvp_depth_hypos: null pointer argument");
    PSCV_CHECK_ARG(B > 0 && H > 0 && W > 0 && (long)H * W < (1L << 30), "pscv_cvp_depth_hypos: bad sizes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hw = H * W;
    unsigned long long* ws = keys + (long)B * hw;
    hipLaunchKernelGGL(hypo_keys_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, st, depth, cams, keys, ws, SEL_WS_U64, H, W);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(keys)");
    int nsel = (hw + 256 * 16 - 1) / (256 * 16);
    if (nsel > 1024) nsel = 1024;
    for (int pass = 0; pass < 8; ++pass)
        hipLaunchKernelGGL(hypo_select_pass_kernel, dim3(nsel, B), dim3(256), 0, st, keys, ws, hw, pass);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(select)");
    hipLaunchKernelGGL(hypo_planes_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, st, depth, ws, fallback, steps, hypos, hw);
    PSCV_CHECK_LAUNCH("pscv_cvp_depth_hypos(planes)");
    return 0;
}
