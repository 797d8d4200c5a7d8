// Host-only parts of the pscv C ABI: error channel, version, conv3d weight packing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <mutex>
#include <unordered_map>

#include "pscv_common.h"

namespace pscv {

// Largest dynamic-LDS size granted per (kernel, device).  A launch first looks into a small per-thread cache (no lock, no map: the
// common case of a launch-bound forward such as Vis-MVSNet's ~300 launches, or of several host threads launching on several
// streams); only a miss takes the process-wide mutex and the map, whose key is the (kernel, device) PAIR.
namespace {
struct DynLdsKey {
    const void* kernel;
    int dev;
    bool operator==(const DynLdsKey& o) const { return kernel == o.kernel && dev == o.dev; }
};
struct DynLdsHash {
    size_t operator()(const DynLdsKey& k) const { return std::hash<const void*>()(k.kernel) * 31u + std::hash<int>()(k.dev); }
};
struct DynLdsSlot { const void* kernel; int dev; int bytes; };
constexpr int DYN_LDS_SLOTS = 64;      // direct-mapped; a collision only costs the slow path
thread_local DynLdsSlot g_dyn_lds_tl[DYN_LDS_SLOTS];
}  // namespace

hipError_t ensure_dyn_lds(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    DynLdsSlot& slot = g_dyn_lds_tl[(reinterpret_cast<uintptr_t>(kernel) >> 4 ^ (uintptr_t)dev * 0x9e37u) % DYN_LDS_SLOTS];
    if (slot.kernel == kernel && slot.dev == dev && slot.bytes >= bytes) return hipSuccess;
    static std::mutex mu;
    static std::unordered_map<DynLdsKey, int, DynLdsHash> done;
    int granted;
    {
        std::lock_guard<std::mutex> lk(mu);
        const DynLdsKey key{kernel, dev};
        auto it = done.find(key);
        if (it == done.end() || it->second < bytes) {
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            done[key] = bytes;
            granted = bytes;
        } else {
            granted = it->second;
        }
    }
    slot = DynLdsSlot{kernel, dev, granted};
    return hipSuccess;
}

// Compute units of the CURRENT device (persistent kernels size their grids with it); cached per device, not per process.
int device_cu_count() {
    constexpr int MAXDEV = 64;
    static int cached[MAXDEV];          // 0 = not queried yet; racing writers store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < MAXDEV) {
        const int c = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    if (dev >= 0 && dev < MAXDEV) __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    return n;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local int g_knob_tl[KNOB_COUNT];
static thread_local unsigned g_knob_tl_mask = 0;

bool knob_thread_value(int id, int* v) {
    if (!(g_knob_tl_mask >> id & 1u)) return false;
    *v = g_knob_tl[id];
    return true;
}
void knob_thread_set(int id, int v, bool enable) {
    g_knob_tl[id] = v;
    g_knob_tl_mask = enable ? (g_knob_tl_mask | 1u << id) : (g_knob_tl_mask & ~(1u << id));
}
Knob::operator int() const {
    int v;
    return knob_thread_value(id, &v) ? v : __atomic_load_n(&process, __ATOMIC_RELAXED);
}

}  // namespace pscv

extern "C" const char* pscv_last_error(void) { return pscv::g_err; }
extern "C" int pscv_abi_version(void) { return PSCV_ABI_VERSION; }

// Packed layout: [k-step][16-channel tile][lane 0..63][8 bf16]; lane (m = lane & 15, g = lane >> 4) holds
// W[c_out = tile*16 + m][k = step*32 + g*8 + j], k = tap * c_in + ci.  Dense kinds order taps (kd, kh, kw)
// row-major; T2 concatenates its 8 output-parity classes pc = pd*4 + ph*2 + pw, each with taps ordered
// (sub_d, sub_h, sub_w) where along a parity-1 dim sub 0 is kernel index 0 (input offset +1) and sub 1 is
// kernel index 2 (offset 0), and a parity-0 dim has the single kernel index 1.
//
// Special layouts:
//   S1P8  depth-sweep [p_rel 0..3][tap (kh,kw) 0..8][lane][8]: rows 0-7 hold kernel slice kd = p_rel for output plane d,
//         rows 8-15 hold kd = p_rel - 1 for output plane d+1 (zero where kd falls outside 0..2).  c_in = 16: [pair 0..1][tap]
//         with lane group g holding plane 2 pair + (g >> 1), channels 8 (g & 1) + j; c_in = 8: [tap] with group g = plane g.
//         16 -> 16: [set 0..1][tap], rows = output channels, set 0 = kernel slices kd = (g >> 1), set 1 = kd 2 for g >> 1 = 1, zero else.
//   T2P8  parity-pair: 9 k-steps ordered (pd, ph, sub_d <= pd, sub_h <= ph); K = 32 = two W taps (input x, x+1) x 16
//         channels; rows 0-7 = output x parity 0 (kernel index kw = 1 on tap 0), rows 8-15 = parity 1 (kw = 2 on tap 0,
//         kw = 0 on tap 1).  Along D / H a parity-1 class takes kernel index 0 at input offset +1 (sub 0) and kernel
//         index 2 at offset 0 (sub 1); a parity-0 class takes kernel index 1.
//   S1C1  depth-in-rows [step][lane][8] for the 1-channel MFMA kernel (conv3d_c1.hip): row m < 6 of the A operand is output
//         plane d0 + m of a 6-plane block, the reduction runs over input planes p = 0..7 (d0 - 1 + p), taps t = kh*3 + kw
//         and channels; A[m][p, t, ci] = w[ci][kd = p - m][t], zero outside 0 <= kd <= 2.  k-step s = (q = s / 9, tap
//         t = s % 9); lane group g = lane >> 4 holds   c_in = 8: plane 4 (g >> 1) + 2 q + (g & 1), channels j (q = 0..1);
//         c_in = 16: plane 4 (g >> 1) + q, channels 8 (g & 1) + j (q = 0..3).
namespace pscv {

struct PackDesc { int c_in, c_out, kind, transposed; };

__host__ __device__ static int pk_ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static int pk_t2_ntaps(int pc) { return (1 + ((pc >> 2) & 1)) * (1 + ((pc >> 1) & 1)) * (1 + (pc & 1)); }

__host__ __device__ static long pack_count(const PackDesc& d) {
    if (d.kind == PSCV_CONV_S1P8) return (long)(4 * 9 * d.c_in / 32) * 64 * 8;
    if (d.kind == PSCV_CONV_T2P8) return 9L * 64 * 8;
    if (d.kind == PSCV_CONV_S1C1) return (long)(8 * 9 * d.c_in / 32) * 64 * 8;
    const int nt = pk_ceil_div(d.c_out, 16);
    int total_steps = 0;
    if (d.kind == PSCV_CONV_T2) {
        for (int pc = 0; pc < 8; ++pc) total_steps += pk_ceil_div(pk_t2_ntaps(pc) * d.c_in, 32);
    } else {
        total_steps = pk_ceil_div(27 * d.c_in, 32);
    }
    return (long)total_steps * nt * 64 * 8;
}

// fp32 value of packed element `idx` (one function for the host loop and the device kernel: the layouts cannot drift apart)
__host__ __device__ static float pack_value(const float* w, const PackDesc& d, long idx) {
    const int c_in = d.c_in, c_out = d.c_out;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long blk = idx >> 9;
    const int m = lane & 15, g = lane >> 4;
    if (d.kind == PSCV_CONV_S1P8) {
        // one MFMA reduces over 32 = (planes x channels): c_in 32: one plane; 16: planes 2 (blk / 9) + (g >> 1), channel half
        // g & 1; 8: planes g (conv3d_sweep.hip)
        const int t = (int)(blk % 9);
        int p, co, kd;
        const int ci = c_in == 32 ? g * 8 + j : c_in == 16 ? (g & 1) * 8 + j : j;
        if (c_out == 16) {   // 16 -> 16: rows = the 16 channels of one output plane o; set 0 = planes (o-1, o), set 1 = (o, o+1) with
            const int set = (int)(blk / 9);                      // zero weights on the repeated plane o
            co = m;
            kd = set + (g >> 1);
            if (set == 1 && (g >> 1) == 0) return 0.f;
        } else {
            p = c_in == 32 ? (int)(blk / 9) : c_in == 16 ? 2 * (int)(blk / 9) + (g >> 1) : g;
            co = m & 7;
            kd = m < 8 ? p : p - 1;
        }
        if (kd < 0 || kd > 2) return 0.f;
        if (d.transposed)    // stride-1 deconv == conv with flipped taps and swapped channel axes
            return w[((long)ci * c_out + co) * 27 + (2 - kd) * 9 + (8 - t)];
        return w[((long)co * c_in + ci) * 27 + kd * 9 + t];
    }
    if (d.kind == PSCV_CONV_T2P8) {
        const int step = (int)blk;
        // steps in (pd, ph, sd <= pd, sh <= ph) order: 0:(0,0,0,0) 1:(0,1,0,0) 2:(0,1,0,1) 3:(1,0,0,0) 4:(1,0,1,0) 5..8:(1,1,*,*)
        int pd, ph, sd, sh;
        if (step == 0) { pd = 0; ph = 0; sd = 0; sh = 0; }
        else if (step <= 2) { pd = 0; ph = 1; sd = 0; sh = step - 1; }
        else if (step <= 4) { pd = 1; ph = 0; sd = step - 3; sh = 0; }
        else { pd = 1; ph = 1; sd = (step - 5) >> 1; sh = (step - 5) & 1; }
        const int kd = pd ? (sd == 0 ? 0 : 2) : 1, kh = ph ? (sh == 0 ? 0 : 2) : 1;
        const int tap = g >> 1, ci = (g & 1) * 8 + j, co = m & 7;
        int kw;
        if (m < 8) kw = tap == 0 ? 1 : -1;
        else kw = tap == 0 ? 2 : 0;
        return kw >= 0 ? w[((long)ci * c_out + co) * 27 + (kd * 3 + kh) * 3 + kw] : 0.f;
    }
    if (d.kind == PSCV_CONV_S1C1) {
        const int s = (int)blk;
        const int q = s / 9, t = s % 9;
        const int p = c_in == 8 ? 4 * (g >> 1) + 2 * q + (g & 1) : 4 * (g >> 1) + q;
        const int ci = c_in == 8 ? j : 8 * (g & 1) + j;
        const int kd = p - m;
        return (m < 6 && kd >= 0 && kd <= 2) ? w[(long)ci * 27 + kd * 9 + t] : 0.f;
    }
    const int nt = pk_ceil_div(c_out, 16);
    const int tile = (int)(blk % nt);
    int step = (int)(blk / nt);
    int pc = 0, ntaps = 27;
    if (d.kind == PSCV_CONV_T2) {
        for (pc = 0; pc < 8; ++pc) {
            const int ns = pk_ceil_div(pk_t2_ntaps(pc) * c_in, 32);
            if (step < ns) break;
            step -= ns;
        }
        ntaps = pk_t2_ntaps(pc);
    }
    const int co = tile * 16 + m;
    const int kk = step * 32 + g * 8 + j;
    const int tap = kk / c_in, ci = kk % c_in;
    if (co >= c_out || tap >= ntaps) return 0.f;
    int kd, kh, kw;
    if (d.kind == PSCV_CONV_T2) {
        const int pd = (pc >> 2) & 1, ph = (pc >> 1) & 1, pw = pc & 1;
        const int tw = tap % (1 + pw), th = (tap / (1 + pw)) % (1 + ph), td = tap / ((1 + pw) * (1 + ph));
        kd = pd ? (td == 0 ? 0 : 2) : 1;
        kh = ph ? (th == 0 ? 0 : 2) : 1;
        kw = pw ? (tw == 0 ? 0 : 2) : 1;
    } else {
        kd = tap / 9; kh = (tap / 3) % 3; kw = tap % 3;
        if (d.transposed) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }   // stride-1 deconv == conv with flipped taps
    }
    const int kidx = (kd * 3 + kh) * 3 + kw;
    return d.transposed ? w[((long)ci * c_out + co) * 27 + kidx] : w[((long)co * c_in + ci) * 27 + kidx];
}

static int pack_check(const PackDesc& d, int dtype, const char* fn) {
    PSCV_CHECK_ARG(d.c_in > 0 && d.c_in % 8 == 0 && d.c_out > 0, "%s: bad channels %d -> %d", fn, d.c_in, d.c_out);
    PSCV_CHECK_ARG(d.kind == PSCV_CONV_S1 || d.kind == PSCV_CONV_S2 || d.kind == PSCV_CONV_T2 || d.kind == PSCV_CONV_S1P8 ||
                       d.kind == PSCV_CONV_S1C1 || d.kind == PSCV_CONV_T2P8, "%s: kind %d", fn, d.kind);
    PSCV_CHECK_ARG(d.kind != PSCV_CONV_T2 || d.transposed, "%s: T2 needs a ConvTranspose3d weight", fn);
    PSCV_CHECK_ARG(d.kind != PSCV_CONV_S2 || !d.transposed, "%s: S2 takes a Conv3d weight", fn);
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "%s: dtype %d must be bf16 or fp16", fn, dtype);
    PSCV_CHECK_ARG(d.kind != PSCV_CONV_S1P8 || ((d.c_in == 8 || d.c_in == 16 || d.c_in == 32) && d.c_out == 8) || (d.c_in == 16 && d.c_out == 16),
                   "%s: S1P8 is 8|16|32 -> 8 and 16 -> 16 only", fn);
    PSCV_CHECK_ARG(d.kind != PSCV_CONV_T2P8 || (d.c_in == 16 && d.c_out == 8 && d.transposed), "%s: T2P8 is ConvTranspose3d 16 -> 8 only", fn);
    PSCV_CHECK_ARG(d.kind != PSCV_CONV_S1C1 || (d.c_out == 1 && (d.c_in == 8 || d.c_in == 16) && !d.transposed),
                   "%s: S1C1 is Conv3d 8|16 -> 1 only", fn);
    return 0;
}

__global__ void pack_conv3d_kernel(const float* __restrict__ w, PackDesc d, int dtype, long n, uint16_t* __restrict__ packed) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = pack_value(w, d, idx);
    packed[idx] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
}

}  // namespace pscv

extern "C" long pscv_pack_conv3d_weights(const float* w, int c_in, int c_out, int kind, int transposed, int dtype,
                                         uint16_t* packed) {
    using namespace pscv;
    const PackDesc d{c_in, c_out, kind, transposed};
    if (pack_check(d, dtype, "pscv_pack_conv3d_weights")) return -1;
    const long n = pack_count(d);
    if (!packed) return n;
    PSCV_CHECK_ARG(w, "pscv_pack_conv3d_weights: null weight pointer");
    for (long i = 0; i < n; ++i) {
        const float v = pack_value(w, d, i);
        packed[i] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
    }
    return n;
}

// Same packing on the device (w and packed are device pointers): a training step repacks every layer's weights after the
// optimizer step, and the host version costs a device -> host copy and a synchronisation per layer.
extern "C" int pscv_pack_conv3d_weights_device(const float* w, int c_in, int c_out, int kind, int transposed, int dtype,
                                               uint16_t* packed, void* stream) {
    using namespace pscv;
    const PackDesc d{c_in, c_out, kind, transposed};
    if (pack_check(d, dtype, "pscv_pack_conv3d_weights_device")) return -1;
    PSCV_CHECK_ARG(w && packed, "pscv_pack_conv3d_weights_device: null pointer argument");
    const long n = pack_count(d);
    hipLaunchKernelGGL(pack_conv3d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, d,
                       dtype, n, packed);
    PSCV_CHECK_LAUNCH("pscv_pack_conv3d_weights_device");
    return 0;
}
