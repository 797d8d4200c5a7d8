// Host-only parts of the pscv C ABI: error channel, version, conv3d weight packing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pscv_common.h"

namespace pscv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// constexpr helpers duplicated from conv3d.hip's geometry (kept in sync by tests/test_pack_weights.py,
// which re-derives the packed layout in numpy)
static int ceil_div(int a, int b) { return (a + b - 1) / b; }
static int t2_ntaps(int pc) { return (1 + ((pc >> 2) & 1)) * (1 + ((pc >> 1) & 1)) * (1 + (pc & 1)); }

}  // namespace pscv

extern "C" const char* pscv_last_error(void) { return pscv::g_err; }
extern "C" int pscv_abi_version(void) { return PSCV_ABI_VERSION; }

// Packed layout: [k-step][16-channel tile][lane 0..63][8 bf16]; lane (m = lane & 15, g = lane >> 4) holds
// W[c_out = tile*16 + m][k = step*32 + g*8 + j], k = tap * c_in + ci.  Dense kinds order taps (kd, kh, kw)
// row-major; T2 concatenates its 8 output-parity classes pc = pd*4 + ph*2 + pw, each with taps ordered
// (sub_d, sub_h, sub_w) where along a parity-1 dim sub 0 is kernel index 0 (input offset +1) and sub 1 is
// kernel index 2 (offset 0), and a parity-0 dim has the single kernel index 1.
extern "C" long pscv_pack_conv3d_weights(const float* w, int c_in, int c_out, int kind, int transposed, int dtype,
                                         uint16_t* packed) {
    using namespace pscv;
    PSCV_CHECK_ARG(c_in > 0 && c_in % 8 == 0 && c_out > 0, "pscv_pack_conv3d_weights: bad channels %d -> %d", c_in, c_out);
    PSCV_CHECK_ARG(kind == PSCV_CONV_S1 || kind == PSCV_CONV_S2 || kind == PSCV_CONV_T2 || kind == PSCV_CONV_S1P8 ||
                       kind == PSCV_CONV_S1C1 || kind == PSCV_CONV_T2P8,
                   "pscv_pack_conv3d_weights: kind %d", kind);
    PSCV_CHECK_ARG(kind != PSCV_CONV_T2 || transposed, "pscv_pack_conv3d_weights: T2 needs a ConvTranspose3d weight");
    PSCV_CHECK_ARG(kind != PSCV_CONV_S2 || !transposed, "pscv_pack_conv3d_weights: S2 takes a Conv3d weight");
    PSCV_CHECK_ARG(dtype == PSCV_BF16 || dtype == PSCV_F16, "pscv_pack_conv3d_weights: dtype %d must be bf16 or fp16", dtype);
    if (kind == PSCV_CONV_S1P8) {
        // Depth-sweep layout [p_rel 0..3][tap (kh,kw) 0..8][lane][8]: rows 0-7 hold kernel slice kd = p_rel for
        // output plane d, rows 8-15 hold kd = p_rel - 1 for output plane d+1 (zero where kd falls outside 0..2).
        PSCV_CHECK_ARG(c_in == 32 && c_out == 8 && !transposed, "pscv_pack_conv3d_weights: S1P8 is Conv3d 32 -> 8 only");
        const long n = 4L * 9 * 64 * 8;
        if (!packed) return n;
        PSCV_CHECK_ARG(w, "pscv_pack_conv3d_weights: null weight pointer");
        for (int p = 0; p < 4; ++p)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int m = lane & 15, ci = (lane >> 4) * 8 + j;
                        const int co = m & 7, kd = m < 8 ? p : p - 1;
                        float v = 0.f;
                        if (kd >= 0 && kd <= 2) v = w[((long)co * c_in + ci) * 27 + kd * 9 + t];
                        packed[(((long)p * 9 + t) * 64 + lane) * 8 + j] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
                    }
        return n;
    }
    if (kind == PSCV_CONV_T2P8) {
        // Parity-pair layout: 9 k-steps ordered (pd, ph, sub_d <= pd, sub_h <= ph); K = 32 = two W taps (input x, x+1)
        // x 16 channels; rows 0-7 = output x parity 0 (kernel index kw = 1 on tap 0), rows 8-15 = parity 1 (kw = 2 on
        // tap 0, kw = 0 on tap 1).  Along D / H a parity-1 class takes kernel index 0 at input offset +1 (sub 0) and
        // kernel index 2 at offset 0 (sub 1); a parity-0 class takes kernel index 1.
        PSCV_CHECK_ARG(c_in == 16 && c_out == 8 && transposed, "pscv_pack_conv3d_weights: T2P8 is ConvTranspose3d 16 -> 8 only");
        const long n = 9L * 64 * 8;
        if (!packed) return n;
        PSCV_CHECK_ARG(w, "pscv_pack_conv3d_weights: null weight pointer");
        int step = 0;
        for (int pd = 0; pd < 2; ++pd)
            for (int ph = 0; ph < 2; ++ph)
                for (int sd = 0; sd <= pd; ++sd)
                    for (int sh = 0; sh <= ph; ++sh, ++step) {
                        const int kd = pd ? (sd == 0 ? 0 : 2) : 1, kh = ph ? (sh == 0 ? 0 : 2) : 1;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int m = lane & 15, gk = lane >> 4;
                                const int tap = gk >> 1, ci = (gk & 1) * 8 + j, co = m & 7;
                                int kw = -1;
                                if (m < 8) kw = tap == 0 ? 1 : -1;
                                else kw = tap == 0 ? 2 : 0;
                                float v = 0.f;
                                if (kw >= 0) v = w[((long)ci * c_out + co) * 27 + (kd * 3 + kh) * 3 + kw];
                                packed[((long)step * 64 + lane) * 8 + j] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
                            }
                    }
        return n;
    }
    if (kind == PSCV_CONV_S1C1) {
        // Depth-in-rows layout [step][lane][8] for the 1-channel MFMA kernel (conv3d_c1.hip): row m < 6 of the A operand
        // is output plane d0 + m of a 6-plane block, the reduction runs over input planes p = 0..7 (d0 - 1 + p), taps
        // t = kh*3 + kw and channels; A[m][p, t, ci] = w[ci][kd = p - m][t], zero outside 0 <= kd <= 2.
        // k-step s = (q = s / 9, tap t = s % 9); lane group g = lane >> 4 holds
        //   c_in = 8:  plane 4 (g >> 1) + 2 q + (g & 1), channels j            (q = 0..1)
        //   c_in = 16: plane 4 (g >> 1) + q, channels 8 (g & 1) + j            (q = 0..3)
        PSCV_CHECK_ARG(c_out == 1 && (c_in == 8 || c_in == 16) && !transposed, "pscv_pack_conv3d_weights: S1C1 is Conv3d 8|16 -> 1 only");
        const int nsteps = 8 * 9 * c_in / 32;
        const long n = (long)nsteps * 64 * 8;
        if (!packed) return n;
        PSCV_CHECK_ARG(w, "pscv_pack_conv3d_weights: null weight pointer");
        for (int s = 0; s < nsteps; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = lane & 15, g = lane >> 4;
                    const int q = s / 9, t = s % 9;
                    const int p = c_in == 8 ? 4 * (g >> 1) + 2 * q + (g & 1) : 4 * (g >> 1) + q;
                    const int ci = c_in == 8 ? j : 8 * (g & 1) + j;
                    const int kd = p - m;
                    float v = 0.f;
                    if (m < 6 && kd >= 0 && kd <= 2) v = w[(long)ci * 27 + kd * 9 + t];
                    packed[((long)s * 64 + lane) * 8 + j] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
                }
        return n;
    }
    const int nt = ceil_div(c_out, 16);
    int total_steps = 0;
    if (kind == PSCV_CONV_T2) {
        for (int pc = 0; pc < 8; ++pc) total_steps += ceil_div(t2_ntaps(pc) * c_in, 32);
    } else {
        total_steps = ceil_div(27 * c_in, 32);
    }
    const long n_elem = (long)total_steps * nt * 64 * 8;
    if (!packed) return n_elem;
    PSCV_CHECK_ARG(w, "pscv_pack_conv3d_weights: null weight pointer");

    // weight accessor in (co, ci, kd, kh, kw) terms of the *transposed-or-not* source tensor
    auto W = [&](int co, int ci, int kd, int kh, int kw) -> float {
        const int kidx = (kd * 3 + kh) * 3 + kw;
        return transposed ? w[((long)ci * c_out + co) * 27 + kidx] : w[((long)co * c_in + ci) * 27 + kidx];
    };
    auto put = [&](int step, int tile, int lane, int j, float v) {
        packed[(((long)step * nt + tile) * 64 + lane) * 8 + j] = dtype == PSCV_BF16 ? f32_to_bf16(v) : f32_to_f16_bits(v);
    };

    int step0 = 0;
    const int nclass = kind == PSCV_CONV_T2 ? 8 : 1;
    for (int pc = 0; pc < nclass; ++pc) {
        const int pd = (pc >> 2) & 1, ph = (pc >> 1) & 1, pw = pc & 1;
        const int ntaps = kind == PSCV_CONV_T2 ? t2_ntaps(pc) : 27;
        const int nsteps = ceil_div(ntaps * c_in, 32);
        for (int s = 0; s < nsteps; ++s)
            for (int tile = 0; tile < nt; ++tile)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = tile * 16 + (lane & 15);
                        const int kk = s * 32 + (lane >> 4) * 8 + j;
                        const int tap = kk / c_in, ci = kk % c_in;
                        float v = 0.f;
                        if (co < c_out && tap < ntaps) {
                            int kd, kh, kw;
                            if (kind == PSCV_CONV_T2) {
                                const int tw = tap % (1 + pw), th = (tap / (1 + pw)) % (1 + ph), td = tap / ((1 + pw) * (1 + ph));
                                kd = pd ? (td == 0 ? 0 : 2) : 1;
                                kh = ph ? (th == 0 ? 0 : 2) : 1;
                                kw = pw ? (tw == 0 ? 0 : 2) : 1;
                            } else {
                                kd = tap / 9; kh = (tap / 3) % 3; kw = tap % 3;
                                if (transposed) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }   // stride-1 deconv == conv with flipped taps
                            }
                            v = W(co, ci, kd, kh, kw);
                        }
                        put(step0 + s, tile, lane, j, v);
                    }
        step0 += nsteps;
    }
    return n_elem;
}
