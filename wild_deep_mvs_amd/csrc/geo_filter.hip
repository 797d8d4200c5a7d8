// Geometric-consistency filter of a depth map against its source views (the step AFTER the plane-sweep path).  gfx950.
//
// One lane per reference pixel, one pass over the source views: unproject with the reference depth, project into
// source i, sample that view's depth map (bilinear, zero padding), unproject with the sampled depth, project back, and
// threshold the reprojection error, the relative depth difference and the triangulation angle.  Nothing is
// materialised (the reference builds N x h x w x 3 point clouds and grids on the CPU); the kernel reads N + 1 depth
// maps and writes three byte masks.  Arithmetic is fp32 in the reference's operation order, so the masks differ from
// the reference's only on pixels whose tested quantity sits within rounding of a threshold (tests/).
//
// Replaces (fdarmon/wild_deep_mvs): evaluation/filtering.py:60-83 with utils/utils_3D.py unproject :116-141,
// project_all :64-74, normalize :243-272, unproj_all :144-160, project :96-113, compute_triangulation_angles :300-315.
#include "pscv_common.h"

namespace pscv {

struct GeoArgs {
    const float* depth;                    // [h,w] reference depth
    const float* src[PSCV_GEO_MAX_SRC];    // source depth maps [src_h[i], src_w[i]]
    int src_h[PSCV_GEO_MAX_SRC], src_w[PSCV_GEO_MAX_SRC];
    const float* cams;                     // [n_src + 1][PSCV_GEO_CAM_FLOATS]: K, K^-1, R (row-major 3x3 each), t; view 0 = reference
    uint8_t* mask_depth;
    uint8_t* mask_disp;
    uint8_t* geo_mask;
    int* counts;                           // optional [3][h][w]: per-criterion number of consistent sources
    int n_src, h, w, need;
    float max_reproj, depth_thr, min_tri;
};

// row-vector times matrix^T, i.e. M v, accumulated k = 0, 1, 2 like a 3-wide GEMM row
__device__ __forceinline__ void mat_vec(const float* M, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = fmaf(M[2], z, fmaf(M[1], y, M[0] * x));
    oy = fmaf(M[5], z, fmaf(M[4], y, M[3] * x));
    oz = fmaf(M[8], z, fmaf(M[7], y, M[6] * x));
}
// row-vector times matrix, i.e. M^T v
__device__ __forceinline__ void matT_vec(const float* M, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = fmaf(M[6], z, fmaf(M[3], y, M[0] * x));
    oy = fmaf(M[7], z, fmaf(M[4], y, M[1] * x));
    oz = fmaf(M[8], z, fmaf(M[5], y, M[2] * x));
}

__global__ __launch_bounds__(256) void geo_filter_kernel(const GeoArgs a) {
    __shared__ float cam_lds[(PSCV_GEO_MAX_SRC + 1) * PSCV_GEO_CAM_FLOATS];
    for (int i = threadIdx.x; i < (a.n_src + 1) * PSCV_GEO_CAM_FLOATS; i += 256) cam_lds[i] = a.cams[i];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= a.h * a.w) return;
    const int y = pix / a.w, x = pix - y * a.w;
    const float xf = (float)x, yf = (float)y;
    const float d = a.depth[pix];

    const float* K0 = cam_lds;
    const float* Ki0 = cam_lds + 9;
    const float* R0 = cam_lds + 18;
    const float* t0 = cam_lds + 27;
    // X = R_0^T (K_0^-1 (x, y, 1) d - t_0)                                             utils_3D.py:116-141
    float ax, ay, az, X, Y, Z;
    mat_vec(Ki0, xf * d, yf * d, d, ax, ay, az);
    matT_vec(R0, ax - t0[0], ay - t0[1], az - t0[2], X, Y, Z);
    // reference ray: X - c_0, c_0 = -R_0^T t_0                                          utils_3D.py:308
    float c0x, c0y, c0z;
    matT_vec(R0, t0[0], t0[1], t0[2], c0x, c0y, c0z);
    const float r1x = X + c0x, r1y = Y + c0y, r1z = Z + c0z;
    const float n1 = fmaxf(sqrtf(r1x * r1x + r1y * r1y + r1z * r1z), 1e-12f);

    int n_depth = 0, n_disp = 0, n_geo = 0;
    for (int i = 0; i < a.n_src; ++i) {
        const float* K = cam_lds + (i + 1) * PSCV_GEO_CAM_FLOATS;
        const float* Ki = K + 9;
        const float* R = K + 18;
        const float* t = K + 27;
        // p = K_i (R_i X + t_i)                                                         utils_3D.py:64-74
        float cx, cy, cz, px, py, pz;
        mat_vec(R, X, Y, Z, cx, cy, cz);
        mat_vec(K, cx + t[0], cy + t[1], cz + t[2], px, py, pz);
        const float zc = fmaxf(pz, 1e-6f);
        const float u = px / zc, v = py / zc;
        // normalize with (size - 1), sample with align_corners=False                   utils_3D.py:267-268, filtering.py:66-68
        const int hs = a.src_h[i], ws = a.src_w[i];
        const float gx = 2.0f * u / ((float)ws - 1.0f) - 1.0f;
        const float gy = 2.0f * v / ((float)hs - 1.0f) - 1.0f;
        const float ix = ((gx + 1.0f) * (float)ws - 1.0f) / 2.0f;
        const float iy = ((gy + 1.0f) * (float)hs - 1.0f) / 2.0f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
        const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
        const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
        float dw = 0.0f;
        // (a NaN / out-of-range coordinate fails every bounds test and samples 0, like grid_sample's zero padding)
        if (ix > -2.0f && ix < (float)ws + 1.0f && iy > -2.0f && iy < (float)hs + 1.0f) {
            const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
            const float* sp = a.src[i];
            const bool vx0 = (unsigned)x0 < (unsigned)ws, vx1 = (unsigned)x1 < (unsigned)ws;
            const bool vy0 = (unsigned)y0 < (unsigned)hs, vy1 = (unsigned)y1 < (unsigned)hs;
            if (vx0 && vy0) dw = fmaf(sp[y0 * ws + x0], nw, dw);
            if (vx1 && vy0) dw = fmaf(sp[y0 * ws + x1], ne, dw);
            if (vx0 && vy1) dw = fmaf(sp[y1 * ws + x0], sw, dw);
            if (vx1 && vy1) dw = fmaf(sp[y1 * ws + x1], se, dw);
        }
        // X' = R_i^T (K_i^-1 (u, v, 1) d_i - t_i);  q = K_0 (R_0 X' + t_0)              utils_3D.py:144-160, 96-106
        float bx, by, bz, Xr, Yr, Zr, ex, ey, ez, qx, qy, qz;
        mat_vec(Ki, u * dw, v * dw, dw, bx, by, bz);
        matT_vec(R, bx - t[0], by - t[1], bz - t[2], Xr, Yr, Zr);
        mat_vec(R0, Xr, Yr, Zr, ex, ey, ez);
        mat_vec(K0, ex + t0[0], ey + t0[1], ez + t0[2], qx, qy, qz);
        const float zr = qz + 1e-6f;
        const float rx = qx / zr - xf, ry = qy / zr - yf;
        const bool disp_ok = sqrtf(rx * rx + ry * ry) < a.max_reproj;                                   // filtering.py:73-74
        const bool depth_ok = fabsf(zr - d) < fmaxf(zr, d) * a.depth_thr && zr > 0.0f && pz > 0.0f;     // filtering.py:76-77
        // triangulation angle between X - c_0 and X - c_i, degrees                      utils_3D.py:300-315
        float cix, ciy, ciz;
        matT_vec(R, t[0], t[1], t[2], cix, ciy, ciz);
        const float r2x = X + cix, r2y = Y + ciy, r2z = Z + ciz;
        const float n2 = fmaxf(sqrtf(r2x * r2x + r2y * r2y + r2z * r2z), 1e-12f);
        float cosv = (r1x * r2x + r1y * r2y + r1z * r2z) / n1 / n2;
        cosv = fminf(fmaxf(cosv, -1.0f), 1.0f);
        const bool tri_ok = acosf(cosv) / 3.14159274f * 180.0f > a.min_tri;
        n_depth += depth_ok;
        n_disp += disp_ok;
        n_geo += depth_ok && disp_ok && tri_ok;
    }
    if (a.mask_depth) a.mask_depth[pix] = n_depth >= a.need;
    if (a.mask_disp) a.mask_disp[pix] = n_disp >= a.need;
    if (a.geo_mask) a.geo_mask[pix] = n_geo >= a.need;
    if (a.counts) {
        const int hw = a.h * a.w;
        a.counts[pix] = n_depth; a.counts[hw + pix] = n_disp; a.counts[2 * hw + pix] = n_geo;
    }
}

}  // namespace pscv

extern "C" int pscv_geo_filter(const float* depth, const float* const* src_depth, const int* src_hw, int n_src,
                               const float* cams, int h, int w, float max_reproj_error, float depth_threshold,
                               float min_tri_angle, int num_consistent, unsigned char* mask_depth,
                               unsigned char* mask_disp, unsigned char* geo_mask, int* counts, void* stream) {
    using namespace pscv;
    PSCV_CHECK_ARG(depth && src_depth && src_hw && cams, "pscv_geo_filter: null pointer argument");
    PSCV_CHECK_ARG(n_src >= 1 && n_src <= PSCV_GEO_MAX_SRC, "pscv_geo_filter: n_src=%d outside [1,%d]", n_src, PSCV_GEO_MAX_SRC);
    PSCV_CHECK_ARG(h > 0 && w > 0 && (long)h * w < (1L << 31), "pscv_geo_filter: bad size %dx%d", h, w);
    GeoArgs a;
    a.depth = depth;
    for (int i = 0; i < PSCV_GEO_MAX_SRC; ++i) {
        a.src[i] = i < n_src ? src_depth[i] : nullptr;
        a.src_h[i] = i < n_src ? src_hw[2 * i] : 1;
        a.src_w[i] = i < n_src ? src_hw[2 * i + 1] : 1;
        if (i < n_src) {
            PSCV_CHECK_ARG(src_depth[i], "pscv_geo_filter: src_depth[%d] is null", i);
            PSCV_CHECK_ARG(a.src_h[i] > 1 && a.src_w[i] > 1 && (long)a.src_h[i] * a.src_w[i] < (1L << 31),
                           "pscv_geo_filter: source %d has bad size %dx%d", i, a.src_h[i], a.src_w[i]);
        }
    }
    a.cams = cams;
    a.mask_depth = mask_depth; a.mask_disp = mask_disp; a.geo_mask = geo_mask; a.counts = counts;
    a.n_src = n_src; a.h = h; a.w = w; a.need = num_consistent - 1;
    a.max_reproj = max_reproj_error; a.depth_thr = depth_threshold; a.min_tri = min_tri_angle;
    const long nblk = ((long)h * w + 255) / 256;
    hipLaunchKernelGGL(geo_filter_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    PSCV_CHECK_LAUNCH("pscv_geo_filter");
    return 0;
}
