/*
 * pscv.h -- C ABI of the MI355X (gfx950) plane-sweep cost-volume engine.
 *
 * The reference (fdarmon/wild_deep_mvs) has no FFI or operator registry: its hot
 * path is a chain of ATen calls inside models/<arch> forward().  This header is
 * the boundary a maintainer would bind instead of those calls (ctypes stub in
 * INTEGRATION.md; the in-tree binding is wild_deep_mvs_amd/_lib.py).  Each entry
 * point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every tensor pointer is a DEVICE pointer
 *     borrowed for the duration of the call unless marked "host";
 *   - asynchronous on the given hipStream_t (passed as void*); no allocation,
 *     no global mutable state, re-entrant per stream;
 *   - return 0 on success, negative on error; the message is available from
 *     pscv_last_error() (thread-local);
 *   - tensors are channels-last: feature maps [B,h,w,C], volumes [B,D,h,w,C];
 *     dtype codes select fp32, bf16 or fp16 STORAGE, arithmetic is always fp32
 *     (bf16 / fp16 MFMA products accumulate in fp32).
 */
#ifndef PSCV_H
#define PSCV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSCV_ABI_VERSION 9

/* storage dtypes */
#define PSCV_F32 0
#define PSCV_BF16 1
#define PSCV_F16 2 /* IEEE half, stores saturate at +-65504 */

/* sampling geometry of the warp */
#define PSCV_GEOM_PROJ 0  /* q = rot*(x,y,1)*d + trans, integer pixel centres, sample at index (u,v):
                             MVSNet models/MVSNet/module.py:127-166, CVP models/CVP_MVSNet/models/modules.py:74-128 */
#define PSCV_GEOM_HOMOG 1 /* hom = A*(x+.5,y+.5,1) - Bm*(..)/(d+1e-9), sample at index u*(W-1)/W:
                             Vis-MVSNet models/VisMVSNet/homography.py:23-120 */

/* cost aggregation fused behind the warp */
#define PSCV_COST_VARIANCE 0     /* sum f^2/N - (sum f)^2/N^2          models/MVSNet/model.py:113-139 */
#define PSCV_COST_VARIANCE_CVP 1 /* sum f^2/N - (sum f / N)^2          models/CVP_MVSNet/models/net.py:129-152, modules.py:229-293 */
#define PSCV_COST_SOFTMIN 2      /* sum_v e diff/(sum_v e + 1e-6)      models/MVSNet/model.py:141-173 */
#define PSCV_COST_GROUPCORR 3    /* per source, 4-channel group dot    models/VisMVSNet/nn_utils.py:473-490 (call model_cas.py:340) */
#define PSCV_COST_WARP_ONLY 4    /* per source warped volume           homo_warping / homography_warping themselves */
#define PSCV_COST_VARIANCE_PARTIAL 5 /* fp32 partial sums (sum f, sum f^2) over the GIVEN views for a source-view shard across GPUs:
                                    out [2][B,D,h,w,C] fp32; ref may be NULL (only one rank adds the reference view).  The ranks'
                                    outputs are all-reduced (RCCL) and pscv_variance_finish turns them into the cost volume */

#define PSCV_MAX_SRC 16
#define PSCV_CAM_FLOATS 18 /* per (source, batch): PROJ = rot[9] trans[3] pad[6]; HOMOG = A[9] Bm[9] */

/* conv3d kinds */
#define PSCV_CONV_S1 0 /* Conv3d k3 s1 p1 (also ConvTranspose3d k3 s1 p1, packed flipped) */
#define PSCV_CONV_S2 1 /* Conv3d k3 s2 p1 */
#define PSCV_CONV_T2 2 /* ConvTranspose3d k3 s2 p1 output_padding 1 */
#define PSCV_CONV_S1P8 3 /* Conv3d k3 s1 p1 (or ConvTranspose3d k3 s1 p1) with c_in = 8, 16 or 32 and c_out = 8, or 16 -> 16, on the depth-sweep
                            kernels (plane-pair packed MFMA rows, register-resident weights); same result as PSCV_CONV_S1, own
                            packed layout */
#define PSCV_CONV_T2P8 5 /* ConvTranspose3d k3 s2 p1 op1 with c_in = 16, c_out = 8 on the parity-pair packed kernel; same
                            result as PSCV_CONV_T2, own packed layout */
#define PSCV_CONV_S1C1 4 /* Conv3d k3 s1 p1 with c_out = 1, c_in = 8 or 16 (the `prob` heads) on the depth-in-rows MFMA
                            kernel (six output planes share one MFMA tile); same result as PSCV_CONV_S1, own packed layout */

/* epilogue flags for pscv_conv3d */
#define PSCV_EPI_RELU_PRE 1  /* y = max(y, 0) before the skip add   (MVSNet/CVP: skip + relu(bn(deconv))) */
#define PSCV_EPI_RELU_POST 2 /* y = max(y, 0) after the skip add    (Vis BasicBlock: relu(bn(conv) + residual)) */

const char* pscv_last_error(void);
int pscv_abi_version(void);

/* Tuning knobs for measurement runs (not part of the reference's surface).  A knob holds ONE process-wide value (a relaxed
 * atomic; every launching host thread reads it, including PyTorch's autograd thread and DataParallel's replica threads) and an
 * optional per-thread override (pscv_set_tuning_thread).  Kernel selection never depends on anything else that is mutable.  Keys:
 *   "warp_q2"   1 (default): 32-channel 16-bit sweeps run on the quad-mapped kernel (one texel per lane quad, two depth
 *               planes per quad); 0: always the generic kernel.  "warp_lpv" != 0 also selects the generic kernel.
 *   "warp_lpv"  lanes sharing one voxel in the generic pscv_warp_cost kernel (1, 2 or 4 for C=32; 0 = default)
 *   "warp_ppd"  depth planes per workgroup in pscv_warp_cost (0 = default: 8 direct kernels, 32 LDS-staged kernel)
 *   "warp_tiled" 1 (default; -1 restores it; 2 = the same): pscv_warp_cost stages the source patches of a reference tile in LDS
 *               as fp32 where it applies (C = 32, 16-bit features, per-batch planes, PROJ geometry, 1-4 source views, variance /
 *               softmin) -- same bits as the direct-gather kernels; 0: always the direct-gather kernels.  "warp_lpv" != 0
 *               also selects the direct kernel.  (The kernel is built without packed fp32 instructions: `v_pk_*_f32` with
 *               the op_sel bit of src1 set is unreliable beside MFMA kernels of another stream)
 *               4: the lane-owns-voxel kernel (csrc/warp_cost_lv.hip; variance costs; same stored bits): 14 % fewer vector-ALU
 *               instructions and 2 % less time than the default on narrow-baseline rigs, ~2x slower where boxes do not fit the
 *               LDS arena (wide baselines) -- an alternative, not the default.
 *   "warp_tile" test / measurement aids (0 = off).  LDS-staged kernel: 1 = no adaptive split (a 32-plane chunk whose source boxes do not
 *               fit the LDS arena is normally swept as two 16-plane halves with their own boxes instead of taking global taps).
 *               Lane-owns-voxel kernel: 2 = the same split (off by default there: slower on narrow baselines); 7 = every block on its general path; ablations: 8 = no stores, 9 = no taps, 10 = neither
 *   "warp_gc_lds" 1 (default): group-wise correlation volumes (C = 32, 16-bit, HOMOG geometry, maps of >= 21 x 21 texels) over PER-BATCH
 *               planes run the LDS-staged kernel (csrc/warp_gc_lv.hip; 1.8x the quad kernel on the stage-1 shape of BASELINE
 *               configuration 5; values equal to one 16-bit ulp); 2: per-pixel planes too (slower when the per-pixel depths of a tile
 *               spread widely: boxes that do not fit the LDS take slow global taps); 0: always the quad kernel
 *   "warp_lds_pad" KiB of LDS the LDS-staged warp kernel requests on top of its need (0 = default): fewer workgroups per CU with
 *               the same code (occupancy / stream co-residency experiments)
 *   "sweep_dc"  depth planes per workgroup of the depth-sweep convs (0 = default heuristic)
 *   "sweepc_slots" resident-workgroup target that sizes the depth chunks of the 8|16 -> 8 depth-sweep conv (0 = 768)
 *   "sweepc_pd" prefetch distance in iterations (1..3) of the same kernel (0 = 1)
 *   "sweep_th16" 1: the 32->8 depth-sweep conv uses 16-row tiles / 512 threads; 0 (default): 8-row tiles / 256 threads
 *   "sweep_kdm" 1: the 32->8 depth-sweep conv runs the kd-in-rows formulation (32x32x16 MFMAs whose rows hold the three depth
 *               taps; one input plane per iteration, 3-slot ring, three workgroups per CU) with depth chunks sized for 768 resident
 *               workgroups; 2: sized for 1024; 0 (default): the plane-pair kernel.  Same products, another fp32 summation order
 *               (1e-4 apart before the 16-bit store).  "sweep_kdm_pd": planes in flight per workgroup of that kernel, 1 (default) | 2
 *   "warp_bwd_direct" 1: pscv_warp_cost_bwd issues one global float atomic per tap; 0 (default): accumulates per-workgroup
 *               LDS patches and flushes them coalesced
 *   "c1_sweep"  1 (default): 1-channel heads with 8 input channels and at least three 6-plane blocks per depth chunk run the
 *               depth-sweep variant; 0: always the brick variant; 2: the sweep at any depth
 *   "conv_s2_sweep"  1 (default): stride-2 layers with 8 input channels and <= 32 output channels on volumes of >= 64 Ki output
 *               voxels run the stride-2 depth-sweep kernel; 0: always the brick kernel; 2: the sweep at any size (same packed weights, same result up to
 *               fp32 summation order).  "s2s_slots": resident-workgroup target that sizes its depth chunks (0 = 768)
 *   "conv2d_wlds"  1 (default): 64-channel k3 s1 2-D layers with 32 | 64 output channels and >= 512 tiles run the persistent
 *               kernel that keeps the layer's packed weights in LDS; 0: always conv2d_kernel; 2: at any size (same bits)
 *   "conv_wide"  1 (default): stride-1 3-D layers with 64 input and 32 | 64 output channels on volumes of >= 512 tiles (4 x 4 x 16
 *               voxels) run the persistent 8-wave kernel with the weights through a shared LDS double buffer (csrc/conv3d_wide.hip;
 *               same bits as the brick kernel); 0: always the brick kernel; 2: at any size, and the 32-input layers too
 *   "conv_tall64"  1 (default): on the brick kernel, 64-input stride-1 layers take 4 x 8 x 16 tiles where the grid fills the chip;
 *               0: 4 x 4 x 16; 2: at any size
 *   "tail_nbk"  6-plane blocks per depth chunk of pscv_tail_sweep (0 = default: one resident round of workgroups)
 *   "conv_small_tiles"  1 (default): small volumes use 1x4x16 tiles with the output channels split over
 *               blockIdx.y; 0: always the large-tile variant */
int pscv_set_tuning(const char* key, int value);

/* The same knobs for the CALLING host thread only (enable = 1: this thread reads `value` instead of the process-wide one;
 * enable = 0: drop the override).  pscv_set_tuning itself is process-wide: launches issued from other host threads -- PyTorch
 * runs autograd's backward on its own thread, nn.DataParallel runs replicas on worker threads -- see it too. */
int pscv_set_tuning_thread(const char* key, int value, int enable);
/* The value the CALLING thread's next launch would use (its override if set, else the process-wide value). */
int pscv_get_tuning(const char* key, int* value);

/*
 * Camera blocks of the PROJ geometry in one launch: for every source view v != reference_frame,
 * rot|trans of P_v * P_ref^-1 (fp64 inside).  Replaces the torch.inverse / matmul pair of
 * models/MVSNet/module.py:128-130 (models/CVP_MVSNet/models/modules.py:89-98).
 *   proj  device fp32 [B,V,4,4] with last rows (0,0,0,1);  cams  device fp32 [V-1][B][PSCV_CAM_FLOATS],
 *   sources in view order with the reference view skipped.
 */
int pscv_proj_cams(const float* proj, int B, int V, int reference_frame, float* cams, void* stream);

/*
 * Camera blocks of the HOMOG geometry (Vis-MVSNet) in one launch: A and Bm of hom(d) = A p - Bm p / (d + 1e-9).
 * Replaces scale_camera + the matrix products of get_homographies (models/VisMVSNet/preproc.py:63-92,
 * models/VisMVSNet/homography.py:23-74; the per-plane 3x3 products disappear into the warp kernel).
 *   ref_cam  device fp32 [B,2,4,4] ([R|t], [K ; .]) as built by Frontend.fill_cam_array (frontend.py:14-24)
 *   src_cams device fp32 [n_src][B,2,4,4];  scale = 1 / s_scale;  cams device fp32 [n_src][B][PSCV_CAM_FLOATS]
 */
int pscv_homog_cams(const float* ref_cam, const float* src_cams, int B, int n_src, float scale, float* cams, void* stream);

/*
 * Fused plane-sweep warp + cost aggregation (one pass, the warped per-view volumes never reach HBM).
 * Replaces: MVSNet.build_cost_volume (models/MVSNet/model.py:109-176) + homo_warping (module.py:111-169);
 *           CVP net.py:129-152 and proj_cost (modules.py:229-293); Vis SingleStage.build_cost_volume +
 *           groupwise_correlation (model_cas.py:176-186,340).
 *
 *   ref     [B,h,w,C]                 reference feature map (unused for WARP_ONLY, may be NULL)
 *   srcs    host array of n_src device pointers, each [B,hs,ws,C]
 *   cams    device [n_src][B][PSCV_CAM_FLOATS] fp32
 *   depth   device fp32; per-batch planes: element (b,d) at depth[b*depth_bstride + d];
 *           per-pixel planes (depth_per_pixel=1): element (b,d,y,x) at depth[b*depth_bstride + (d*h + y)*w + x]
 *   out     VARIANCE*, SOFTMIN: [B,D,h,w,C];  GROUPCORR: [n_src][B,D,h,w,C/4];  WARP_ONLY: [n_src][B,D,h,w,C]
 *   C must be a multiple of 8 with C/8 in {1,2,4,8}; n_src <= PSCV_MAX_SRC.
 */
int pscv_warp_cost(const void* ref, const void* const* srcs, int n_src, const float* cams, const float* depth,
                   long depth_bstride, int depth_per_pixel, int geom, int cost, float temp, void* out, int B, int C,
                   int h, int w, int hs, int ws, int D, int in_dtype, int out_dtype, void* stream);

/*
 * pscv_warp_cost on a ROW SLAB of the reference image (ABI 7; the row-sharded Vis-MVSNet stages of a multi-GPU forward,
 * wild_deep_mvs_amd/models/VisMVSNet/model_cas.py forward_row_shard): `ref`, a per-pixel `depth` and `out` hold rows
 * [ref_y0, ref_y0 + h) of the full reference grid, the cameras are those of the WHOLE image, and every reference pixel is
 * evaluated at the coordinates (x, y + ref_y0) it has there -- the same fp32 operations on the same values as the
 * whole-image launch, so the slab's voxels are bit-identical to the corresponding rows of pscv_warp_cost's output.
 * ref_y0 = 0 is pscv_warp_cost.
 */
int pscv_warp_cost_rows(const void* ref, const void* const* srcs, int n_src, const float* cams, const float* depth,
                        long depth_bstride, int depth_per_pixel, int geom, int cost, float temp, void* out, int B, int C,
                        int h, int w, int hs, int ws, int D, int in_dtype, int out_dtype, int ref_y0, void* stream);

/*
 * Second half of a source-view-sharded variance cost volume (MVSNet / CVP-MVSNet): after the all-reduce of the
 * PSCV_COST_VARIANCE_PARTIAL outputs, out = sum2 / N - (sum / N)^2 in the rounding order of `cost`
 * (PSCV_COST_VARIANCE: models/MVSNet/model.py:134, PSCV_COST_VARIANCE_CVP: models/CVP_MVSNet/models/net.py:148).
 *   sums fp32 [2][n] (n = B*D*h*w*C, a multiple of 8), n_views = total number of views incl. the reference, out [n] in `dtype`.
 */
int pscv_variance_finish(const float* sums, long n, int n_views, int cost, int dtype, void* out, void* stream);

/*
 * Weight packing for pscv_conv3d (host side, done once at model-load time).
 * Replaces nothing in the reference; it is the layout contract between a checkpoint's
 * [C_out,C_in,3,3,3] (Conv3d) / [C_in,C_out,3,3,3] (ConvTranspose3d) fp32 tensors and the MFMA kernel.
 *   kind        PSCV_CONV_*
 *   transposed  1 if `w` is a ConvTranspose3d weight ([C_in,C_out,3,3,3]); required for T2, and with S1 it
 *               packs the spatially flipped kernel (ConvTranspose3d k3 s1 p1 == Conv3d with flipped taps)
 *   dtype       PSCV_BF16 or PSCV_F16: the 16-bit format of the packed weights (= the layer's activation format)
 *   returns the number of 16-bit elements of the packed buffer (call with packed == NULL to query), <0 on error.
 */
long pscv_pack_conv3d_weights(const float* w /*host*/, int c_in, int c_out, int kind, int transposed, int dtype,
                              uint16_t* packed /*host, may be NULL*/);
/* The same packing as one launch on device pointers (fp32 weights in, packed 16-bit out; the element count comes from
 * pscv_pack_conv3d_weights(NULL ...)): no device -> host copy when the weights change every optimizer step. */
int pscv_pack_conv3d_weights_device(const float* w /*device*/, int c_in, int c_out, int kind, int transposed, int dtype,
                                    uint16_t* packed /*device*/, void* stream);

/*
 * 3x3x3 convolution as an MFMA implicit GEMM with fused per-channel affine (folded eval-mode BatchNorm or bias),
 * ReLU and skip-add epilogue.
 * Replaces: ConvBnReLU3D / ConvBn3D (models/MVSNet/module.py:41-58), the Sequential(ConvTranspose3d, BatchNorm3d,
 *           ReLU) blocks and `prob` of CostRegNet (models/MVSNet/model.py:43-84; CVP net.py:50-85) and the conv /
 *           deconv members of the Vis UNet (models/VisMVSNet/nn_utils.py:194-278).
 *
 *   dtype    PSCV_BF16 or PSCV_F16: format of `in`, `skip` and `packed` (one MFMA operand type per launch)
 *   in       [B,Di,Hi,Wi,in_cstride], the layer reads channels [in_coff, in_coff + c_in)
 *   packed   device copy of pscv_pack_conv3d_weights output
 *   scale,bias  device fp32 [c_out]:  y = acc*scale + bias   (scale may be NULL = 1, bias may be NULL = 0)
 *   floor    device fp32 [c_out] or NULL: with PSCV_EPI_RELU_PRE, y = max(y, floor[c]) instead of max(y, 0)
 *            (floor = -inf keeps a channel linear; lets one launch carry ReLU and linear channels)
 *   skip     NULL or [B,Do,Ho,Wo,skip_cstride] read at channel offset skip_coff, added after RELU_PRE
 *   out      [B,Do,Ho,Wo,out_cstride] written at channel offset out_coff; out_dtype = dtype or PSCV_F32
 *   (Do,Ho,Wo) = (Di,Hi,Wi) for S1, ceil(./2) for S2, 2x for T2.
 *   c_in in {8,16,32,64}; c_out in {1,8,16,32,64}.
 */
int pscv_conv3d(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                const float* bias, const float* floor, const void* skip, int skip_cstride, int skip_coff, void* out,
                int out_cstride, int out_coff, int out_dtype, int B, int Di, int Hi, int Wi, int c_in, int c_out,
                int kind, int epi_flags, void* stream);

/*
 * The same stride-1 3x3x3 convolution for a 16-channel input that is the CONCATENATION of two 8-channel slices living in different
 * tensors: channels 0..7 = in_a[..., a_coff : a_coff + 8], channels 8..15 = in_b[..., b_coff : b_coff + 8] -- the
 * `torch.cat([deconv_out, enc], dim=1)` in front of the Vis U-Net's decoder conv (models/VisMVSNet/nn_utils.py:269-272) is never
 * written: its two producers store dense 8-channel volumes (a strided half-voxel store into a 16-channel buffer costs the 8 -> 8
 * sweep 119 us instead of 65 us at 256 x 144 x 200) and this launch gathers them while staging.  `packed` is the PSCV_CONV_S1P8
 * packing of the [c_out, 16, 3, 3, 3] weight; c_out 8 or 16; everything else as pscv_conv3d.
 */
int pscv_conv3d_cat2(const void* in_a, int a_cstride, int a_coff, const void* in_b, int b_cstride, int b_coff, int dtype,
                     const uint16_t* packed, const float* scale, const float* bias, const float* floor, const void* skip,
                     int skip_cstride, int skip_coff, void* out, int out_cstride, int out_coff, int out_dtype, int B, int D, int H,
                     int W, int c_out, int epi_flags, void* stream);

/*
 * Input side of the 2-D extractors: fp32 NCHW images -> [B,H,W,8] 16-bit channels-last pixels (channels C..7 zero: the first
 * layer's padded input), and one level of CVP's image pyramid (FeaturePyramid.forward, models/CVP_MVSNet/models/net.py:34-47:
 * F.interpolate(img, scale_factor=0.5, mode='bilinear') -- with align_corners = False the 2 x 2 mean, computed in ATen's operation
 * order: same bits).  One pass over the image writes any of
 *   out_cl8  [B,H,W,8] 16-bit        the image itself in the extractor layout            (may be null)
 *   half_img [B,C,H/2,W/2] fp32      the half-resolution image (input of the next level)  (may be null)
 *   half_cl8 [B,H/2,W/2,8] 16-bit    the half-resolution image in the extractor layout    (may be null)
 * img device fp32 [B,C,H,W] contiguous, C <= 8; the half-resolution outputs need an even W.
 */
int pscv_image_prep(const float* img, int B, int C, int H, int W, int dtype, void* out_cl8, float* half_img, void* half_cl8, void* stream);

/*
 * Vis-MVSNet's UncertNet (models/VisMVSNet/model_cas.py:77-98) in eval mode, one launch for the entropy maps of all pairs:
 *   t1 = relu(s1 * conv3x3(x; w1) + b1)   1 -> 8      (BatchNorm folded: s = gamma / sqrt(var + eps), b = beta - mean * s)
 *   t2 = relu(s2 * conv3x3(t1; w2) + b2) + x          (x broadcast over the 8 channels, model_cas.py:95)
 *   out = conv3x3(t2; head)               8 -> 1      (head_convs[0]; the reference's live model has one head)
 * every convolution zero-pads its own input (padding 1), fp32 throughout like the reference.
 *   entropy [N, H, W] fp32 -> out [N, H, W] fp32 (the log-uncertainty that weights a pair in pscv_fuse_pairs).
 *   params  752 floats: w1 as [tap][c_out] (72) | s1 (8) | b1 (8) | w2 as [c_in][tap][c_out] (576) | s2 (8) | b2 (8) |
 *           head as [c_in][tap] (72); tap = 3 * ky + kx.
 */
int pscv_uncert_net(const float* entropy, const float* params, float* out, int N, int H, int W, void* stream);

/*
 * A residual block of two 8 -> 8 stride-1 3x3x3 convolutions in ONE depth sweep (Vis-MVSNet's 3-D U-Net: BasicBlock, models/
 * VisMVSNet/nn_utils.py:27-37 -- conv1 + bn1 + relu, conv2 + bn2, + identity, relu):
 *   t   = epi1(scale1 * conv(in; packed1) + bias1)                       kept in LDS as 16-bit planes, never written
 *   out = epi2(scale2 * conv(t; packed2) + bias2 [+ in if residual])
 * Same values as pscv_conv3d(layer 1) followed by pscv_conv3d(layer 2, skip = in): t is rounded to the storage format exactly
 * as the stored volume was, zero outside the volume (layer 2's padding).  packed1 / packed2: PSCV_CONV_S1P8 packing of
 * [8, 8, 3, 3, 3] weights; scale / bias / floor [8] or null; epi flags as pscv_conv3d; out 16-bit (the storage dtype), must not
 * alias in.
 */
int pscv_conv3d_block8(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed1, const float* scale1,
                       const float* bias1, const float* floor1, int epi1, const uint16_t* packed2, const float* scale2,
                       const float* bias2, const float* floor2, int epi2, int residual, void* out, int out_cstride, int out_coff,
                       int B, int D, int H, int W, void* stream);

/*
 * Visibility-weighted fusion of per-pair volumes (Vis-MVSNet, mode 'soft'), one pass:
 *     out = sum_v exp(-uncert_v) * interm_v / sum_v exp(-uncert_v)
 * Replaces the accumulate / divide sequence of models/VisMVSNet/model_cas.py:354-357,385-386.
 *   interm   host array of n_src device pointers, each [B,D,h,w,8] in `dtype` (bf16 / fp16)
 *   uncert   host array of n_src device pointers, each [B,h,w] fp32 (the UncertNet head output)
 *   out      normalise = 1: [B,D,h,w,8] in `dtype`;  normalise = 0: fp32 partial sums [B,D,h,w,8] (for a
 *            source-view shard: all-reduce `out` and `wsum_out` across ranks, then divide)
 *   wsum_out [B,h,w] fp32 sum of the weights, or NULL
 */
int pscv_fuse_pairs(const void* const* interm, const float* const* uncert, int n_src, int dtype, void* out,
                    float* wsum_out, int normalise, int B, int D, int h, int w, void* stream);

/*
 * Second half of a source-view-sharded fusion: out = partial / wsum after the partial sums of all ranks were
 * all-reduced (RCCL).  partial fp32 [B,D,h,w,8], wsum fp32 [B,h,w], out [B,D,h,w,8] in `dtype`.
 */
int pscv_fuse_finish(const float* partial, const float* wsum, int dtype, void* out, int B, int D, int h, int w, void* stream);

/*
 * 2-D convolution over channels-last 16-bit feature maps with a fused per-channel affine (folded BatchNorm) and ReLU:
 * the 2-D feature extractor in front of the path.  Replaces ConvBnReLU / Conv2d of
 * models/MVSNet/module.py:23-38, models/MVSNet/model.py:21-41.  Supported layers: k3 s1 p1 and k5 s2 p2,
 * c_in 8 / 16 / 32 / 64 (a 3-channel image is zero-padded to 8 by the caller), c_out 8 / 16 / 32 / 64.
 *   pscv_pack_conv2d_weights: w fp32 [c_out, c_in, k, k] (PyTorch layout) -> MFMA A-fragment order for a layer whose
 *       input has c_in_padded channels; returns the number of 16-bit elements (packed may be NULL to query).
 *   pscv_conv2d: in [B,Hi,Wi,c_in] (c_in = the padded count), out [B,Ho,Wo,c_out] in `out_dtype` (the storage dtype or
 *       fp32); scale / bias fp32 [c_out] or NULL; v = conv * scale + bias, y = max(v, neg_slope * v): neg_slope 0 = ReLU,
 *       0.1 = LeakyReLU(0.1) (models/CVP_MVSNet/models/modules.py:24-28), 1 = no activation.
 */
long pscv_pack_conv2d_weights(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, uint16_t* packed);
/* The same packing with `w` (fp32 [c_out][c_in][ks][ks]) and `packed` on the device, one launch on `stream`, no host round trip
 * (ABI 6): what a training step uses to rebuild the extractor's forward and adjoint layers after an optimiser step. */
int pscv_pack_conv2d_weights_device(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, uint16_t* packed, void* stream);
/* adjoint = 1: `w` is the FORWARD layer's weight [c_in][c_out][ks][ks] (this layer's input channels are its output channels) and the
 * packed layer is its adjoint -- channel axes swapped, taps flipped: the data gradient of a stride-1 conv runs on the forward kernel. */
int pscv_pack_conv2d_weights_device_ex(const float* w, int c_in, int c_in_padded, int c_out, int ks, int dtype, int adjoint, uint16_t* packed,
                                       void* stream);
int pscv_conv2d(const void* in, int dtype, const uint16_t* packed, const float* scale, const float* bias, void* out,
                int out_dtype, int B, int Hi, int Wi, int c_in, int c_out, int ks, int stride, float neg_slope, void* stream);
/*
 * The general form, used by Vis-MVSNet's FeatExt (models/VisMVSNet/model_cas.py:18-35: a 2-D residual U-Net over
 * nn_utils.py:123-278).  On top of pscv_conv2d:
 *   layers   k3 s1|s2 p1, k5 s2 p2, k1 s1|s2 p0 (the BasicBlock shortcuts), and ks = 2: one of the four 2x2-tap parity
 *            sub-convolutions of ConvTranspose2d(k3, s2, p1, op1) -- parity = 2 * (output row parity) + (output column parity),
 *            weights [c_out, c_in, 2, 2] with tap (ty, tx) applied to input (i + ty, j + tx); output pixel (2 i + row parity,
 *            2 j + column parity) of a [B, 2 Hi, 2 Wi, *] map.  A k3 s1 layer may carry a parity too (the sub-convolutions of
 *            ConvTranspose2d(k5, s2, p2, op1) = the data gradient of a k5 s2 conv: taps at -1, 0, +1 around input (i, j)).
 *            parity = -1 for every other layer.
 *   skip     NULL or [B,Ho,Wo,skip_cstride] read at channel offset skip_coff, added BEFORE the activation
 *            (BasicBlock: relu(bn(conv(x)) + shortcut))
 *   out      [B,Ho,Wo,out_cstride] written at channel offset out_coff (the decoder's cat([deconv, skip]) needs no copy)
 *   c_in up to 128; c_out up to 64, or 128 (output tiles split over blockIdx.y).
 */
int pscv_conv2d_ex(const void* in, int dtype, const uint16_t* packed, const float* scale, const float* bias, const void* skip,
                   int skip_cstride, int skip_coff, void* out, int out_cstride, int out_coff, int out_dtype, int B, int Hi, int Wi,
                   int c_in, int c_out, int ks, int stride, int parity, float neg_slope, void* stream);

/*
 * Geometric-consistency filter of one depth map against its source views (the step after the
 * path).  Replaces the body of evaluation/filtering.py:60-83 (unproject -> project_all -> grid_sample of the source
 * depth -> unproj_all -> project -> reprojection / relative-depth / triangulation-angle tests -> per-pixel vote).
 *   depth      device fp32 [h,w]            reference depth map
 *   src_depth  host array of n_src device pointers, fp32 [src_hw[2i], src_hw[2i+1]] (maps may differ in size)
 *   src_hw     host int array [n_src][2] = (height, width) of each source map
 *   cams       device fp32 [n_src+1][PSCV_GEO_CAM_FLOATS]: K, K^-1, R (row-major 3x3 each), t (3); view 0 = reference;
 *              intrinsics at the resolution of the respective depth map
 *   max_reproj_error (pixels), depth_threshold (relative), min_tri_angle (degrees), num_consistent: the reference's
 *              command-line parameters (pipeline_utils.py:49-52); a pixel passes a test when at least
 *              num_consistent - 1 sources agree
 *   mask_depth, mask_disp, geo_mask   device uint8 [h,w] (0/1), each may be NULL
 *   counts     device int32 [3][h,w] number of agreeing sources per test (depth, disp, geo), or NULL
 */
#define PSCV_GEO_MAX_SRC 32
#define PSCV_GEO_CAM_FLOATS 30
int pscv_geo_filter(const float* depth, const float* const* src_depth, const int* src_hw, int n_src, const float* cams,
                    int h, int w, float max_reproj_error, float depth_threshold, float min_tri_angle, int num_consistent,
                    unsigned char* mask_depth, unsigned char* mask_disp, unsigned char* geo_mask, int* counts, void* stream);

/*
 * Softmax over the depth axis + expectation(s), fused.
 * Replaces: F.softmax + depth_regression + photometric confidence (models/MVSNet/model.py:207-215, module.py:174-178;
 *           CVP net.py:161-162,203-219) and soft_argmin / entropy (models/VisMVSNet/nn_utils.py:453-470).
 *
 *   logits        [B,D,h,w] fp32, bf16 or fp16 (logit_dtype)
 *   depth         same addressing as pscv_warp_cost (per-batch or per-pixel planes); may be NULL (then out_depth must be NULL)
 *   out_depth     [B,h,w] fp32  sum_d p_d depth_d                         (NULL to skip)
 *   out_index     [B,h,w] fp32  sum_d p_d d                               (NULL to skip)
 *   out_conf      [B,h,w] fp32  window probability (NULL to skip):
 *                   conf_mode 0: p[i-1]+p[i]+p[i+1]+p[i+2], i = trunc(E[index])   (MVSNet / CVP)
 *                   conf_mode 1: sum of p_d with |d - E[index]| <= window         (Vis soft_argmin(window=))
 *   out_entropy   [B,h,w] fp32  -sum p log(clamp(p,1e-9,1))               (NULL to skip)
 *   out_prob      [B,D,h,w] fp32 full probability volume                  (NULL to skip)
 *   out_partials  [B,4,h,w] fp32 (max, sum exp, sum exp*depth, sum exp*index) of THIS depth shard, for the
 *                 cross-GPU log-sum-exp merge (NULL to skip); index counts from index_offset.
 */
int pscv_softargmin(const void* logits, int logit_dtype, const float* depth, long depth_bstride, int depth_per_pixel,
                    float* out_depth, float* out_index, float* out_conf, float* out_entropy, float* out_prob,
                    float* out_partials, int conf_mode, float window, int index_offset, int B, int D, int h, int w,
                    void* stream);
/*
 * Depth-plane shard across GPUs, last step of the window probability (Vis soft_argmin(window=), nn_utils.py:464-465): this
 * shard's part of sum_{|d - E[index]| <= window} p_d under the GLOBALLY merged softmax.  logits fp32 [B,D,h,w] = the shard's
 * planes (global index = d + index_offset), stats fp32 [B,3,h,w] = (global max, global sum of exp, global expected index) from
 * the merged out_partials; out fp32 [B,h,w]; the shards' outputs are summed by one all-reduce.
 */
int pscv_softargmin_window(const float* logits, const float* stats, float* out, float window, int index_offset, int B, int D,
                           int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training path: what loss.backward() needs from the hot path when train.py drives the model
 * (train.py:185-191, models/trainer.py:96-206).  The reference gets all of it from ATen autograd; here each piece
 * is one entry point.  Forward in train() mode = pscv_warp_cost, pscv_conv3d without folded statistics
 * (scale = bias = NULL, no ReLU), pscv_bn_stats, pscv_bn_act, pscv_softargmin.
 * ------------------------------------------------------------------------------------------------------------------ */

/* number of floats of the `workspace` argument of pscv_bn_stats / pscv_bn_bwd_reduce */
long pscv_train_workspace_floats(void);

/*
 * Per-channel batch statistics of a channels-last 16-bit volume: sums[0][c] = sum y, sums[1][c] = sum y^2 over all
 * nvox voxels (BatchNorm3d in train(): models/MVSNet/module.py:41-58 normalise with the statistics of the batch).
 * Two-phase, fixed summation order (bit-reproducible).  sums: device fp32 [2][C].  C in {8,16,32,64,128} (128: ABI 6, the Vis-MVSNet extractor's widest layers).
 */
int pscv_bn_stats(const void* y, int dtype, long nvox, int C, float* workspace, float* sums, void* stream);

/*
 * The per-channel bookkeeping between pscv_bn_stats and pscv_bn_act, one launch (nn.BatchNorm3d.train() semantics): from sums
 * [2][C]: mean = s0 / nvox, var = max(s1 / nvox - mean^2, 0), invstd = 1 / sqrt(var + eps); out fp32 [4][C] = (scale = gamma invstd,
 * bias = beta - mean scale, mean, invstd).  gamma / beta may be NULL (1 / 0).  When running_mean / running_var are given they are
 * updated in place with `momentum` and the unbiased variance, and *num_batches_tracked (int64, may be NULL) is incremented.
 */
int pscv_bn_finalize(const float* sums, long nvox, int C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, long long* num_batches_tracked, float* out, void* stream);
/*
 * The bookkeeping between pscv_bn_bwd_reduce and pscv_bn_bwd_apply: from sums [2][C] = (sum dz, sum dz y): out fp32 [5][C] =
 * (ca, cb, cc, d gamma, d beta) with dy = ca dz + cb y + cc the BatchNorm input gradient.
 */
int pscv_bn_bwd_coeffs(const float* sums, const float* mean, const float* invstd, const float* gamma, long nvox, int C, float* out,
                       void* stream);

/* out = [relu](y * scale + bias) + skip   (scale / bias fp32 [C] = the batch-statistics affine; skip may be NULL;
 * `skip + relu(bn(deconv(x)))` of models/MVSNet/model.py:79-81).  relu: 0 none, 1 before the skip add, 2 AFTER it
 * (`relu(bn(conv(x)) + shortcut)` of the Vis BasicBlock, models/VisMVSNet/nn_utils.py:123-171). */
int pscv_bn_act(const void* y, int dtype, long nvox, int C, const float* scale, const float* bias, int relu,
                const void* skip, void* out, void* stream);

/* BatchNorm(+ReLU) backward, pass 1: with z = y * scale + bias and dz = dact * [z > 0] (dz = dact when relu = 0),
 * sums[0][c] = sum dz, sums[1][c] = sum dz * y. */
int pscv_bn_bwd_reduce(const void* dact, const void* y, int dtype, long nvox, int C, const float* scale,
                       const float* bias, int relu, float* workspace, float* sums, void* stream);

/* BatchNorm(+ReLU) backward, pass 2: dy = ca[c] * dz + cb[c] * y + cc[c] (the caller folds gamma, 1/std, the batch
 * mean and the two sums of pass 1 into the three per-channel coefficients). */
int pscv_bn_bwd_apply(const void* dact, const void* y, int dtype, long nvox, int C, const float* scale,
                      const float* bias, int relu, const float* ca, const float* cb, const float* cc, void* dy,
                      void* stream);

/*
 * Grouped forms of the six BatchNorm entry points above (ABI 6): the tensor is `groups` consecutive slices of `nvox` voxels, and
 * every slice has its OWN batch statistics and constants -- the views of a 2-D extractor batch, which the reference normalises
 * one view at a time (models/MVSNet/model.py:101-107: `self.feature(img)` per view in train()), in one launch per pass.
 *   pscv_bn_stats_grouped / pscv_bn_bwd_reduce_grouped   sums fp32 [groups][2][C]
 *   pscv_bn_finalize_grouped    out fp32 [groups][4][C] = (scale, bias, mean, invstd) per group; the running statistics are updated
 *                               once per group IN ORDER (the module saw `groups` forward calls), num_batches_tracked += groups
 *   pscv_bn_bwd_coeffs_grouped  out fp32 [groups][5][C]; mean / invstd of group g at mean + g * stat_stride floats
 *   pscv_bn_act_grouped / _bwd_reduce_grouped / _bwd_apply_grouped   constants of group g at scale + g * param_stride floats
 *                               (ca / cb / cc: + g * coeff_stride); with finalize's layout param_stride = 4 C, coeff_stride = 5 C
 * groups = 1 (strides unused) is exactly the plain form; d gamma / d beta of a shared BatchNorm are the sums over the groups.
 */
int pscv_bn_stats_grouped(const void* y, int dtype, long nvox, int groups, int C, float* workspace, float* sums, void* stream);
int pscv_bn_finalize_grouped(const float* sums, long nvox, int groups, int C, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, long long* num_batches_tracked, float* out,
                             void* stream);
int pscv_bn_act_grouped(const void* y, int dtype, long nvox, int groups, int C, const float* scale, const float* bias,
                        int param_stride, int relu, const void* skip, void* out, void* stream);
int pscv_bn_bwd_reduce_grouped(const void* dact, const void* y, int dtype, long nvox, int groups, int C, const float* scale,
                               const float* bias, int param_stride, int relu, float* workspace, float* sums, void* stream);
int pscv_bn_bwd_coeffs_grouped(const float* sums, const float* mean, const float* invstd, int stat_stride, const float* gamma,
                               long nvox, int groups, int C, float* out, void* stream);
int pscv_bn_bwd_apply_grouped(const void* dact, const void* y, int dtype, long nvox, int groups, int C, const float* scale,
                              const float* bias, int param_stride, int relu, const float* ca, const float* cb, const float* cc,
                              int coeff_stride, void* dy, void* stream);

/*
 * Backward of softmax over D + the regression heads (models/MVSNet/model.py:207-209, module.py:174-178;
 * models/VisMVSNet/nn_utils.py:453-470), p = softmax(logits); every upstream gradient is optional (NULL):
 *   grad_depth    of depth   = sum_d p_d depth_d                       (needs the depth planes, as in pscv_softargmin)
 *   grad_index    of index   = sum_d p_d d
 *   grad_entropy  of entropy = -sum_d p_d log clamp(p_d, 1e-9, 1)
 * logits fp32 [B,D,h,w], gradients fp32 [B,h,w]  ->  dlogits8 [B,D,h,w,8] in `dtype` with the gradient in channel 0
 * and zeros in channels 1-7 (the layout the 8-channel MFMA kernels read: the 1-channel heads' backward runs on them).
 */
int pscv_softargmin_bwd(const float* logits, const float* depth, long depth_bstride, int depth_per_pixel,
                        const float* grad_depth, const float* grad_index, const float* grad_entropy, void* dlogits8,
                        int dtype, int B, int D, int h, int w, void* stream);

/* dpre = dout * [out > 0]: backward of a ReLU applied AFTER a residual add (Vis BasicBlock, nn_utils.py:123-171; the
 * forward is pscv_bn_act with relu = 2).  16-bit channels-last volumes, C a multiple of 8. */
int pscv_relu_bwd(const void* dout, const void* out, int dtype, long nvox, int C, void* dpre, void* stream);
/* The same with a negative-side slope (ABI 6): dpre = dout * (out > 0 ? 1 : slope) -- the backward of LeakyReLU(slope), whose output
 * has the sign of its input (CVP-MVSNet's 2-D pyramid: conv + LeakyReLU(0.1), models/CVP_MVSNet/models/modules.py:24-28); slope = 0 is
 * pscv_relu_bwd. */
int pscv_leaky_relu_bwd(const void* dout, const void* out, int dtype, long nvox, int C, float slope, void* dpre, void* stream);
/* The same pass that also returns sums[0][c] = sum over the voxels of the STORED dpre (fp32 [2][C], row 1 = 0): the bias gradient of a
 * conv + bias + LeakyReLU layer without a second read of dpre.  workspace: pscv_train_workspace_floats() floats.  C in {8,...,128}. */
int pscv_leaky_relu_bwd_sum(const void* dout, const void* out, int dtype, long nvox, int C, float slope, void* dpre, float* workspace,
                            float* sums, void* stream);

/*
 * Backward of pscv_fuse_pairs (normalise = 1): d interm_v = G w_v / W, d uncert_v = -w_v / W sum_{d,c} G (interm_v - fused)
 * (models/VisMVSNet/model_cas.py:354-357,385-386 under autograd).  grad_fused [B,D,h,w,8] and dinterm[v] in `dtype`,
 * duncert[v] fp32 [B,h,w]; host arrays of n_src device pointers.
 */
int pscv_fuse_pairs_bwd(const void* const* interm, const float* const* uncert, int n_src, int dtype, const void* grad_fused,
                        void* const* dinterm, float* const* duncert, int B, int D, int h, int w, void* stream);

/*
 * Weight gradient of a 3x3x3 convolution, an MFMA contraction over voxels:
 *     dw[a][b][t] = sum_{n,o} P[n,o,a] * Q[n, stride*o + t - 1, b]        (t = (tz,ty,tx), zero outside Q)
 *   Conv3d weight [C_out,C_in,27]:           P = grad of the layer output (a = c_out), Q = layer input  (b = c_in)
 *   ConvTranspose3d weight [C_in,C_out,27]:  P = layer input (a = c_in),  Q = grad of the layer output (b = c_out)
 * P [B,Dp,Hp,Wp,p_cstride] read at channel offset p_coff; Q [B,stride*Dp,stride*Hp,stride*Wp,q_cstride] at q_coff;
 * both in `dtype` (bf16 / fp16).  ca, cb multiples of 8 in [8,64].  workspace: device fp32,
 * pscv_conv3d_wgrad_workspace(...) floats.  dw: device fp32 [ca][cb][27]; accumulate != 0 adds to it.
 * Fixed summation order (bit-reproducible).
 */
long pscv_conv3d_wgrad_workspace(int B, int Dp, int Hp, int Wp, int ca, int cb, int stride);
int pscv_conv3d_wgrad(const void* p, int p_cstride, int p_coff, int ca, const void* q, int q_cstride, int q_coff, int cb,
                      int dtype, int B, int Dp, int Hp, int Wp, int stride, float* workspace, float* dw, int accumulate,
                      void* stream);

/*
 * Backward of pscv_warp_cost with respect to the feature maps (the sampling grid carries no gradient in the
 * reference: models/MVSNet/module.py:127).  Arguments as pscv_warp_cost; grad_out has the layout of that call's
 * `out` in grad_dtype (the features' 16-bit format or fp32).  dref fp32 [B,h,w,C] (may be NULL for WARP_ONLY),
 * dsrcs host array of n_src device fp32 [B,hs,ws,C], dtemp device fp32 [1] (SOFTMIN, may be NULL): all are
 * ACCUMULATED into with float atomics, the caller zero-fills them.  C in {16, 32}.
 */
int pscv_warp_cost_bwd(const void* ref, const void* const* srcs, int n_src, const float* cams, const float* depth,
                       long depth_bstride, int depth_per_pixel, int geom, int cost, float temp, const void* grad_out,
                       float* dref, float* const* dsrcs, float* dtemp, int B, int C, int h, int w, int hs, int ws, int D,
                       int in_dtype, int grad_dtype, void* stream);

/*
 * CVP-MVSNet refinement hypotheses in eval mode without a host round trip.  Replaces calDepthHypo,
 * models/CVP_MVSNet/models/modules.py:131-226 (a per-batch Python loop in fp64 with a median): per pixel the depth step
 * that moves its projection into the FIRST source view by one pixel along the epipolar line (fp64), the lower median of
 * |step| over the valid pixels (exact radix select), and the planes depth + k * median, k = -4..3.
 *   depth    device fp32 [B,H,W]    the upsampled depth of the previous level
 *   cams     device fp64 [B][39]:   K_ref^-1 (9), rows 0..2 of E_src E_ref^-1 (12), K_src (9),
 *                                   A = (K_ref R_ref)(K_src R_src)^-1 (9); row-major, first source view
 *   fallback device fp32 [B]        (depth_max - depth_min) / 128, the step used when no pixel is valid
 *   keys     device uint64 workspace [B*H*W + B*1040];  steps  device fp64 out [B];  hypos  device fp32 out [B,8,H,W]
 */
int pscv_cvp_depth_hypos(const float* depth, const double* cams, const float* fallback, unsigned long long* keys,
                         double* steps, float* hypos, int B, int H, int W, void* stream);

/*
 * Fused tail of the MVSNet regulariser as ONE depth sweep (ABI 8): transposed 3x3x3 convolution 16 -> 8 (stride 2, output_padding 1)
 * + folded BatchNorm + ReLU + skip add, then the 1-channel 3x3x3 head on the result, which never leaves the CU -- the 8-channel
 * full-resolution volume of the unfused path is neither written nor read.  Same operation chains as pscv_conv3d(kind T2P8)
 * followed by pscv_conv3d(kind S1C1): the logits are bit-identical to the two launches.
 * Replaces: conv11 (Sequential(ConvTranspose3d(16, 8), BatchNorm3d, ReLU)), `conv0 + conv11(x)` and `prob` of CostRegNet.forward,
 *           models/MVSNet/model.py:67-72,81-82.
 *   in        device 16-bit [B,Di,Hi,Wi,in_cstride], channels [in_coff, in_coff + 16)
 *   packed_up device copy of pscv_pack_conv3d_weights(kind PSCV_CONV_T2P8); up_scale / up_bias / up_floor device fp32 [8] or null
 *   skip      null or device 16-bit [B,2Di,2Hi,2Wi,skip_cstride] read at skip_coff (added after the first ReLU)
 *   packed_head device copy of pscv_pack_conv3d_weights(kind PSCV_CONV_S1C1, c_in 8); hd_scale / hd_bias / hd_floor device fp32 [1] or null
 *   logits    device fp32 out [B,2Di,2Hi,2Wi]
 *   depth     null, or device fp32 planes (row b at depth + b * depth_bstride, 2Di entries): the sweep then also keeps the softmax
 *             statistics of its logits per depth chunk and a merge launch writes out_depth [B,2Hi,2Wi] = sum softmax(logits) x depth
 *             and, if out_conf is not null, the 4-plane photometric confidence (models/MVSNet/model.py:207-215) -- the separate
 *             pscv_softargmin pass disappears; workspace: device fp32, pscv_tail_sweep_workspace(B, Di, Hi, Wi) floats
 * Returns 0, or 1 when the shape is not covered (call the two layers), negative on error.
 */
long pscv_tail_sweep_workspace(int B, int Di, int Hi, int Wi);
int pscv_tail_sweep(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed_up, const float* up_scale,
                    const float* up_bias, const float* up_floor, int up_epi, const void* skip, int skip_cstride, int skip_coff,
                    const uint16_t* packed_head, const float* hd_scale, const float* hd_bias, const float* hd_floor, int hd_epi,
                    float* logits, const float* depth, long depth_bstride, float* workspace, long workspace_floats,
                    float* out_depth, float* out_conf, int B, int Di, int Hi, int Wi, void* stream);

/*
 * Fused tail of the MVSNet regulariser: the 1-channel `prob` head (kind S1C1 packing, 8 input channels, depth-sweep variant) writes
 * the fp32 logits AND per-depth-chunk softmax partials; a merge launch gives depth and photometric confidence.  Replaces
 * CostRegNet.prob + F.softmax + depth_regression + the 4-plane confidence (models/MVSNet/model.py:72,82,207-215) -- the separate
 * pscv_softargmin pass over the 15.7 MB logit volume disappears.  Returns -3 (with pscv_last_error) when the layer / size does not
 * qualify (then call pscv_conv3d and pscv_softargmin).
 *   in      device 16-bit [B,D,H,W,in_cstride] (channel slice [in_coff, in_coff + c_in)), packed = S1C1 weights, scale/bias/floor [1]
 *   depth   device fp32 planes, row b at depth + b * depth_bstride, D entries;  logits device fp32 out [B,D,H,W]
 *   workspace device fp32, pscv_prob_softargmin_workspace(B,D,H,W) floats;  out_depth / out_conf device fp32 [B,H,W] (conf may be null)
 */
long pscv_prob_softargmin_workspace(int B, int D, int H, int W);
int pscv_prob_softargmin(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                         const float* bias, const float* floor, int c_in, int epi_flags, const float* depth, long depth_bstride,
                         float* logits, float* workspace, long workspace_floats, float* out_depth, float* out_conf, int B, int D,
                         int H, int W, void* stream);

/*
 * Fused head of a Vis-MVSNet PAIR branch: RegPair.final_conv (8 -> 1, kind S1C1 packing, depth-sweep variant) + soft_argmin's expected
 * plane index + the entropy of the softmax over depth, in one pass over the pair volume.  Replaces RegPair.forward + soft_argmin +
 * entropy (models/VisMVSNet/model_cas.py:55-59,342-348; nn_utils.py:453-470): the pair branch only needs these two maps, so the fp32
 * score volume need not exist (`logits` may be null: 24 -> 16 B of traffic per voxel and one launch instead of two).
 *   entropy = log Z - sum_d e_d (l_d - max) / Z  with e_d = exp(l_d - max), Z = sum e_d: the reference's -sum p log(clamp(p, 1e-9, 1))
 *   without the clamp (planes with p < 1e-9 differ by < 3e-6 in total).
 * Returns -3 when the layer / size does not qualify (then call pscv_conv3d and pscv_softargmin).  Arguments as pscv_prob_softargmin;
 *   workspace pscv_prob_softargmin_workspace(B,D,H,W) floats (used when the depth axis is split into chunks);
 *   out_index / out_entropy device fp32 [B,H,W].
 */
int pscv_head_index_entropy(const void* in, int dtype, int in_cstride, int in_coff, const uint16_t* packed, const float* scale,
                            const float* bias, const float* floor, int c_in, int epi_flags, float* logits, float* workspace,
                            long workspace_floats, float* out_index, float* out_entropy, int B, int D, int H, int W, void* stream);

/*
 * Function-level homography warp: one 3x3 matrix per batch item or per reference pixel.  Replaces homography_warping +
 * interpolate of models/VisMVSNet/homography.py:84-120 for direct callers (inside the model the homographies are built in the
 * fused sweep and never materialised).  Pixel centres at +0.5, z <= 0 -> zero sample, divisor clamped at 1e-9,
 * normalise -> clamp(+-1.1) -> align_corners=True bilinear, zero padding.
 *   image device fp32 [m,hs,ws,c] (channels last);  H device fp32 [m,9] (per_pixel = 0) or [m,h,w,9] (per_pixel = 1), row-major
 *   out   device fp32 [m,h,w,c]
 */
int pscv_homography_warp(const float* image, const float* H, int per_pixel, float* out, int m, int c, int h, int w, int hs, int ws,
                         void* stream);

/*
 * Adjoint of pscv_homography_warp with respect to the image (ABI 9).  The reference evaluates the sample positions under no_grad
 * (models/VisMVSNet/homography.py:110-118) and differentiates only grid_sample's `input` (:101-102): grad_image accumulates
 * grad_out x the four bilinear weights at the taps the forward read.
 *   grad_out   device fp32 [m,h,w,c];  H as in the forward
 *   grad_image device fp32 [m,hs,ws,c], ZEROED by the caller (the kernel adds with fp32 atomics)
 */
int pscv_homography_warp_bwd(const float* grad_out, const float* H, int per_pixel, float* grad_image, int m, int c, int h, int w, int hs,
                             int ws, void* stream);

/*
 * Every camera block of a CVP-MVSNet forward pass in one launch.  Replaces conditionIntrinsics, the per-level projection
 * stacks of homo_warping / proj_cost and the camera products of calDepthHypo (models/CVP_MVSNet/models/modules.py:31-50,
 * 89-98, 229-293, 131-226), which are hundreds of 3x3 tensor ops per forward.
 *   ref_in   device fp32 [B,3,3], src_in [B,N,3,3]      intrinsics at image resolution
 *   ref_ex   device fp32 [B,4,4], src_ex [B,N,4,4]      extrinsics (last row 0 0 0 1)
 *   level_scale HOST fp32 [L]                           image_height / level_height of each pyramid level (L <= 8)
 *   warp_cams device fp32 out [L][N][B][PSCV_CAM_FLOATS]  the pscv_proj_cams block of each level (input of pscv_warp_cost)
 *   hypo_cams device fp64 out [L][B][39] or null          the `cams` of pscv_cvp_depth_hypos of each level (first source view)
 */
int pscv_cvp_cams(const float* ref_in, const float* src_in, const float* ref_ex, const float* src_ex, const float* level_scale,
                  int B, int N, int L, float* warp_cams, double* hypo_cams, void* stream);

/* ---- unsupervised photometric loss (replaces models/trainer.py:209-278 + utils/ssimLoss.py:27-60) ---------
 *
 * pscv_photo_warp: depth map -> flows -> warped source images, the body of Trainer.get_flow_from_depthmap /
 * photometricloss / masked_photometricloss (trainer.py:209-236, 258-266; utils_3D.py:185-208, 243-272):
 *   P3 = inv_ref (x d, y d, d, 1);  q = proj_src P3;  f = q_xy / max(q_z, 1e-6);  g = 2 f / (size - 1) - 1;
 *   g = -10 where q_z <= 0; clamp to +-10; sample = grid_sample(bilinear, zeros, align_corners=False) (index g -> ((g+1) size - 1)/2,
 *   the reference's (size-1)-normalised flows under align_corners=False, reproduced).
 * src_imgs fp32 [B,S,C,h,w] (may be null with C = 0), depth [B,h,w], inv_ref [B,16] = inverse of the reference view's 4x4
 * projection, proj_src [B,S,16]; optional src_depth [B,S,h,w].  Outputs, each optional: warped [B,S,C,h,w], mask [B,S,h,w]
 * (1.0 where |g| < 1 in both axes), z_src [B,S,h,w] (depth in the source view), flows [B,S,h,w,2], warped_depth [B,S,h,w].
 * pscv_photo_warp_bwd: grad_warped [B,S,C,h,w] -> grad_depth [B,h,w] through the sampling position (no gradient where the
 * z <= 0 assignment or the clamp cuts the reference's graph); no atomics, deterministic.
 */
int pscv_photo_warp(const float* src_imgs, const float* depth, const float* inv_ref, const float* proj_src, const float* src_depth,
                    float* warped, float* mask, float* z_src, float* flows, float* warped_depth, int B, int S, int C, int h, int w,
                    void* stream);
int pscv_photo_warp_bwd(const float* src_imgs, const float* depth, const float* inv_ref, const float* proj_src,
                        const float* grad_warped, float* grad_depth, int B, int S, int C, int h, int w, void* stream);
/*
 * SSIM loss map of utils/ssimLoss.py:27-60: out = 1 - SSIM(img1, img2) per channel, 11x11 Gaussian window (sigma 1.5), zero
 * padding.  img1 fp32 [n1,C,h,w], img2 [n1*rep,C,h,w] (item n of img2 is compared with item n / rep of img1: the reference
 * image against its rep warped sources), out [n1*rep,C,h,w].  pscv_ssim_bwd: grad_out -> grad_img2 (img1 is data);
 * workspace 3 * n1*rep*C*h*w floats.
 */
int pscv_ssim(const float* img1, const float* img2, float* out, int n1, int rep, int C, int h, int w, void* stream);
int pscv_ssim_bwd(const float* img1, const float* img2, const float* grad_out, float* workspace, float* grad_img2, int n1, int rep,
                  int C, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSCV_H */
