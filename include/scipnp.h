/* scipnp.h -- C ABI of the MI355X-native adaptive PnP-ADMM engine for Bayer video snapshot
 * compressive imaging (drop-in for the hot path of xyvirtualgroup/AdaptivePnP_SCI).
 *
 * Rules of the boundary
 *   - plain C: raw DEVICE pointers, ints, floats and a stream handle; no torch / C++ types;
 *   - the library never allocates, frees or synchronises: the caller owns every buffer (the Python
 *     host passes PyTorch-ROCm allocations) and passes its HIP stream; every entry is asynchronous
 *     on that stream and may be captured in a hipGraph;
 *   - every entry returns 0 on success or a negative SCIPNP_E* code; scipnp_last_error() gives the
 *     text (thread-local).  The few HOST functions (weight packing from host arrays, scipnp_host_legacy_normal) say so.
 *   - all arrays are float32 unless stated (c8s tensors and packed split weights are fp16 pairs).  Memory layouts:
 *       "reference layout": what the reference's Python functions hold --
 *           planes (M,N,B,4): quarter-resolution Bayer planes R,G1,G2,B in the LAST dim, frame next;
 *           mosaic (H,W,B), y (H,W), rgb cube (H,W,3,B) with H = 2M, W = 2N;
 *       "plane-major layout": what the engine keeps resident in HBM between iterations --
 *           state  [B][4][M][N]   (one contiguous M x N image per frame and Bayer plane),
 *           meas   [4][M][N]      (y and Phi_sum),
 *           rgb    [B][3][H][W]   (planar full-resolution colour frames),
 *           c8     [n][C/8][h][w][8]  (fp32 activations of the denoiser: 8-channel groups innermost),
 *           c8s    [n][C/8][2][h][w][8]  fp16: the same tensor as an error-compensated pair of planes, v = hi + lo'/2048
 *                  (hi = fp16(v), lo' = fp16((v - hi)*2048)), the operand format of the split-fp16 MFMA convolutions.
 *
 * Each entry cites the reference code it replaces (paths relative to the reference repository).
 */
#ifndef SCIPNP_H
#define SCIPNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* scipnp_stream_t; /* a hipStream_t */

#define SCIPNP_OK 0
#define SCIPNP_EINVAL (-1)   /* bad shape / null pointer / unsupported size  */
#define SCIPNP_EALIGN (-2)   /* pointer not 16-byte aligned                  */
#define SCIPNP_EHIP (-3)     /* HIP runtime error at launch                  */
#define SCIPNP_EWORKSPACE (-4) /* workspace too small                        */

const char* scipnp_version(void);
const char* scipnp_last_error(void);
/* name of the GPU architecture the kernels were compiled for ("gfx950") */
const char* scipnp_arch(void);

/* ---------------------------------------------------------------- reference-layout operators
 * These take exactly the tensors the reference's functions take, so a maintainer can bind them
 * one-to-one.  planes = (M,N,B,4) contiguous, meas = (M,N,4). */

/* y[m,n,ib] = sum_t x[m,n,t,ib]*Phi[m,n,t,ib]      -- utilspy.py:28-33  A_(x, Phi) on all 4 planes */
int scipnp_A(const float* x, const float* Phi, float* y, int M, int N, int B, scipnp_stream_t s);
/* x[m,n,t,ib] = y[m,n,ib]*Phi[m,n,t,ib]            -- utilspy.py:35-44  At_(y, Phi) */
int scipnp_At(const float* y, const float* Phi, float* x, int M, int N, int B, scipnp_stream_t s);
/* Phi_sum = sum_t Phi, zeros replaced by 1          -- dvp_linear_inv_2_stage_ADMM_tensor_online.py:72-75 */
int scipnp_phisum(const float* Phi, float* Phisum, int M, int N, int B, scipnp_stream_t s);
/* mosaic (H,W,B) <-> planes (M,N,B,4)               -- utils/utils_image.py:130-151, dvp...:66-69,170-172 */
int scipnp_bayer_split(const float* mosaic, float* planes, int M, int N, int B, scipnp_stream_t s);
int scipnp_bayer_merge(const float* planes, float* mosaic, int M, int N, int B, scipnp_stream_t s);
/* two-stage ADMM Euclidean projection               -- dvp...:128-140
 *   p = theta - inv_rho*b ;  x = p + Phi*((y - sum_t p*Phi)/(alpha_rho + Phisum))
 * x may alias theta (the reference's first iteration does).  One quad (2x2 pixels x B frames) is one
 * 128-byte line for B = 8; the frame reduction is a wavefront shuffle for B = 1,2,4,8,16 and a per-quad loop for
 * any other B.  1 <= B <= 511 everywhere (A, At, phisum, both projections, the plane-major engine): the sums over
 * frames reproduce PyTorch's CPU summation orders (ATen cascade_sum with its flush after 16 groups of four), restated up
 * to 511 addends (csrc/common.hpp). */
int scipnp_proj_twostage(const float* theta, const float* b, const float* Phi, const float* y,
                         const float* Phisum, float* x, int M, int N, int B,
                         float inv_rho, float alpha_rho, scipnp_stream_t s);
/* one-stage ("GAP form") projection                 -- dvp...:389-391
 *   v = theta + b ;  x = v + lambda*Phi*((y - sum_t v*Phi)/(Phisum + gamma)) */
int scipnp_proj_onestage(const float* theta, const float* b, const float* Phi, const float* y,
                         const float* Phisum, float* x, int M, int N, int B,
                         float lambda, float gamma, scipnp_stream_t s);

/* ---------------------------------------------------------------- layout conversion (entry / exit only) */
/* mosaic (H,W,B) -> state [B][4][M][N] and back     -- replaces the strided copies dvp...:66-83, :312-315 */
int scipnp_mosaic_to_state(const float* mosaic, float* state, int M, int N, int B, scipnp_stream_t s);
int scipnp_state_to_mosaic(const float* state, float* mosaic, int M, int N, int B, scipnp_stream_t s);
/* y (H,W) -> meas [4][M][N] */
int scipnp_y_to_meas(const float* y, float* meas, int M, int N, scipnp_stream_t s);
/* planar rgb [B][3][H][W] -> reference cube (H,W,3,B)   -- the (H,W,3,B) return value dvp...:324 */
int scipnp_rgb_to_cube(const float* rgb, float* cube, int H, int W, int B, scipnp_stream_t s);
/* reference cube (H,W,3,B) -> planar rgb [B][3][H][W] */
int scipnp_cube_to_rgb(const float* cube, float* rgb, int H, int W, int B, scipnp_stream_t s);

/* ---------------------------------------------------------------- plane-major engine kernels */
/* setup: Phisum[4][M][N] = sum_t Phi (0 -> 1); if x0 != NULL: x0 = y*Phi  -- dvp...:72-80 */
int scipnp_pm_setup(const float* Phi, const float* y, float* Phisum, float* x0,
                    int M, int N, int B, scipnp_stream_t s);
/* projection on plane-major state; mode 0 = two-stage (c0 = inv_rho, c1 = alpha_rho),
 * mode 1 = one-stage (c0 = lambda, c1 = gamma).  x may alias theta.        -- dvp...:128-140 / :389-391
 * Up to 32 frames every tensor is read exactly once (frames held in registers); 33..511 frames take two passes. */
int scipnp_pm_project(const float* theta, const float* b, const float* Phi, const float* y,
                      const float* Phisum, float* x, int M, int N, int B, int mode,
                      float c0, float c1, scipnp_stream_t s);

/* Chambolle TV prior on C independent M x N channels (plane-major: channel = contiguous image)
 *   theta = TV(x + coef*b)   [b may be NULL: theta = TV(x)]
 * -- replaces the D2H + skimage.restoration.denoise_tv_chambolle(v, weight, n_iter_max=5,
 *    multichannel=True) + H2D round trip at dvp...:153-160 and :403-407.  eps is skimage's default
 *    2e-4.  Per-channel early stop is evaluated on device from fp64 block partials, in launch order
 *    (deterministic).  stop_iter (int32[C], may be NULL) receives the iteration whose `out` was kept. */
size_t scipnp_tv_workspace_bytes(int M, int N, int C, int n_iter_max);
int scipnp_tv_chambolle(const float* x, const float* b, float coef, float* theta,
                        int M, int N, int C, float weight, float eps, int n_iter_max,
                        void* workspace, size_t workspace_bytes, int32_t* stop_iter,
                        scipnp_stream_t s);
/* the same with the kernel named: 0 = as scipnp_tv_chambolle chooses (planes up to 256 columns and <= 5 iterations: the
 * banded form, 3; otherwise 2 where it fits, else 1); 1 = tiled, one launch of 16 x 256 tiles per iteration; 2 = whole
 * plane, all iterations in ONE launch, a workgroup per channel holding its plane in registers (planes up to 128 x 128);
 * 3 = banded, all iterations in one launch of many workgroups per channel (bands of 16 / 32 rows computed with a 4-row
 * halo) plus one launch for the stop test and the recomputation of channels that stopped early; 4 = the banded CANDIDATE
 * form (n_iter_max >= 2, stop_iter required): one launch stores the `out` of every iteration and every band's partial
 * energy sums -- nothing is recomputed, the bands do not communicate -- followed here by the stop test per channel and the
 * copy of every channel's selected candidate to theta as two small launches (scipnp_admm_tv_iterate with defer_state does
 * both inside its dual-update launch instead).  SCIPNP_EINVAL if the named kernel does not fit the shape.  All five give bit-identical `theta` and stop
 * iterations. */
int scipnp_tv_chambolle_ex(const float* x, const float* b, float coef, float* theta,
                           int M, int N, int C, float weight, float eps, int n_iter_max,
                           void* workspace, size_t workspace_bytes, int32_t* stop_iter,
                           int kernel, scipnp_stream_t s);

/* theta = clip(theta_raw,0,1); b = b + sign*(x - theta).  If sse_part != NULL and orig != NULL,
 * also writes per-block partial sums of (orig - report)^2, report = theta (which=0) or x (which=1),
 * to sse_part[0..nblocks) (double); returns the number of blocks through *nblocks.
 * -- dvp...:265-267 (sign=+1) / :501-503 (sign=-1) and the per-iteration IQA :274-279 / :506-512. */
int scipnp_pm_dual_update(const float* theta_raw, const float* x, float* theta, float* b,
                          const float* orig, double* sse_part, int which, float sign,
                          int M, int N, int B, int* nblocks, scipnp_stream_t s);

/* pre-denoiser fusion: mosaic = x + inv_rho*b (never materialised), Malvar-2004 demosaic with the
 * torch port's reflect-101 border, x_rgb [B][3][H][W] stored, net_in = x_rgb - inv_tau*w emitted
 * (a) as planar rgb_w [B][3][H][W] if rgb_w != NULL and (b) 2x2 pixel-unshuffled + noise-level map
 * into c8 layout [B][2][M][N][8] (channels 0..11 = c*4+dy*2+dx, 12 = sigma, 13..15 = 0) if
 * net_in_c8 != NULL.  w may be NULL (treated as 0).
 * -- dvp...:169-172 (merge), :186-191 + malvar2004.py:169-246 (demosaic), :198 (x_rgb - w/tau),
 *    network_ffdnet.py:54-64 (unshuffle + sigma map). */
int scipnp_pm_pre_denoise(const float* x, const float* b, const float* w, float* x_rgb,
                          float* rgb_w, float* net_in_c8, int M, int N, int B,
                          float inv_rho, float inv_tau, float sigma, scipnp_stream_t s);

/* same, additionally emitting the denoiser input in the split-fp16 c8s layout (net_in_c8s, may be NULL) */
int scipnp_pm_pre_denoise_ex(const float* x, const float* b, const float* w, float* x_rgb, float* rgb_w,
                             float* net_in_c8, void* net_in_c8s, int M, int N, int B, float inv_rho,
                             float inv_tau, float sigma, scipnp_stream_t s);

/* the same fusion with only the MOSAIC handed on to the post-denoiser kernel (round 6): instead of x_rgb ([B][3][H][W], 12 bytes per
 * state element written here and read back by scipnp_pm_post_denoise for w += x_rgb - out) the kernel stores the mosaic x + inv_rho*b
 * it demosaicked, plane-major [B][4][M][N] like x (4 bytes per element), and scipnp_pm_post_denoise_mosaic demosaicks its pixels
 * again from it -- the same operations in the same order, so w receives bit for bit what it receives through x_rgb.  `mosaic` must
 * not alias x or b; at least one of rgb_w / net_in_c8 / net_in_c8s is required.  -- the same reference lines as above. */
int scipnp_pm_pre_denoise_mosaic(const float* x, const float* b, const float* w, float* mosaic, float* rgb_w,
                                 float* net_in_c8, void* net_in_c8s, int M, int N, int B, float inv_rho,
                                 float inv_tau, float sigma, scipnp_stream_t s);

/* closed-form RGB update of the reference's `close_form_demosaic=True` branch for iterations k > 0:
 *   x_rgb = (rho*x3 + b3 + tau*out_prev + w) / (rho*cfa_mask + tau), clipped to [0,1] if clip != 0,
 * x3/b3 = Bayer planes scattered to their CFA sites; then x_rgb - inv_tau*w in the same layouts as above.
 * -- dvp...:175-182 (FFDNet branch, clipped) / :224-230 (FastDVDnet branch, not clipped) */
int scipnp_pm_pre_closed_form(const float* x, const float* b, const float* w, const float* out_prev,
                              float* x_rgb, float* rgb_w, float* net_in_c8, void* net_in_c8s, int M, int N, int B,
                              float rho, float tau, float inv_tau, int clip, float sigma, scipnp_stream_t s);

/* x_rgb already holds the demosaicked frames (deep demosaicking, dvp...:192-194 / :242-244): form x_rgb - inv_tau*w
 * (dvp...:198 / :246) and emit it as planar rgb_w and/or the FFDNet input layouts (c8 with sigma map, c8s). */
int scipnp_pm_pre_rgb(const float* w, float* x_rgb, float* rgb_w, float* net_in_c8, void* net_in_c8s, int M, int N,
                      int B, float inv_tau, float sigma, scipnp_stream_t s);

/* post-denoiser fusion: theta_raw = denoised RGB sampled at the CFA sites, theta = clip(theta_raw),
 * b += x_eff - theta with x_eff = theta_raw when first_iter_alias (the reference's k = 0 tensor
 * aliasing, SURVEY 3.2: x and theta are one tensor, so x is overwritten with theta_raw too) else x;
 * w += x_rgb - out (skipped when w == NULL).  The denoised frames come either as planar rgb
 * [B][3][H][W] (out_rgb) or as the FFDNet tail in c8 layout [B][2][M][N][8] before pixel-shuffle
 * (out_c8; exactly one of the two non-NULL); if out_rgb_store != NULL the pixel-shuffled planar
 * frames are also written there (needed for the returned colour cube).  SSE partials as above.
 * -- dvp...:206-209 / :256-259, :265, :267, :271, :274-279 and network_ffdnet.py:66-68. */
int scipnp_pm_post_denoise(const float* out_rgb, const float* out_c8, float* out_rgb_store,
                           float* x, const float* x_rgb, float* theta, float* b, float* w,
                           const float* orig, double* sse_part, int first_iter_alias,
                           int M, int N, int B, int* nblocks, scipnp_stream_t s);

/* the same with x_rgb recomputed from the mosaic scipnp_pm_pre_denoise_mosaic stored (Malvar-2004 on the 6x6 window of every Bayer
 * quad, reflect-101 border: the pre kernel's own operations); w is required (without a w update nothing needs x_rgb: use
 * scipnp_pm_post_denoise with x_rgb = NULL).  -- dvp...:206-209 / :256-259, :265, :267, :271, :274-279, malvar2004.py:169-246. */
int scipnp_pm_post_denoise_mosaic(const float* out_rgb, const float* out_c8, float* out_rgb_store,
                                  float* x, const float* mosaic, float* theta, float* b, float* w,
                                  const float* orig, double* sse_part, int first_iter_alias,
                                  int M, int N, int B, int* nblocks, scipnp_stream_t s);

/* sum of squared differences, per-block fp64 partials (deterministic two-level reduction; the
 * host adds the nblocks doubles)  -- skimage peak_signal_noise_ratio at dvp...:279,:320 */
int scipnp_sse_partials(const float* a, const float* b, size_t n, double* part, int* nblocks,
                        scipnp_stream_t s);

/* ---------------------------------------------------------------- denoiser: 3x3 convolutions on MFMA
 * in/out in c8 layout [n][C/8][h][w][8]; weights pre-packed by scipnp_pack_conv3x3_weights.
 * fp32 operands, fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 products, fmaf chain).
 * Cin, Cout multiples of 8 (pad with zero channels); zero padding 1.
 * flags: bit0 = ReLU, bit1 = add `residual` (same layout and shape as out) before the activation,
 *        bit2 = stride 2 (out is ((h-1)/2+1) x ((w-1)/2+1)),
 *        bit3 = PixelShuffle(2) folded into the store: out is [n][Cout/32][2h][2w][8] with conv channel
 *               4c+2dy+dx written to pixel (2y+dy, 2x+dx), channel c (needs Cout % 32 == 0),
 *        bit4 = ReLU-backward mask: out = (residual > 0) ? out : 0 with `residual` = the forward activation
 *               (backward-data pass of the online finetune; excludes bit1/bit3),
 *        bit8 = "head layer" tag (same arithmetic, separate kernel symbol for profiling).
 * -- replaces nn.Conv2d(...,3,1,1)+ReLU at models/basicblock.py:61-98 as used by network_ffdnet.py:46-48
 *    and every conv of packages/fastdvdnet/models.py:16-89 (CvBlock, InputCvBlock as a block-diagonal
 *    dense conv, DownBlock stride 2, UpBlock + nn.PixelShuffle(2), OutputCvBlock). */
size_t scipnp_conv3x3_packed_floats(int Cin, int Cout);
/* w: [Cout_real][Cin_real][3][3] (PyTorch OIHW), bias: [Cout_real] or NULL; scale/shift fold an
 * eval-mode BatchNorm (y = conv*scale[co] + shift[co]; NULL = identity).  HOST pointers in, HOST
 * packed buffer out (the caller uploads it once per weight update). */
int scipnp_pack_conv3x3_weights(const float* w, const float* bias, const float* bn_scale,
                                const float* bn_shift, int Cin_real, int Cout_real, int Cin, int Cout,
                                float* packed);
int scipnp_conv3x3_c8(const float* in, const float* packed_w, float* out, const float* residual,
                      int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
/* same with separate pointers for the residual add (bit1) and the ReLU-backward mask source (bit4), so that a
 * skip-connection gradient can be added before masking: out = mask(conv + residual) */
int scipnp_conv3x3_c8_ex(const float* in, const float* packed_w, float* out, const float* residual,
                         const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags,
                         scipnp_stream_t s);

/* ---- error-compensated split-fp16 variant (csrc/conv_split.hip): every fp32 operand v = hi + lo'*2^-11 as two
 * fp16 numbers, three exact fp16 products per fp32 product on v_mfma_f32_32x32x16_f16 (fp32 accumulate).
 * Activations in layout c8s [n][C/8][2][h][w][8] fp16 (hi plane, lo' plane; same bytes as fp32 c8).
 * zero pad 1; flags: bit0 ReLU, bit2 stride 2, bit3 PixelShuffle(2) store (output is then fp32 c8
 * [n][Cout/32][2h][2w][8]), bit5 (32) = write fp32 c8 instead of c8s (network tails), bit8 head tag.
 * Per-iterate error in the PnP loop <= 3.4e-6 (bar 1e-5); needs |w| < 31.9 and activations inside fp16 range. */
size_t scipnp_conv3x3_split_packed_bytes(int Cin, int Cout);
/* HOST pointers: w OIHW fp32, bias or NULL -> packed split weights; _bn folds an eval-mode BatchNorm
 * (w*scale[co], bias*scale + shift) before splitting */
int scipnp_pack_conv3x3_split(const float* w, const float* bias, int Cin_real, int Cout_real, int Cin, int Cout,
                              void* packed);
int scipnp_pack_conv3x3_split_bn(const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
                                 int Cin_real, int Cout_real, int Cin, int Cout, void* packed);
int scipnp_conv3x3_c8s(const void* in_c8s, const void* packed_split, void* out, int n, int Cin, int Cout,
                       int h, int w, int flags, scipnp_stream_t s);
/* Range guard of the split format.  A kernel that writes a c8s tensor (or packs split weights on the device) raises a
 * 4-byte DEVICE word when a value is >= 65000 in magnitude or NaN (|w| >= 31.9 for weights): results are then invalid,
 * rerun with the fp32 kernels.  WHICH word is part of the call, not of the library: every launch passes the word the
 * calling thread bound with scipnp_bind_overflow_word -- a device int the caller owns and zeroes, one per solve / engine
 * (scipnp_twostage_ffdnet_args.overflow_word names it per call) -- so solves that overlap in time on different host
 * threads never see each other's report.  Threads that never bind one share the library's process-wide word.
 *   scipnp_bind_overflow_word(w): w for this thread's following launches (NULL: back to the process-wide word).
 *   scipnp_read_overflow_word(w, reset, flag_out, s): *flag_out = the word (w == NULL: the one this thread has bound /
 *     the process-wide one), zeroed afterwards if reset; SYNCHRONISES the stream -- call once per reconstruction.
 *   scipnp_split_overflow(reset, flag_out, s) = scipnp_read_overflow_word(NULL, ...). */
int scipnp_bind_overflow_word(int* dev_word);
int scipnp_read_overflow_word(const int* dev_word, int reset, int* flag_out, scipnp_stream_t s);
int scipnp_split_overflow(int reset, int* flag_out, scipnp_stream_t s);
/* fp32 c8 -> c8s; _add adds a c8s residual first (skip connections behind a PixelShuffle conv) */
int scipnp_c8_to_c8s(const float* in_c8, void* out_c8s, int n, int C, int h, int w, scipnp_stream_t s);
int scipnp_c8_add_to_c8s(const float* in_c8, const void* residual_c8s, void* out_c8s, int n, int C, int h, int w,
                         scipnp_stream_t s);

/* split-fp16 conv with a ReLU-mask epilogue (flag bit4 = 16): out = conv(in) where mask_c8s (a c8s tensor of the
 * output's shape, the stashed forward activation) is positive, else 0 -- the backward-data convolution of the online
 * finetune on the fp16 MFMA (packages/ffdnet/test_ffdnet_ipol.py:296 `loss.backward()`).  Other flags as
 * scipnp_conv3x3_c8s; mask_c8s may be NULL when bit4 is clear.  Flag bit1 = 2 adds residual_c8s (a c8s tensor of
 * the output's shape: the gradient arriving over a skip connection) before the mask.  Flag bit6 = 64 together with
 * bit3: the PixelShuffle(2)-ed result is stored as c8s [n][Cout/32][2][2h*2w][8] (residual_c8s, if bit1, has that
 * shape: the UpBlock's skip tensor, fastdvdnet models.py:190-193) instead of fp32 c8. 
 * Flags 0x200 / 0x400 / 0x800 select measured-alternative forms of the 96-channel stride-1 kernel for
 * tools/conv_bench.py (weights double-buffered with 2 workgroups per CU / v_mfma_f32_16x16x32_f16 / linear instead of
 * XCD-aware tile order); same results to fp32 re-association. */
int scipnp_conv3x3_c8s_ex(const void* in_c8s, const void* packed_split, void* out, const void* residual_c8s,
                          const void* mask_c8s, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);

/* device-side packing of fp32 master weights (device pointers) into the split layout: transpose_flip = 0 forward
 * (with bias), 1 the backward-data convolution W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] (Cin/Cout are the FORWARD
 * conv's padded channel counts; no bias).  Raises the split-overflow flag if |w| >= 31.9. */
int scipnp_pack_conv3x3_split_device(const float* w, const float* bias, void* packed, int Cin_real, int Cout_real,
                                     int Cin, int Cout, int transpose_flip, scipnp_stream_t s);
/* same with a folded eval-mode BatchNorm: forward W*scale[co] (bias = BN shift), backward-data W*scale[co] as well */
int scipnp_pack_conv3x3_split_device_scaled(const float* w, const float* bias, const float* scale, void* packed,
                                            int Cin_real, int Cout_real, int Cin, int Cout, int transpose_flip,
                                            scipnp_stream_t s);

/* weight / bias gradients from c8s operands on the fp16 MFMA (error-compensated, transposing LDS reads); same
 * workspace and slab scheme as scipnp_conv3x3_wgrad / scipnp_conv_bias_grad.  dz_c8s carries the gradient
 * pre-scaled by a power of two, `scale` (its reciprocal) is applied to the result.
 * -- packages/ffdnet/test_ffdnet_ipol.py:296 (loss.backward()) */
int scipnp_conv3x3_wgrad_split(const void* act_c8s, const void* dz_c8s, float* dW, float* workspace, int nslab, int n,
                               int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, float scale,
                               scipnp_stream_t s);
int scipnp_conv_bias_grad_split(const void* dz_c8s, float* db, float* workspace, int n, int Cout_real, int Cout, int h,
                                int w, float scale, scipnp_stream_t s);

/* c8s -> fp32 c8 and fp32 c8 -> c8s with an exact power-of-two rescale (gradients travel through the split kernels
 * pre-scaled into fp16 range) */
int scipnp_c8s_to_c8(const void* in_c8s, float* out_c8, float scale, int n, int C, int h, int w, scipnp_stream_t s);
int scipnp_c8_scale_to_c8s(const float* in_c8, void* out_c8s, float scale, int n, int C, int h, int w,
                           scipnp_stream_t s);

/* whole FFDNet-colour forward on B frames: 12 (nb) conv layers ping-ponging between two c8 scratch
 * buffers of n*nc*h*w floats each.  in_c8: [B][2][M][N][8] from scipnp_pm_pre_denoise, out_c8:
 * [B][2][M][N][8] (12 valid channels, pixel-shuffle is folded into scipnp_pm_post_denoise).
 * packed: array of nb device pointers to packed layer weights.
 * -- models/network_ffdnet.py:54-69 called per frame by packages/ffdnet/test_ffdnet_ipol.py:340-354 */
int scipnp_ffdnet_forward(const float* in_c8, float* out_c8, const float* const* packed, int nb, int nc,
                          float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s);
/* the same on the split-fp16 kernels (the default precision): in_c8s [B][2][2][M*N][8] fp16 from
 * scipnp_pm_pre_denoise_ex, fp32 c8 output, packed_split from scipnp_pack_conv3x3_split(_device), two c8s scratch
 * buffers of B*nc*M*N*4 bytes each; everything on `s`.
 * _2s: with side_stream != NULL (and B >= 2) the second half of the frames runs on the CALLER's side stream, forked from
 * and joined to `s` inside the call through the caller's two events (hipEvent_t, passed as void*; created with
 * hipEventDisableTiming) -- identical results, the other stream's launches fill the CUs a grid's last generation of
 * workgroups leaves idle; hipGraph capture on `s` records both branches.  The library itself creates no stream or event
 * and reads no environment variable here. */
int scipnp_ffdnet_forward_c8s_2s(const void* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc,
                                 void* scratch0, void* scratch1, int B, int M, int N, scipnp_stream_t s,
                                 scipnp_stream_t side_stream, void* fork_event, void* join_event);
int scipnp_ffdnet_forward_c8s(const void* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc,
                              void* scratch0, void* scratch1, int B, int M, int N, scipnp_stream_t s);

/* ---------------------------------------------------------------- online finetune (measurement loss)
 * -- packages/ffdnet/test_ffdnet_ipol.py:248-300: Adam steps on  MSE( sum_t Phi * bayer_sample(net(x)), y ).
 * Backward-data of a conv layer is scipnp_conv3x3_c8 with weights packed by scipnp_pack_conv3x3_device(...,
 * transpose_flip=1) and flag bit4. */
/* loss partials (sum of squared residuals per block, double; L = sum/(4MN)) and dL/d(net tail output) in c8
 * [B][2][M][N][8]; loss_part == NULL only returns *nblocks.            -- test_ffdnet_ipol.py:275-293 */
int scipnp_ffdnet_loss_grad(const float* out_c8, const float* Phi, const float* y, float* gout_c8,
                            double* loss_part, int M, int N, int B, int* nblocks, scipnp_stream_t s);
/* dW (OIHW, real channel counts) of a 3x3/pad-1/stride-1 conv from its input activations and output gradient;
 * MFMA GEMM over pixels with nslab persistent workgroups; workspace >= ..._workspace_floats floats. */
size_t scipnp_conv3x3_wgrad_workspace_floats(int Cin, int Cout, int nslab);
int scipnp_conv3x3_wgrad(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                         int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s);
/* the same gradient in the Winograd F(2x2,3x3) domain (csrc/wgrad_wino.hip): dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G,
 * 16 exact fp32 products per 2x2 tile and channel pair instead of 36, accumulated in fp32 on v_mfma_f32_32x32x2_f32 with the
 * tiles as the K dimension; equal to scipnp_conv3x3_wgrad up to fp32 re-association (tests/test_gpu_ops.py).  Own workspace
 * size (16 Winograd positions per slab instead of 9 taps).  The fp32 trainers use it unless SCIPNP_F32_CONV=direct. */
size_t scipnp_conv3x3_wgrad_wino_workspace_floats(int Cin, int Cout, int nslab);
int scipnp_conv3x3_wgrad_wino(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                              int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s);
/* the same gradient in the Winograd F(4x4,3x3) domain (csrc/wgrad_wino4.hip): 36 fp32 products per 4x4 tile and channel pair
 * (2.25 per output instead of 4 / 9), transforms with factors up to 8: rel-L2 ~3e-6 against float64 where the F(2x2) form gives
 * ~5e-7 (tests/test_gpu_ops.py gates 1e-5).  One slab = 36 positions x roundup(Cout, 32) x roundup(Cin, 32) floats; nslab x
 * (Cout/32 x Cin/32 blocks) persistent workgroups, one per CU (nslab ~ 255 / blocks), whole slabs placed on one XCD.  382 us
 * against 441 us of the F(2x2) form at 96 -> 96 on 8 x 256 x 256; the fp32 FFDNet trainer uses it unless SCIPNP_F32_WGRAD=f2. */
size_t scipnp_conv3x3_wgrad_wino4_workspace_floats(int Cin, int Cout, int nslab);
int scipnp_conv3x3_wgrad_wino4(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                               int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s);
/* dW == NULL: only the kernel that fills the slabs; the slab reductions and back-transforms of n such layers (each with its own
 * workspace) then take two launches in all -- a trainer finishes every layer at the end of its backward pass */
int scipnp_conv3x3_wgrad_wino4_finish_multi(int n, float* const* workspace, float* const* dW, const int* nslab, const int* Cin_real,
                                            const int* Cout_real, const int* Cin, const int* Cout, scipnp_stream_t s);
/* db[co] = sum dz; workspace >= (Cout/8)*64*8 floats */
int scipnp_conv_bias_grad(const float* dz_c8, float* db, float* workspace, int n, int Cout_real, int Cout,
                          int h, int w, scipnp_stream_t s);
/* db == NULL: the partial sums only (workspace keeps them); the reductions of n layers in one launch */
int scipnp_conv_bias_grad_reduce_multi(int n, const float* const* workspace, float* const* db, const int* Cout_real,
                                       scipnp_stream_t s);
/* one torch.optim.Adam step (amsgrad=False, weight_decay=0) on a flat float32 tensor; step counts from 1 */
int scipnp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                     double beta1, double beta2, double eps, int step, scipnp_stream_t s);
/* device-side weight packing (OIHW device tensor -> packed buffer of scipnp_conv3x3_packed_floats floats);
 * transpose_flip = 1 packs the backward-data convolution (Cout -> Cin channels, taps flipped, no bias). */
int scipnp_pack_conv3x3_device(const float* w, const float* bias, float* packed, int Cin_real, int Cout_real,
                               int Cin, int Cout, int transpose_flip, scipnp_stream_t s);

/* same with a per-output-channel scale folded into the weights (eval-mode BatchNorm: scale = gamma/sqrt(var+eps),
 * `bias` = the folded shift) */
int scipnp_pack_conv3x3_device_scaled(const float* w, const float* bias, const float* scale, float* packed,
                                      int Cin_real, int Cout_real, int Cin, int Cout, int transpose_flip,
                                      scipnp_stream_t s);
/* the same packs for n layers in ONE launch (host arrays of n entries; bias / scale may be NULL or hold NULL entries): a trainer
 * repacks every layer in both directions after each optimiser step, and a launch of its own per pack is a dependent launch of
 * ~4 us each (the reference re-reads the module's parameters every forward: test_ffdnet_ipol.py:289-300) */
int scipnp_pack_conv3x3_device_multi(int n, const float* const* w, const float* const* bias, const float* const* scale,
                                     float* const* packed, const int* Cin_real, const int* Cout_real, const int* Cin,
                                     const int* Cout, const int* transpose_flip, scipnp_stream_t s);
/* ---------------------------------------------------------------- fp32 Winograd F(2x2,3x3) form of the same convolution
 * (stride 1, zero padding 1; csrc/conv_wino.hip): Y = A^T[(G g G^T) (.) (B^T d B)]A with every product an exact fp32
 * product accumulated in fp32 on v_mfma_f32_16x16x4_f32 -- 2.25x fewer multiply-adds than scipnp_conv3x3_c8, results
 * equal to it up to fp32 re-association (<= 2e-6 relative L2 per layer in tests/test_gpu_ops.py).  The default fp32 path
 * of the FFDNet / FastDVDnet / DDnet engines for their stride-1 layers (SCIPNP_F32_CONV=direct selects the direct kernel).
 * packed_wino: scipnp_conv3x3_wino_packed_floats(Cin, Cout) floats, derived ON THE DEVICE from a buffer packed by
 * scipnp_pack_conv3x3_weights / _device(_scaled) (so bias, BatchNorm folding and the transposed backward-data packing
 * carry over); layout [Cin/8][CoutP/32][xi][h][j][lane][nu] -- per channel group and 32-channel block the 16-byte vector
 * a lane reads for (patch row xi, channel half h, k-step j): its four column positions nu, lane = (ci%4)*16 + co%16
 * (pack_wino_kernel, csrc/conv_wino.hip) -- then bias[CoutP].
 * flags: bit0 ReLU, bit1 add `residual`, bit3 PixelShuffle(2) folded into the store (out and residual [n][Cout/32][2h][2w][8],
 *        Cout a multiple of 32, as scipnp_conv3x3_c8_ex; the shuffled tile is assembled in LDS and stored in whole 128-byte lines),
 *        bit4 ReLU-backward mask from `mask_src`, bit8 head-layer tag (as conv3x3_c8_ex).
 * h * w < 2^25 (planes are addressed through 32-bit buffer offsets; SCIPNP_EINVAL beyond -- the engines then use scipnp_conv3x3_c8).
 * -- replaces the same nn.Conv2d(..., 3, 1, 1) call sites as scipnp_conv3x3_c8 (models/basicblock.py:61-98,
 *    models/network_ffdnet.py:46-48, packages/fastdvdnet/models.py:16-89); the reference's fp32 cuDNN path makes the
 *    same algorithmic choice for 3x3 convolutions. */
size_t scipnp_conv3x3_wino_packed_floats(int Cin, int Cout);
int scipnp_pack_conv3x3_wino(const float* packed_f32, float* packed_wino, int Cin, int Cout, scipnp_stream_t s);
int scipnp_conv3x3_c8w(const float* in, const float* packed_wino, float* out, const float* residual,
                       const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);

/* ---- fp32 Winograd F(4x4,3x3) form (csrc/conv_wino4.hip, round 3): 36 exact fp32 products per 16 outputs and channel pair,
 * 2.25 multiply-adds per output (F(2x2): 4, direct: 9), interpolation points 0, +-1, +-2, inf, fp32 accumulation on
 * v_mfma_f32_16x16x4_f32; results equal to scipnp_conv3x3_c8 up to fp32 re-association and the transforms' rounding
 * (<= 4e-6 relative L2 per layer in tests/test_gpu_ops.py; 3e-7 through the 12 FFDNet layers against float64).
 * packed_wino4: scipnp_conv3x3_wino4_packed_floats(Cin, Cout) floats derived on the device from the fp32 direct packing
 * (U = G g G^T in double), layout [2*Cin/8 k-steps][CoutP/32][xi half][9 vectors][lane][4], then bias[CoutP].
 * flags: bit0 ReLU, bit1 add `residual`, bit3 PixelShuffle(2) folded into the store (out and residual [n][Cout/32][2h][2w][8], Cout a
 *        multiple of 32, no mask; as scipnp_conv3x3_c8w), bit4 ReLU-backward mask from `mask_src`, bit8 head-layer tag; stride 1.
 * The engines use it for layers of at least 16 input and 32 output channels (a narrower output is padding in its 32-channel
 * workgroups); SCIPNP_WINO_F4=0 keeps them on scipnp_conv3x3_c8w.
 * -- same nn.Conv2d(..., 3, 1, 1) call sites as scipnp_conv3x3_c8w. */
size_t scipnp_conv3x3_wino4_packed_floats(int Cin, int Cout);
int scipnp_pack_conv3x3_wino4(const float* packed_f32, float* packed_wino4, int Cin, int Cout, scipnp_stream_t s);
/* ... for n layers in ONE launch (host arrays of n entries) */
int scipnp_pack_conv3x3_wino4_multi(int n, const float* const* packed_f32, float* const* packed_wino4, const int* Cin,
                                    const int* Cout, scipnp_stream_t s);
int scipnp_conv3x3_c8w4(const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);

/* The whole FFDNet-colour pass as ONE call with mixed Winograd forms: layer l runs on scipnp_conv3x3_c8w4 when
 * packed_wino4 != NULL and packed_wino4[l] != NULL (the 96 -> 96 body layers, packed by scipnp_pack_conv3x3_wino4), else on
 * scipnp_conv3x3_c8w with packed_wino[l]; other arguments as scipnp_ffdnet_forward_c8w. */
int scipnp_ffdnet_forward_c8w4(const float* in_c8, float* out_c8, const float* const* packed_wino, const float* const* packed_wino4,
                               int nb, int nc, float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s);

/* The whole FFDNet-colour pass (test_ffdnet_ipol.py:340-359 -> network_ffdnet.py:65-66) as ONE call on the Winograd
 * kernel: same arguments as scipnp_ffdnet_forward with every layer packed by scipnp_pack_conv3x3_wino. */
int scipnp_ffdnet_forward_c8w(const float* in_c8, float* out_c8, const float* const* packed_wino, int nb, int nc,
                              float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s);

/* eval-mode BatchNorm after a bias-free conv, y = conv(x;W)*s + t: fold (s, t) and the parameter gradients
 * dW = s*G, dgamma = (<W,G> - mean*sum(dy))/sqrt(var+eps), dbeta = sum(dy), G = wgrad(x, dy), K = Cin*9
 * -- packages/fastdvdnet/models.py:16-89 with BN kept in eval() during finetune (test_fastdvdnet.py:376-379) */
int scipnp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   float* scale, float* shift, int C, scipnp_stream_t s);
int scipnp_bn_fold_grads(const float* W, const float* G, const float* sdy, const float* gamma, const float* mean,
                         const float* var, float eps, float* dW, float* dgamma, float* dbeta, int Cout, int K,
                         scipnp_stream_t s);

/* ---------------------------------------------------------------- FastDVDnet glue
 * DenBlock input from planar frames [B][3][H][W] with circular temporal indexing: out c8 [B][2][H][W][8],
 * entry n = (frame n-1, sigma, frame n, sigma | frame n+1, sigma, 0...), indices mod B.
 * -- packages/fastdvdnet/models.py:187 (torch.cat), packages/fastdvdnet/fastdvdnet.py:113-116 */
int scipnp_fastdvd_pack_triplets(const float* frames, float* out_c8, int B, int H, int W, float sigma,
                                 scipnp_stream_t s);
/* same in the split-fp16 c8s layout */
int scipnp_fastdvd_pack_triplets_c8s(const float* frames, void* out_c8s, int B, int H, int W, float sigma,
                                     scipnp_stream_t s);
/* the same for a unit batch of `units` sequences of B frames each, frame-major with the unit inside (frame t of unit u at index
 * t * units + u, the layout of scipnp_pm_project_units): the window of frame (t, u) is (t-1, u), (t, u), (t+1, u), circular in t
 * within the unit -- the units of a batch never see each other's frames. */
int scipnp_fastdvd_pack_triplets_units(const float* frames, float* out_c8, int B, int units, int H, int W, float sigma,
                                       scipnp_stream_t s);
int scipnp_fastdvd_pack_triplets_c8s_units(const float* frames, void* out_c8s, int B, int units, int H, int W, float sigma,
                                           scipnp_stream_t s);
/* DenBlock residual: out[n][c] = center[n][c] - x_c8[n][0][..][c], c < 3  -- models.py:196 (in1 - x) */
int scipnp_fastdvd_finish(const float* center, const float* x_c8, float* out, int B, int H, int W,
                          scipnp_stream_t s);

/* backward-pass glue of the FastDVDnet online finetune (reference test_fastdvdnet.py:343-451) */
/* zero-insertion upsample (gradient path of a stride-2 conv): out[n][cg][2y][2x] = in[n][cg][y][x] */
int scipnp_upsample_zero_c8(const float* in, float* out, int n, int C, int h, int w, int H, int W,
                            scipnp_stream_t s);
/* PixelShuffle(2) backward: dshuf [n][Cs/8][2h][2w][8] -> dconv [n][4Cs/8][h][w][8] */
int scipnp_pixel_shuffle_bwd_c8(const float* dshuf, float* dconv, int n, int Cs, int h, int w, scipnp_stream_t s);
/* the same two permutations on c8s tensors (applied to the hi and lo' planes alike) */
int scipnp_upsample_zero_c8s(const void* in, void* out, int n, int C, int h, int w, int H, int W, scipnp_stream_t s);
int scipnp_pixel_shuffle_bwd_c8s(const void* dshuf, void* dconv, int n, int Cs, int h, int w, scipnp_stream_t s);
/* measurement loss on planar frames [B][3][H][W] and its gradient (zero off the CFA sites); loss partials as in
 * scipnp_ffdnet_loss_grad                                       -- test_fastdvdnet.py:424-431 */
int scipnp_fastdvd_loss_grad(const float* out, const float* Phi, const float* y, float* dout, double* loss_part,
                             int M, int N, int B, int* nblocks, scipnp_stream_t s);
/* d(center - x)/dx: dx_c8 [B][1][H][W][8] = (-dout.rgb, 0...) */
int scipnp_fastdvd_finish_bwd(const float* dout, float* dx_c8, int B, int H, int W, scipnp_stream_t s);
/* gradient w.r.t. the planar frames behind scipnp_fastdvd_pack_triplets, plus `extra` (may be NULL) */
int scipnp_fastdvd_unpack_bwd(const float* dtin_c8, const float* extra, float* dframes, int B, int H, int W,
                              scipnp_stream_t s);

/* Final-report metrics per frame of the mosaic cube (plane-major states [B][4][M][N]): part[t][blk] = {sum of squared
 * error (fp32 squares, fp64 sum), sum of the SSIM map over the image minus a (win-1)/2 border} -- PSNR_t =
 * 10 log10(range^2 / (sum0 / HW)), SSIM_t = sum1 / ((H-win+1)(W-win+1)).  scikit-image 0.18 semantics (win x win uniform
 * window, sample covariance, K1 = 0.01, K2 = 0.03, fp64).  part == NULL only returns the block count per frame.
 * -- dvp...:316-321 / :542-547 */
int scipnp_frame_metrics(const float* ref_state, const float* img_state, double* part, int M, int N, int B, int win,
                         double data_range, int* nblocks, scipnp_stream_t s);

/* ---------------------------------------------------------------------------------------------------------------
 * Whole ADMM iterations as single calls: the launch sequences of the solver loop for hosts that run it natively
 * (the Python stepper issues the same launches one by one).  All pointers are device pointers in the plane-major
 * layouts above; nothing is allocated or synchronised.
 * ABI guard: the first member of either argument block is `struct_size`, which the caller sets to sizeof(the struct it was
 * compiled against) (memset the block to 0 first); a block of any other size is refused with SCIPNP_EINVAL instead of
 * being misread (the blocks changed layout between library versions 0.1 and 0.2).
 * ------------------------------------------------------------------------------------------------------------- */

/* two-stage ADMM + FFDNet-colour, Malvar demosaic (dvp...:121-271, one pass of the loop body):
 *   x = project(theta, b);  x_rgb = malvar(mosaic(x + b/rho));  out = FFDNet(x_rgb - w/tau, sigma);
 *   theta = clip(CFA(out));  b += x - theta;  w += x_rgb - out;  optional squared-error partials vs orig. */
typedef struct {
    size_t struct_size;                 /* = sizeof(scipnp_twostage_ffdnet_args) */
    int M, N, B;                        /* quarter-resolution plane size (H/2, W/2), frames */
    float *theta, *b, *x;               /* state [B][4][M][N]; theta holds the start point at the first iteration */
    const float *Phi, *y, *Phisum;      /* [B][4][M][N], [4][M][N], [4][M][N] (scipnp_pm_setup) */
    float *w, *x_rgb;                   /* RGB dual and demosaicked frames [B][3][2M][2N] */
    float* out_rgb;                     /* denoised frames [B][3][2M][2N], or NULL when not wanted this iteration */
    void* net_in_c8s;                   /* FFDNet input  [B][2][2][M*N][8] fp16 */
    float* net_out_c8;                  /* FFDNet output [B][2][M][N][8] fp32 */
    const void* const* packed_split;    /* nb packed layers (scipnp_pack_conv3x3_split) */
    int nb, nc;                         /* 12, 96 for ffdnet_color */
    void *scratch0, *scratch1;          /* B*nc*M*N*4 bytes each */
    const float* orig;                  /* ground truth state [B][4][M][N] or NULL */
    double* sse_part;                   /* per-block partial sums for the PSNR, or NULL */
    double rho, alpha, tau;             /* reference constants: rho = 1 (0.55 with FastDVDnet / closed form), alpha = 1,
                                         * tau = 100 (:101-110).  double: the kernels take float(1/rho), float(alpha*rho),
                                         * float(1/tau) rounded ONCE from double, as the reference's Python scalars are
                                         * (1.0f/0.55f is one ulp away from float(1/0.55)) */
    float sigma;
    int first_iter;                     /* 1 on the very first iteration (x and theta are one tensor there) */
    /* fp32 arithmetic instead of split-fp16 (zero-initialised blocks keep the split path): when packed_wino != NULL the
     * FFDNet pass runs on the Winograd fp32 kernel (scipnp_ffdnet_forward_c8w) from net_in_c8; packed_split / net_in_c8s
     * may then be NULL */
    const float* const* packed_wino;    /* nb layers packed by scipnp_pack_conv3x3_wino */
    float* net_in_c8;                   /* FFDNet input [B][2][M][N][8] fp32 */
    /* split-fp16 path only (NULL / zero: the thread's bound word; everything on one stream) */
    int* overflow_word;                 /* range-guard word of THIS solve (device int, zeroed by the caller) */
    scipnp_stream_t side_stream;        /* the caller's second stream for half of the frames of the network pass ... */
    void *side_fork_event, *side_join_event;   /* ... and its two hipEvent_t (as scipnp_ffdnet_forward_c8s_2s) */
    /* fp32 path only (NULL: every layer on the F(2x2,3x3) kernel): per-layer F(4x4,3x3) packings, NULL entries for the layers
     * that keep packed_wino[l] (scipnp_ffdnet_forward_c8w4) */
    const float* const* packed_wino4;
    /* Unit batch (round 4; 0 or 1: one problem): `units` independent problems of ONE shape that share the denoiser weights, stepped
     * by the same launch sequence -- state, Phi, orig in the unit-batched layout [B][units][4][M][N], y / Phisum
     * [units][4][M][N], RGB buffers and network buffers for B*units frames (frame f = t*units + u), sse_part with
     * B*units frames of partials.  Every unit's numbers are bit-identical to its own single-unit call. */
    int units;
    /* The arithmetic of the network pass, named (round 5; mirrors adaptivepnp_sci_amd.config.Config: precision / wino_f4).
     * 0: inferred from the pointers as before (packed_wino != NULL: fp32, with the F(4x4) kernel where packed_wino4[l] != NULL);
     * SCIPNP_CONV_SPLIT_F16 (1): split-fp16 -- packed_split and net_in_c8s required;
     * SCIPNP_CONV_F32_WINO_F2 (2): fp32 Winograd F(2x2,3x3) on every layer -- packed_wino and net_in_c8 required, packed_wino4 ignored;
     * SCIPNP_CONV_F32_WINO_F4 (3): fp32 with F(4x4,3x3) where packed -- packed_wino, packed_wino4 and net_in_c8 required.
     * A block whose pointers do not provide the named form is refused (SCIPNP_EINVAL), never silently run in another one. */
    int conv_form;
} scipnp_twostage_ffdnet_args;
#define SCIPNP_CONV_SPLIT_F16 1
#define SCIPNP_CONV_F32_WINO_F2 2
#define SCIPNP_CONV_F32_WINO_F4 3
int scipnp_twostage_ffdnet_iterate(const scipnp_twostage_ffdnet_args* a, int* nblocks, scipnp_stream_t s);

/* ADMM-TV iteration of either solver (dvp...:121-160, :265-271 two-stage; :385-407, :500-509 one-stage):
 * two_stage != 0: c0 = rho, c1 = alpha (theta = TV(x + b/rho), b += x - theta);
 * two_stage == 0: c0 = lambda, c1 = gamma (theta = TV(x - b), b -= x - theta).
 * Planes up to 256 columns with enough channels to fill the chip: three launches -- projection, the one-launch banded TV
 * kernel, dual update -- or TWO with defer_state (below); small problems (fewer than 128 bands of 32 rows,
 * planes up to 128 x 128): two launches, the projection and one whole-plane kernel for all TV iterations plus the dual update
 * (theta_raw is then left untouched); wider planes: projection, a launch per TV iteration, dual update.
 * sse_part must hold the partials of scipnp_pm_dual_update's grid (size query: scipnp_sse_partials); all of them are
 * written (the fused kernel zero-fills the entries it does not use), *nblocks receives their number. */
typedef struct {
    size_t struct_size;                 /* = sizeof(scipnp_admm_tv_args) */
    int M, N, B, two_stage;
    float *theta, *b, *x, *theta_raw;   /* state [B][4][M][N]; theta_raw: scratch for the unclipped TV output */
    const float *Phi, *y, *Phisum;
    double c0, c1;                      /* double for the same reason as scipnp_twostage_ffdnet_args.rho */
    float tv_weight;                    /* tv_weight = 0.1 in the reference */
    int tv_iters;                       /* n_iter_max = 5 in the reference */
    void* tv_workspace;                 /* scipnp_tv_workspace_bytes(M, N, 4*B, tv_iters) */
    size_t tv_workspace_bytes;
    const float* orig;
    double* sse_part;
    /* Deferred dual update (round 3; NULL: every call ends with theta, b and sse_part up to date).  defer_state points at a HOST
     * int owned by the caller, 0 before the first call.  On the banded path (and B <= 32) a call then ends after the TV step
     * and leaves ITS dual update pending (*defer_state = 1); the next call starts with ONE launch that performs the pending
     * dual update -- writing the previous call's squared-error partials to sse_part_prev -- together with its own projection:
     * an iteration is TWO launches (that launch + the banded TV kernel in its one-launch candidate form; theta_raw is then unused).  theta, b and sse_part are current again
     * after scipnp_admm_tv_flush (same argument block; a no-op when nothing is pending).  Same results bit for bit. */
    int* defer_state;
    double* sse_part_prev;              /* sse_part of the PREVIOUS call (written during this one), or NULL */
    /* Unit batch (round 4; 0 or 1: one problem): `units` independent problems of one shape in the unit-batched layout -- state
     * [B][units][4][M][N], y / Phisum [units][4][M][N], tv_workspace for 4*B*units channels; the same two or three launches
     * step all of them.  Bit-identical per unit to its own call. */
    int units;
} scipnp_admm_tv_args;
int scipnp_admm_tv_iterate(const scipnp_admm_tv_args* a, int* nblocks, scipnp_stream_t s);
int scipnp_admm_tv_flush(const scipnp_admm_tv_args* a, int* nblocks, scipnp_stream_t s);
/* 1 if scipnp_admm_tv_iterate on this block runs the whole-plane TV kernel with the dual update in its epilogue (its
 * squared-error partials: one per plane -- plane (t*units + u)*4 + ib -- then zeros up to the dual update's grid size);
 * 0: the banded kernel (partials as scipnp_pm_dual_update / the fused launch write them).  Host-only query. */
int scipnp_admm_tv_plane_path(const scipnp_admm_tv_args* a);

/* Unit batches: U independent problems of one shape stepped by ONE launch sequence (the reference loops its measurements one
 * after the other, two_stage_ADMM_Online_FFD_Warm.py:241-275; small units -- 256x256 tiles, ADMM-TV cubes -- leave most of
 * the chip idle that way).  Layout: frame-major with the unit inside, state [B][U][4][M][N], measurements [U][4][M][N]: the
 * projection is per mosaic pixel, so the U units are simply 4*M*N*U pixels to it; every other kernel of the path is per
 * frame or per plane and takes B*U frames / 4*B*U planes as it stands.  scipnp_pm_setup_units / scipnp_pm_project_units:
 * scipnp_pm_setup / scipnp_pm_project on that layout (utilspy.py:28-44, dvp...:128-140 / :389-391 per unit). */
int scipnp_pm_setup_units(const float* Phi, const float* y, float* Phisum, float* x0, int M, int N, int B, int units,
                          scipnp_stream_t s);
int scipnp_pm_project_units(const float* theta, const float* b, const float* Phi, const float* y, const float* Phisum, float* x,
                            int M, int N, int B, int units, int mode, float c0, float c1, scipnp_stream_t s);
/* squared-error partials the fused launch of scipnp_admm_tv_iterate's deferred form really writes (one per workgroup, each
 * inside ONE unit when 4*M*N is a multiple of its pixels per workgroup; the remaining nfill entries are zeros): lets a caller
 * cut the partials of a unit batch at unit boundaries.  0 for arguments the fused launch does not take. */
int scipnp_pm_dual_project_blocks(int M, int N, int B, int units, int nfill);

/* dual update of one ADMM iteration and projection of the next in ONE launch (csrc/sci_ops.hip; B <= 32:
 * scipnp_pm_dual_project_fits): theta = clip(theta_raw), b +-= x - theta, then x = project(theta, b) in place -- the
 * expressions of scipnp_pm_dual_update followed by scipnp_pm_project (mode, c0, c1 as there), bit-identical theta, b, x.
 * sse_part (NULL: none) receives one partial per workgroup and zeros up to nfill entries. */
int scipnp_pm_dual_project_fits(int M, int N, int B);
int scipnp_pm_dual_project(const float* theta_raw, float* x, float* theta, float* b, const float* Phi, const float* y,
                           const float* Phisum, const float* orig, double* sse_part, int nfill, int M, int N, int B, int mode,
                           float c0, float c1, scipnp_stream_t s);

/* ---------------------------------------------------------------------------------------------------------------
 * Layout steps of the stand-alone denoiser plug-ins (the solver fuses them into its pre/post kernels) -- csrc/plugin.hip
 * ------------------------------------------------------------------------------------------------------------- */

/* FFDNet.forward's input assembly, models/network_ffdnet.py:54-64: x planar [n][C][H][W] (C = 3 colour / 1 gray) ->
 * replicate-padded to even size, pixel-unshuffled (channel c*4 + dy*2 + dx), sigma as channel 4C, zero-padded to
 * CG = ceil((4C+1)/8) groups: out_c8 [n][CG][h][w][8] fp32, h = ceil(H/2), w = ceil(W/2) (scipnp_c8_to_c8s makes the
 * split-fp16 form of it). */
int scipnp_ffdnet_pack_input(const float* x, float sigma, float* out_c8, int n, int C, int H, int W, scipnp_stream_t s);

/* FFDNet.forward's output assembly, models/network_ffdnet.py:66-69: net_out c8 [n][ceil(4C/8)][h][w][8] ->
 * pixel-shuffled and cropped y planar [n][C][H][W]. */
int scipnp_ffdnet_unpack_output(const float* out_c8, float* y, int n, int C, int H, int W, scipnp_stream_t s);

/* small elementwise / reduction steps of the host loops, so that no PyTorch arithmetic runs on the path:
 * out = -in;  out = v + float32(float64(v) + noise) (FastDVDnet finetune input, test_fastdvdnet.py:359 with
 * utils_image.py:183-192);  out[r] = sum of row r of an fp64 table [rows][n] (PSNR / loss partial sums, fixed order) */
int scipnp_negate(const float* in, float* out, size_t n, scipnp_stream_t s);
int scipnp_fastdvd_noisy_input(const float* v, const double* noise, float* out, size_t n, scipnp_stream_t s);
int scipnp_sum_rows_f64(const double* part, double* out, int rows, int n, scipnp_stream_t s);

/* (H,W,3,B) cube -> (H,W,B) sum over the colour axis (packages/DDnet/DDnet_test.py: the demosaicker's input is the
 * mosaic, i.e. the channel sum of a CFA-sampled cube). */
int scipnp_cube_sum3(const float* cube, float* out, int H, int W, int B, scipnp_stream_t s);

/* ---------------------------------------------------------------------------------------------------------------
 * Grayscale (non-Bayer) PnP-ADMM mode -- csrc/gray.hip.  The reference has no grayscale solver: this is
 * its one-stage loop (dvp...:385-407, :500-509) with the Bayer split and the demosaic removed; projection, dual update
 * and PSNR partials are scipnp_pm_project / scipnp_pm_dual_update on any consistent per-pixel layout.
 * ------------------------------------------------------------------------------------------------------------- */

/* FFDNet-gray input from the pixel-unshuffled state [B][4][M][N]: in_c8 [B][1][M][N][8] = {x - b (4 values), sigma, 0,0,0}
 * (the 2x2 pixel-unshuffle of network_ffdnet.py:60-62 is exactly that state layout) */
int scipnp_gray_net_input(const float* x, const float* b, float sigma, float* in_c8, int M, int N, int B, scipnp_stream_t s);
/* FFDNet-gray output c8 [B][1][M][N][8] (4 real channels = the pixel-shuffle phases) -> state [B][4][M][N] */
int scipnp_gray_net_output(const float* out_c8, float* theta_raw, int M, int N, int B, scipnp_stream_t s);
/* (H,W,B) cube, frame index fastest (the reference's array layout) <-> [B][H][W] frames (TV prior on full frames) */
int scipnp_cube_to_frames(const float* cube, float* frames, int H, int W, int B, scipnp_stream_t s);
int scipnp_frames_to_cube(const float* frames, float* cube, int H, int W, int B, scipnp_stream_t s);

/* ---------------------------------------------------------------------------------------------------------------
 * DDnet deep demosaicking -- glue around scipnp_conv3x3_c8 / _c8s.
 * reference: models/network_demosaicking.py:381-463 (DDnet.forward), :186-244, :310-379 (DenBlocks),
 *            packages/DDnet/DDnet_test.py:166-216 (circular 5-frame window), dvp...:192-194, :242-244 (call sites)
 * ------------------------------------------------------------------------------------------------------------- */

/* v = x + coef*b of the plane-major state -> Bayer planes [B][4][M][N] and mosaic [B][2M][2N] (dvp...:168-171) */
int scipnp_pm_ddnet_inputs(const float* x, const float* b, float coef, float* planes, float* mosaic, int M, int N,
                           int B, scipnp_stream_t s);

/* DenBlock input of E evaluations: out[e] = cat_i( src[idx[e][i]] * scale[e][i] ), i = 0..2, src planar [F][C][h*w]
 * (C = 1, 3, 4), idx int32 [E][3] and scale float [E][3][C] in device memory (scale NULL = no multiply), zero-padded
 * to 8*ceil(3C/8) channels, written as fp32 c8 and/or c8s.  -- network_demosaicking.py:441-449 + torch.cat :228 */
int scipnp_ddnet_gather(const float* src, const int* idx, const float* scale, float* out_c8, void* out_c8s, int E,
                        int C, int h, int w, scipnp_stream_t s);

/* DenBlock residual `x = in1 + x` (network_demosaicking.py:241, :374): out planar [E][Cout][h*w] =
 * src[idx[e][1]][c or 0]*scale[e][1][.] + x_c8[e][.][c]; src NULL = plain c8 -> planar copy. */
int scipnp_ddnet_finish(const float* src, const int* idx, const float* scale, const float* x_c8, float* out, int E,
                        int C, int Cout, int h, int w, scipnp_stream_t s);

/* nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True) of planar [E][4][h][w] into the 8-channel c8 / c8s
 * input of the fusion block at [E][1][2h][2w]  -- network_demosaicking.py:375 */
int scipnp_bilinear_up2_c8(const float* in, float* out_c8, void* out_c8s, int E, int h, int w, scipnp_stream_t s);

/* x_out = gates[0][c]*branches[n] + gates[1][c]*branches[B+n], planar [2B][3][H*W] -> [B][3][H*W]  -- :461 */
int scipnp_ddnet_mix(const float* branches, const float* gates, float* out, int B, int H, int W, scipnp_stream_t s);

/* ---- online finetune of the demosaicker (`args.dm_update`, packages/DDnet/DDnet_test.py:248-296; adaptivepnp_sci_amd/ddnet_train.py):
 * scipnp_ddnet_loss_grad  MSE(input CFA-site cube, CFA samples of the output) over F*3*H*W elements (:208-216, :273-275): dout
 *                         [B][3][H][W] = its gradient, loss_part = per-block sums of the squared differences (fp64; *nblocks
 *                         entries; out == NULL: size query);
 * scipnp_ddnet_mix_bwd    adjoint of scipnp_ddnet_mix: d_branches [2B][3][H][W], part = 6 rows (gate (branch, channel)) of *ncols
 *                         partial sums of d gates;
 * scipnp_ddnet_finish_bwd the planar gradient [E][Cout][h][w] at a DenBlock's output as the c8 gradient of its 8-channel tail;
 * scipnp_ddnet_gather_bwd adjoint of scipnp_ddnet_gather plus the `in1 +` path of scipnp_ddnet_finish (d_center: gradient at the
 *                         block output, Cd channels; NULL: none): part = (E / Bn) * 3 * C rows (gate (window j, slot i, channel c),
 *                         evaluation e = j * Bn + n) of *ncols partial sums of g * src (NULL: no gate gradients), d_src = g * scale
 *                         scattered to the source frames (NULL: not wanted; every frame must be referenced once);
 * scipnp_bilinear_up2_bwd_c8  adjoint of scipnp_bilinear_up2_c8: d_in [E][4][h][w] from the c8 gradient at [E][1][2h][2w]. */
int scipnp_ddnet_loss_grad(const float* out, const float* mosaic, float* dout, double* loss_part, int H, int W, int B,
                           int* nblocks, scipnp_stream_t s);
int scipnp_ddnet_mix_bwd(const float* dout, const float* branches, const float* gates, float* d_branches, double* part, int B,
                         int H, int W, int* ncols, scipnp_stream_t s);
int scipnp_ddnet_finish_bwd(const float* d_out, float* d_x8, int E, int Cout, int h, int w, scipnp_stream_t s);
int scipnp_ddnet_gather_bwd(const float* d_tin_c8, const float* d_center, int Cd, const float* src, const int* idx,
                            const float* scale, float* d_src, double* part, int E, int Bn, int C, int h, int w, int* ncols,
                            scipnp_stream_t s);
int scipnp_bilinear_up2_bwd_c8(const float* d_up_c8, float* d_in, int E, int h, int w, scipnp_stream_t s);

/* HOST function (no GPU involved): n deviates of NumPy's LEGACY normal stream -- np.random.normal(loc, scale, n) on the
 * global RandomState, bit for bit -- from / to a generator state in np.random.get_state() form (key[624], pos,
 * has_gauss, cached_gaussian).  The reference draws its FastDVDnet finetune noise from that stream
 * (utils/utils_image.py:183-192, packages/fastdvdnet/test_fastdvdnet.py:359); NumPy holds the GIL while it does, this
 * entry does not. */
int scipnp_host_legacy_normal(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, double loc, double scale,
                              double* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* SCIPNP_H */
