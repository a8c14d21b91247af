/*
 * san_hip.h -- C ABI of libsan_hip.so: the MI355X (gfx950) kernels behind the
 * reconstruction + alignment hot path of woxuankai/SpatialAlignmentNetwork.
 *
 * The reference has NO native/FFI layer (it is pure Python on PyTorch ATen), so
 * there is no existing binding to inherit; each entry point below names the
 * reference call site (file:line in the reference repo) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would
 * add at each of those call sites.
 *
 * Conventions
 *  - Plain C types only.  Every pointer is a DEVICE pointer unless it says "host".
 *  - Ownership: the caller owns every buffer, including outputs and workspaces.
 *    The library allocates device memory only for immutable twiddle tables
 *    (once per FFT length per device; call san_fft_prepare() before hipGraph
 *    capture).  It keeps no caller pointer after a call returns.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*); no
 *    hidden synchronisation, nothing on the default stream.
 *  - Real tensors are fp32 NCHW contiguous.  A "channel view" (ptr, ctot, coff)
 *    addresses channels [coff, coff+C) of a tensor that has `ctot` channels, so
 *    producers can write straight into a concatenation buffer.
 *  - Complex tensors are interleaved (re, im) fp32 pairs = torch.view_as_real,
 *    shape [N, C, H, W]; "planar" complex is a real [N, 2, H, W] (re plane,
 *    im plane) = the reference's complex_to_chan_dim layout (varnet.py:246-248).
 *  - Lazy normalisation: activations are stored RAW (conv output) next to a
 *    per-(sample, channel) affine (scale, shift) and a LeakyReLU slope; every
 *    consumer applies  lrelu(scale*x + shift, slope)  while loading.  A NULL
 *    scale/shift pair means identity.
 *  - Return value: 0 = success; negative = SAN_E_* argument error; positive =
 *    a raw hipError_t from the launch.  Nothing throws across this boundary.
 */
#ifndef SAN_HIP_H
#define SAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAN_OK 0
#define SAN_E_ARG (-1)         /* bad pointer / dimension                       */
#define SAN_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for  */
#define SAN_E_WORKSPACE (-3)   /* workspace too small                           */

/* norm finalisation modes (san_norm_finalize) */
#define SAN_NORM_INSTANCE 0 /* InstanceNorm2d: biased var, eps inside sqrt (varnet.py:141) */
#define SAN_NORM_GROUP 1    /* NormUnet.norm: unbiased std, eps added to std (varnet.py:257-268) */
#define SAN_NORM_BATCH 2    /* BatchNorm2d training: batch stats over N,H,W (unet.py:125)        */
#define SAN_NORM_GROUP_BWD 3 /* GROUP + the two per-plane values its backward needs (aux arrays [2][n][c])  */

const char* san_last_error_string(void);
int san_version(void);

/* ---------------------------------------------------------------- FFT + DC */

/* Build (and cache) the twiddle tables for lengths h and w on the current
 * device.  Optional; required before capturing FFT calls into a hipGraph. */
int san_fft_prepare(int h, int w);

/* Bytes of workspace the FFT-based entry points need for [planes, h, w]. */
size_t san_fft_workspace_bytes(int planes, int h, int w);

/* out = (i)fft2(in * colmask_in[w]) * colmask_out[w], ortho norm, no shift.
 * in/out: interleaved complex [planes, h, w]; masks: fp32 [w] or NULL.
 * planar_ctot == 0: interleaved output.  planar_ctot >= 2: write PLANAR into a
 * real tensor [planes, planar_ctot, h, w] (re -> channel 0, im -> channel 1).
 * Replaces signal_utils.py:4-12 (fft2/ifft2), model.py:110-114 (set_input),
 * varnet.py:395-402 (ACS window + ifft2). */
int san_fft2(const float* in, float* out, int planes, int h, int w, int inverse,
             const float* colmask_in, const float* colmask_out, int planar_ctot,
             void* ws, size_t ws_bytes, void* stream);

/* m[n] = sum_c ifft2(k[n,c]) * conj(sens[n,c]); out is PLANAR: channels 0 (re)
 * and 1 (im) of a real tensor [n, out_ctot, h, w] (out_ctot = 3 when the
 * reference image rides along as channel 2: varnet.py:319).
 * Replaces VarNetBlock.sens_reduce (varnet.py:511-512) + complex_to_chan_dim. */
int san_sens_reduce(const float* k, const float* sens, float* out_planar, int out_ctot, int n, int c, int h, int w,
                    void* ws, size_t ws_bytes, void* stream);

/* k_out = k - dc_w * mask[w] * (k - k0) - fft2(r * sens)   with r PLANAR [n,2,h,w].
 * mask: fp32 [w] (1 = sampled column); dc_w: device pointer to one fp32.
 * Replaces chan_dim_to_complex + sens_expand + soft DC + combine
 * (varnet.py:250-255, :508-509, :527-530). k_out may alias k. */
int san_sens_expand_dc(const float* r_planar, const float* sens, const float* k, const float* k0,
                       const float* mask, const float* dc_w, float* k_out, int n, int c, int h, int w,
                       void* ws, size_t ws_bytes, void* stream);

/* The same two calls with the cascade boundary fused: san_sens_expand_dc_next additionally writes
 * next_cols [n,c,h,w] complex = the (unnormalised) inverse column transform of k_out, i.e. the
 * first pass of the NEXT cascade's sens_reduce (varnet.py:511 right after :530), computed while
 * k_out is still in registers; san_sens_reduce_from_cols consumes it and runs only the row pass.
 * Results are identical to san_sens_expand_dc followed by san_sens_reduce. */
int san_sens_expand_dc_next(const float* r_planar, const float* sens, const float* k, const float* k0,
                            const float* mask, const float* dc_w, float* k_out, float* next_cols,
                            int n, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);
int san_sens_reduce_from_cols(const float* k_cols, const float* sens, float* out_planar, int out_ctot,
                              int n, int c, int h, int w, void* ws, size_t ws_bytes, void* stream);
/* san_ifft2_rss (varnet.py:486) on the same precomputed column transform. */
int san_ifft2_rss_from_cols(const float* k_cols, float* out, int n, int c, int h, int w,
                            void* ws, size_t ws_bytes, void* stream);

/* out[n,0] = sqrt(sum_c |ifft2(k[n,c])|^2)   (real [n,1,h,w]).
 * Replaces rss(ifft2(kspace_pred)) (varnet.py:486). */
int san_ifft2_rss(const float* k, float* out, int n, int c, int h, int w,
                  void* ws, size_t ws_bytes, void* stream);

/* sens[n,c] = est[n,c] / (rss_c(est[n]) + 1e-6), est PLANAR [n*c, 2, h, w] ->
 * sens interleaved [n, c, h, w].  Replaces varnet.py:418-419. */
int san_sens_normalize(const float* est_planar, float* sens, int n, int c, int h, int w, void* stream);

/* out[n,0] = sqrt(sum_c x^2) for real x, or sqrt(sum_c re^2+im^2) for complex
 * interleaved x.  Replaces signal_utils.rss (signal_utils.py:24-26). */
int san_rss(const float* x, float* out, int n, int c, int hw, int is_complex, void* stream);

/* ------------------------------------------------------- conv / norm stack */

/* Repack weights into the layout the kernels read (zero padded):
 *   san_conv_pack_weights_fwd : Conv2d weight [cout, cin, ks, ks], ks in {1,3}, for
 *                               san_conv2d_fwd (MFMA 4x4x1 outer-product layout
 *                               [group][cin][tap][4][quads]);
 *   san_conv_pack_weights     : transposed != 0: ConvTranspose2d weight [cin, cout, 2, 2]
 *                               for san_tconv2x2_fwd (same layout over 4*cout
 *                               virtual channels); transposed == 0 forwards to _fwd.
 * san_conv_packed_floats(cout, cin, ks) gives the element count of `packed`
 * (ks = 2 for the transposed convolution). */
size_t san_conv_packed_floats(int cout, int cin, int ks);
int san_conv_pack_weights_fwd(const float* w, float* packed, int cout, int cin, int ks, void* stream);
int san_conv_pack_weights(const float* w, float* packed, int cout, int cin, int ks, int transposed,
                          void* stream);

/* y[n, y_coff+co] = bias[co] + sum_{ci,ky,kx} Wp[co,ci,ky,kx] * T(x)[n, x_coff+ci, .+ky-p, .+kx-p]
 * with T = lazy normalisation (in_scale/in_shift, in_slope),
 * zero padding applied AFTER T.  ks in {1, 3}.  Optional output affine
 * y = y*out_scale[n,co] + out_shift[n,co] (used for NormUnet.unnorm).
 * in_scale/in_shift are laid out like x's channel axis: fp32 [n, x_ctot],
 * read at [n, x_coff + ci].
 * If part_stats != NULL the kernel also writes per-tile (count, mean, M2) of the
 * raw output for every (n, co): fp32 [n, cout, tiles, 3], tiles from
 * san_conv_stat_tiles(n, h, w, cin, cout, ks); feed them to san_norm_finalize.
 * Replaces F.conv2d at varnet.py:78,140,143 and unet.py:119-140,185-186. */
int san_conv_stat_tiles(int n, int h, int w, int cin, int cout, int ks);
/* Round 5: layers with at most 4 channels on one side (the cascade's 4 -> 18 input and 18 -> 2 output convolutions of
 * varnet.py:139-146,96-104 and their data gradients) run on a direct fp32 kernel behind san_conv2d_fwd; 0 switches that off
 * (everything on the outer-product kernel: A/B and tests), -1 only asks.  Returns the previous setting. */
int san_conv_direct_enable(int on);
/* Round 5: 1x1 and transposed convolutions on fp16-format weights (unet.py's 1x1 layers, varnet.py:159-192 and their data
 * gradients) run as one-stage GEMMs behind san_conv1x1_bf16x3_fwd / san_tconv2x2_bf16x3_fwd / san_conv_bf16x3_dgrad_amax;
 * 0 switches that off (the tiled kernel's KS = 1 form, as before), -1 only asks.  Returns the previous setting. */
int san_conv1x1_gemm_enable(int on);
int san_conv2d_fwd(const float* x, int x_ctot, int x_coff, int cin,
                   const float* in_scale, const float* in_shift, float in_slope,
                   const float* w_packed, const float* bias,
                   float* y, int y_ctot, int y_coff, int cout,
                   const float* out_scale, const float* out_shift,
                   float* part_stats,
                   int n, int h, int w, int ks, void* stream);

/* ---- backward building blocks (training path; forward semantics above) ----
 * Data gradient of san_conv2d_fwd: dx = san_conv2d_fwd(dy, packed_dgrad, cin' = cout, cout' = cin)
 * with packed_dgrad = san_conv_pack_weights_dgrad(w_forward) (flipped taps, swapped channel
 * axes; buffer of san_conv_packed_floats(cin, cout, ks) floats).
 * Weight gradient: dw [cout, cin, ks, ks] (+)= sum_{n,y,x} dy[n,co,y,x] * T(x)[n,ci,y+ky-p,x+kx-p];
 * partial: fp32 [san_conv_wgrad_partitions(n,h,w,cin,cout,ks) * cout*cin*ks*ks] scratch.
 * san_act_bwd: gradient through the lazy (scale, shift, LeakyReLU) read of a raw tensor y:
 *   yh = sc*y + sh, u = g * (yh >= 0 ? 1 : slope);
 *   mode 0: dy = sc*u;  mode 1 (InstanceNorm): dy = sc*(u - mean(u) - yh*mean(u*yh)) per plane;
 *   part: fp32 [n, c, san_bwd_stat_tiles(hw), 2] scratch (mode 1). */
int san_conv_pack_weights_dgrad(const float* w, float* packed, int cout, int cin, int ks, void* stream);
int san_conv_wgrad_partitions(int n, int h, int w, int cin, int cout, int ks);
int san_conv2d_wgrad(const float* x, int x_ctot, int x_coff, int cin,
                     const float* in_scale, const float* in_shift, float in_slope,
                     const float* dy, int dy_ctot, int dy_coff, int cout,
                     float* dw, int accumulate, float* partial,
                     int n, int h, int w, int ks, void* stream);
int san_bwd_stat_tiles(int hw);
/* part[n, c, san_bwd_stat_tiles(hw), 2] = per-chunk (sum u, sum u*yh), yh = sc*y+sh,
 * u = g*(yh >= 0 ? 1 : slope): the two plane reductions every normalisation backward needs. */
int san_plane_dot_stats(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                        const float* sc, const float* sh, float slope, float* part,
                        int n, int c, int hw, void* stream);
/* san_act_bwd_amax with a destination mode.  flags & 1: dy is stored pixel-UNSHUFFLED as [n, d_ctot, h/2, w/2], channel
 * d_coff + 4 ch + 2 (row & 1) + (col & 1) -- the backward of ConvTranspose2d as a 1x1 convolution to 4 Cout virtual channels
 * (varnet.py:176-182) wants dy in that order, and a separate san_unshuffle2_fwd pass over the tensor is saved (w = plane width,
 * w % 4 == 0, even height, 16-byte aligned g / y).  flags & 2: dy += (gradient accumulation, e.g. dL/d ref over the cascades,
 * varnet.py:478-484).  amax may be NULL. */
int san_act_bwd_ex_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                        const float* sc, const float* sh, float slope, int mode, float* part,
                        float* dy, int d_ctot, int d_coff, void* amax, int n, int c, int hw, int w, int flags, void* stream);
int san_act_bwd(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                const float* sc, const float* sh, float slope, int mode, float* part,
                float* dy, int d_ctot, int d_coff, int n, int c, int hw, void* stream);

/* k-space / loss backward pieces (PyTorch's complex-gradient convention g = dL/dRe + i dL/dIm):
 *   san_dc_weight_grad    : partial[256] chunks of  -sum mask[w] * Re(conj(G)*(k-k0))  (varnet.py:527-528)
 *   san_sens_grad_acc     : gS[n,c] += sign1*conj(r[n])*t1[n,c] + x[n,c]*conj(gm[n])  (r, gm planar [n,2,hw]):
 *                           the sensitivity-map gradient of one cascade (sens_expand + sens_reduce)
 *   san_sens_normalize_bwd: backward of S = est/(rss(est)+1e-6)                        (varnet.py:418-419)
 *   san_rss_bwd           : gx_c = g * x_c / rss                                        (signal_utils.py:24-26)
 *   san_ssim_loss_bwd     : gy = gscale * d(1 - mean SSIM(x, y))/dy; ws: fp32 [3*n*(h-6)*(w-6)]  (ssimloss.py:11-40) */
int san_dc_weight_grad(const float* g, const float* k, const float* k0, const float* mask, float* partial,
                       int planes, int h, int w, void* stream);

/* dst[0] += scale * sum(part[0 .. count)), accumulated in double in a fixed order by one workgroup: a scalar parameter
 * gradient that arrives as per-workgroup partials (san_dc_rows backward: dc_weight, varnet.py:523) without host-side glue. */
int san_partials_add(const float* part, int count, float scale, float* dst, void* stream);
int san_sens_grad_acc(float* gs, const float* r_planar, const float* t1, const float* x, const float* gm_planar,
                      float sign1, int n, int c, int hw, void* stream);
/* Image-domain cascade backward (san_dc_rows, backward form): the accumulation of san_sens_grad_acc (skipped when gs is
 * NULL) and, in the same pass, gd[n,c] += gm[n] * sens[n,c]: the gradient that reaches the cascade's input through
 * m = sum_c conj(S_c) x_c joins the one through the data-consistency path. */
int san_sens_grad_prop(float* gs, const float* r_planar, const float* t1, const float* x, const float* gm_planar,
                       float sign1, float* gd, const float* sens, int n, int c, int hw, void* stream);
int san_sens_normalize_bwd(const float* est_planar, const float* gs, float* gest_planar, int n, int c, int hw,
                           void* stream);
int san_rss_bwd(const float* x, const float* y, const float* g, float* gx, int n, int c, int hw, int is_complex,
                void* stream);
int san_ssim_loss_bwd(const float* x, const float* y, float* gy, float gscale, int n, int h, int w, float* ws,
                      void* stream);
/* The same with the upstream gradient also taken from device memory (gscale_dev: fp32 [1] or NULL): the scale
 * -(gscale * gscale_dev[0]) / (n (h-6) (w-6)) is formed inside the kernel, so autograd's grad_output never visits the host and
 * (gscale = w, NULL) gives the same bits as (gscale = 1, gscale_dev -> w).  SSIM is symmetric in its arguments: the
 * gradient wrt x is this call with x and y exchanged.  Replaces autograd through ssimloss.py:11-40. */
int san_ssim_loss_bwd_dev(const float* x, const float* y, float* gy, float gscale, const float* gscale_dev, int n, int h,
                          int w, float* ws, void* stream);

/* san_act_bwd_coef: dy = sc*(u - m1 - (p*yh + q)*m2) with coef[n, c, 4] = (m1, m2, p, q) given by
 *   the caller (BatchNorm training backward: statistics over N,H,W reduced on the host from
 *   san_plane_dot_stats, unet.py:125).
 * san_warp_bwd_grid: gradient of the bilinear warp wrt the sampling grid, as NCHW [n,2,h,w]
 *   (= gradient wrt the predicted offset field; cross.py:29-34).
 * san_gradient_loss_bwd: g (+)= gscale * d gradient_loss / d offset (model.py:21-28). */
int san_act_bwd_coef(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                     const float* sc, const float* sh, float slope, const float* coef,
                     float* dy, int d_ctot, int d_coff, int n, int c, int hw, void* stream);
/* One-launch finalisations of the backward tapes (they replace chains of per-channel host-side tensor
 * arithmetic).  part: the chunk sums san_plane_dot_stats / san_plane_stats wrote.
 * san_bn_bwd_finalize: BatchNorm2d training backward (unet.py:125): dgamma[c] += (S2 - beta S1)/gamma,
 *   dbeta[c] += S1, coef[n,c,4] = (dbeta/cnt, dgamma/cnt, 1/gamma, -beta/gamma) for san_act_bwd_coef.
 * san_bias_grad_from_stats: db[c] += sum over samples and chunks of count*mean (part [n,c,tiles,3]).
 * san_normunet_bwd_coefs: NormUnet.norm/unnorm backward (varnet.py:246-332): the two lazy affines
 *   (a_sc, a_sh over g_ctot channels; m_sc, m_sh over x_ctot channels; zero beyond channel 1) with which
 *   dL/dm = (a_sc g_xh + a_sh) + (m_sc m + m_sh). */
int san_bn_bwd_finalize(const float* part, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                        float* coef, int n, int c, int tiles, double cnt, void* stream);
int san_bias_grad_from_stats(const float* part, float* db, int n, int c, int tiles, void* stream);
/* BatchNorm2d running statistics in one launch: running = (1-m) running + m batch (variance times var_factor),
 * *num_batches_tracked += 1 (int64, may be NULL).  unet.py:125 (torch.nn.BatchNorm2d training forward). */
int san_bn_update_running(float* rmean, float* rvar, long long* num_batches_tracked, const float* bmean,
                          const float* bvar, int c, float momentum, float var_factor, void* stream);
int san_normunet_bwd_coefs(const float* part_b, const float* part_a, int tiles, const float* scale,
                           const float* shift, int x_ctot, const float* stdv, double nel,
                           float* a_sc, float* a_sh, int g_ctot, float* m_sc, float* m_sh, int b, void* stream);
/* NormUnet backward as two launches around the U-Net's own (round 6; varnet.py:246-332 backwards, csrc/san_bwd.hip).
 * san_normunet_bwd_head: g_u = g_out * std (dL/dU) and part_b[b, 2, san_bwd_stat_tiles(hw), 2] = chunk sums (sum g_out,
 *   sum g_out U), U = out_planar * isd + nshift -- san_plane_dot_stats + san_apply_fwd in one pass; the last 1x1 convolution's
 *   bias gradient sum(g_u) = std * sum(g_out) is taken from part_b by the tail (no san_plane_stats / san_bias_grad_from_stats).
 * san_normunet_bwd_tail: san_normunet_bwd_coefs + san_add_fwd + san_sens_grad_prop + the reference channel's InstanceNorm
 *   backward + san_partials_add in one pass: every workgroup derives the affines of dL/dm from part_b and part_x (= san_plane_dot_stats
 *   of (g_xh, xin) over the first xc = 2 or 3 channels), forms g_m per pixel without storing it, and applies gd[n, c] += g_m S[n, c]
 *   (+ the sensitivity-map accumulation when gs != NULL: gs += sign1 conj(r) t1 + xs conj(g_m)); g_ref [b, 1, hw] (NULL: none;
 *   needs xc = 3) (+)= the gradient wrt the reference input; db[2] += the bias gradient; dcw[0] += dcw_scale * sum(dcw_part[0 ..
 *   dcw_count)) (dc_weight, varnet.py:523).  b = samples, c = coils. */
int san_normunet_bwd_head(const float* g_out, const float* out_planar, const float* isd, const float* nshift, const float* stdv,
                          float* g_u, float* part_b, int b, int hw, void* stream);
int san_normunet_bwd_tail(const float* part_b, const float* part_x, int xc, const float* xin, int x_ctot, const float* x_scale,
                          const float* x_shift, const float* stdv, double nel, const float* g_xh, int g_ctot, float* gd,
                          const float* sens, float* gs, const float* r_planar, const float* t1, const float* xs, float sign1,
                          float* g_ref, int ref_accumulate, float* db, const float* dcw_part, int dcw_count, float dcw_scale,
                          float* dcw, int b, int c, int hw, void* stream);
int san_warp_bwd_grid(const float* img, const float* grid, const float* g, float* g_off,
                      int n, int c, int h, int w, void* stream);
int san_gradient_loss_bwd(const float* offset, float* g, float gscale, int accumulate, int n, int h, int w,
                          void* stream);
/* The same with the upstream gradient also taken from device memory (see san_ssim_loss_bwd_dev). */
int san_gradient_loss_bwd_dev(const float* offset, float* g, float gscale, const float* gscale_dev, int accumulate, int n,
                              int h, int w, void* stream);

/* ConvTranspose2d 2x2 stride 2, no bias: y [n, cout, 2h, 2w].
 * w_packed from san_conv_pack_weights(..., ks=2, transposed=1).
 * Evaluated as a 1x1 convolution to 4*cout virtual channels (one per tap) on
 * the MFMA kernel + a 2x2 pixel shuffle in the store.
 * part_stats: fp32 [n, cout, san_tconv_stat_tiles(n, h, w, cout), 3].
 * Replaces nn.ConvTranspose2d at varnet.py:177-179. */
int san_tconv_stat_tiles(int n, int h, int w, int cout);
int san_tconv2x2_fwd(const float* x, int x_ctot, int x_coff, int cin,
                     const float* in_scale, const float* in_shift, float in_slope,
                     const float* w_packed,
                     float* y, int y_ctot, int y_coff, int cout,
                     float* part_stats,
                     int n, int h, int w, void* stream);

/* Merge per-tile partials into the lazy-normalisation affine of the tensor:
 *   INSTANCE: scale = rsqrt(var_b + eps),        shift = -mean*scale   per (n,c)
 *   GROUP   : scale = 1/(std_unbiased + eps),    shift = -mean*scale   per (n,c);
 *             also aux_a[n,c] = std, aux_b[n,c] = mean (for unnorm)
 *   GROUP_BWD: GROUP with aux_a, aux_b of [2][n][c]: [0] as above, aux_a[1] = std > 0 ? 1/std : 0 and
 *             aux_b[1] = -mean * aux_a[1] (the affine that recovers the U-Net output from the un-normalised one: the
 *             training tape of NormUnet, so that its backward needs no element-wise host glue)
 *   BATCH   : stats over all n; scale = gamma*rsqrt(var_b+eps), shift = beta-mean*scale,
 *             broadcast to every n; aux_a[c] = batch mean, aux_b[c] = unbiased
 *             batch var (for the running-stat update).
 * part: fp32 [n, c, tiles, 3]; scale/shift: fp32 views [n, sc_ctot] at channel
 * offset sc_coff.  Replaces the statistics half of InstanceNorm2d / x.std() /
 * BatchNorm2d (varnet.py:141,235,257-268; unet.py:125). */
int san_norm_finalize(const float* part, int n, int c, int tiles, int mode, float eps,
                      const float* gamma, const float* beta,
                      float* scale, float* shift, int sc_ctot, int sc_coff,
                      float* aux_a, float* aux_b, void* stream);

/* BATCH finalisation and BatchNorm2d's running statistics in ONE launch (san_norm_finalize mode BATCH followed by
 * san_bn_update_running: 26 extra launches per training step of the alignment network before):
 * running = (1-m) running + m batch, the unbiased batch variance times var_factor, *num_batches_tracked += 1 (may be NULL).
 * unet.py:125 (torch.nn.BatchNorm2d training forward). */
int san_norm_finalize_bn(const float* part, int n, int c, int tiles, float eps, const float* gamma, const float* beta,
                         float* scale, float* shift, int sc_ctot, int sc_coff, float* aux_a, float* aux_b, float* rmean,
                         float* rvar, long long* num_batches_tracked, float momentum, float var_factor, void* stream);

/* (count, mean, M2) partials of an existing tensor view, san_plane_stat_tiles(hw)
 * chunks per plane: part [n, c, tiles, 3].  Used for tensors no conv produced
 * (ref image, sens_reduce output). */
int san_plane_stat_tiles(int hw);
int san_plane_stats(const float* x, int x_ctot, int x_coff, int c, int n, int hw,
                    float* part, void* stream);

/* Eval-mode BatchNorm as a lazy affine: scale = gamma*rsqrt(rvar+eps),
 * shift = beta - rmean*scale, broadcast over n.  (unet.py:125 in eval()). */
int san_bn_eval_affine(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                       float eps, float* scale, float* shift, int sc_ctot, int sc_coff,
                       int n, int c, void* stream);

/* Element-wise materialisers, all of the form  y = op(T(x)):
 *   avgpool2 : y [n,c,h/2,w/2] = mean of 2x2 block       (varnet.py:98, unet.py:137)
 *   upsample2: y [n,c,2h,2w]   = nearest                 (unet.py:130)
 *   add      : y = T_a(a) + T_b(b)                       (unet.py:24 ResSequential)
 *   apply    : y = T(x)                                   (plain materialise)
 */
int san_avgpool2_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope,
                     float* y, int y_ctot, int y_coff, int n, int c, int h, int w, void* stream);
/* InstanceNorm finalisation + 2 x 2 average pooling in ONE launch (round 6): san_norm_finalize(SAN_NORM_INSTANCE) of x's records
 * and san_avgpool2_fwd of x read through the affine just computed -- the encoder levels' conv -> IN -> LeakyReLU -> avg_pool2d
 * (varnet.py:95-99).  Every workgroup of a plane merges the plane's records itself (the same order and arithmetic as
 * san_norm_finalize: identical bits), the first writes scale / shift, each pools its share.  h, w even. */
int san_norm_finalize_pool(const float* part, int n, int c, int tiles, float eps, float* scale, float* shift, int sc_ctot, int sc_coff,
                           const float* x, int x_ctot, int x_coff, float slope, float* y, int y_ctot, int y_coff, int h, int w,
                           void* stream);
/* Channel `ch` of src [n, ctot, hw] (raw values) with its lazy-affine entries src_scale / src_shift [n, ctot] copied into `count`
 * (<= 16) tensors of the same shape in ONE launch; dst / dst_scale / dst_shift are HOST arrays of `count` device pointers (read
 * during the call).  The cascades' shared reference channel: NormUnet's `ref` input is InstanceNorm-ed once and concatenated into
 * every cascade's U-Net input (varnet.py:315-319) -- three launches per cascade before. */
int san_replicate_channel(const float* src, const float* src_scale, const float* src_shift, const void* dst, const void* dst_scale,
                          const void* dst_shift, int count, int n, int ctot, int ch, int hw, void* stream);
int san_upsample2_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope,
                      float* y, int y_ctot, int y_coff, int n, int c, int h, int w, void* stream);
/* y[n, 4c+2dy+dx, i, j] = x[n, c, 2i+dy, 2j+dx]: x [n,c,2h,2w] -> y [n,4c,h,w] (h, w = OUTPUT dims).
 * Turns the transposed convolution's output gradient into the 4*cout virtual channels of its
 * 1x1-conv form, so its data / weight gradients reuse the conv kernels. */
int san_unshuffle2_fwd(const float* x, int x_ctot, int x_coff, float* y, int y_ctot, int y_coff,
                       int n, int c, int h, int w, void* stream);
int san_add_fwd(const float* a, int a_ctot, int a_coff, const float* a_sc, const float* a_sh, float a_slope,
                const float* b, int b_ctot, int b_coff, const float* b_sc, const float* b_sh, float b_slope,
                float* y, int y_ctot, int y_coff, int n, int c, int hw, void* stream);
int san_apply_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope,
                  float* y, int y_ctot, int y_coff, int n, int c, int hw, void* stream);
/* Copy between planes of different sizes: y[n, c, oy, ox] = T(x)[n, c, oy - off_y, ox - off_x] (T = x's lazy read).
 * mode 0: zero where the source index falls outside x -- NormUnet.pad (zero-pad the NORMALISED image to multiples
 *         of 16, varnet.py:275-289: off = the top / left pad) and NormUnet.unpad (varnet.py:291-299: negative off);
 * mode 1: x plus one reflected row / column at the bottom / right (hy - hx, wy - wx in {0, 1}; F.pad(..., 'reflect')
 *         of the U-Net's up path for odd sizes, varnet.py:107-114);
 * mode 2: the adjoint of mode 1 (x = gradient of the padded plane, y = gradient of the unpadded one). */
int san_window_copy_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope,
                        int hx, int wx, float* y, int y_ctot, int y_coff, int hy, int wy, int off_y, int off_x,
                        int mode, int n, int c, void* stream);

/* Narrow-precision modes of every bf16 matrix-core convolution / transposed convolution / weight gradient
 * (BASELINE.json configs[1] names bf16, configs[4] fp8 U-Net convolutions; the reference's seam is
 * torch.cuda.amp.autocast, model.py:83-87,104).  parts = 3: operands split in three bf16 parts, six products per MAC,
 * fp32-equivalent (default, the only mode held to the 1e-4 parity bar); 2: two parts (16 mantissa bits), three
 * products; 1: plain bf16, one product.  Judged by PSNR against the parts = 3 output.  FFT, data consistency,
 * normalisation statistics and losses are fp32 in every mode.  Process-wide; returns SAN_E_ARG outside 1..3. */
int san_set_conv_precision(int parts);
int san_get_conv_precision(void);

/* fp8 forward operands (BASELINE.json configs[4]: "fp8 (CDNA4 MFMA) U-Net conv path + fp32 FFT/DC").  Pack the weights with
 * mode + 32 (san_conv_bf16x3_pack_ks / _pack_job_ks): one OCP e4m3 value per weight, multiplied by the per-tensor power of
 * two S_w = 2^(7 - floor(log2 max |w|)) that the packing computes on the device and keeps behind the image.  With parts = 1
 * selected, san_conv2d_bf16x3_fwd / _fwd_ws / san_conv1x1_bf16x3_fwd / san_tconv2x2_bf16x3_fwd pick the format up from the
 * packed image: activations x 8 -> e4m3 (clamped to +-448) while staged, v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulators
 * x 1 / (8 S_w) before bias / statistics.  Gradients (data, weight) of the mode run on plain bf16 (mode without + 32). */

/* fp16-format gradients (fp32-equivalent mode only).  The matrix-core kernels can run on TWO fp16 parts per operand (22
 * mantissa bits, three products instead of the six of the bf16 split) when their operands fit fp16's range.  Forward
 * operands (normalised activations, weights) do: pack the weights with mode + 16 and the convolution entry points pick
 * the format up from the packed image.  Gradients (1e-7-sized) need a scale: the `_amax` forms of san_act_bwd /
 * san_act_bwd_coef keep the largest |dy| of everything written to a dy tensor in an AMAX RECORD: san_amax_record_words()
 * uint32 values (64 lines of 128 bytes, the float bits of a maximum in the first word of each line), zeroed by the caller once
 * per step; every workgroup folds its maximum into one of the lines with an integer atomic max (order-independent:
 * deterministic; 64 lines because same-address atomics serialise), and the `_amax` forms of the data / weight gradient take
 * the maximum over the lines, multiply dy by 2^(13 - floor(log2 max)) while loading and the result by the inverse.  No
 * finalising launch, nothing synchronises with the host.  amax == NULL: the plain forms. */
int san_amax_record_words(void);
int san_act_bwd_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                     const float* sc, const float* sh, float slope, int mode, float* part,
                     float* dy, int d_ctot, int d_coff, void* amax, int n, int c, int hw, void* stream);
/* san_act_bwd_amax with g(p) + g2_scale * g2(p / 2) as the incoming gradient (g2: [n, g2_ctot, h/2, w/2], c channels from
 * g2_coff): the U-Net encoder's "skip-connection gradient + average-pool adjoint" (varnet.py:118-134 under autograd: the
 * avg_pool2d backward and the gradient accumulation at the block output) without materialising the up-sampled tensor or
 * the sum.  hw = h * w, w % 4 == 0, h even, 16-byte aligned tensors; amax may be NULL. */
int san_act_bwd_up_amax(const float* g, int g_ctot, int g_coff, const float* g2, int g2_ctot, int g2_coff, float g2_scale,
                        const float* y, int y_ctot, int y_coff, const float* sc, const float* sh, float slope, int mode,
                        float* part, float* dy, int d_ctot, int d_coff, void* amax, int n, int c, int hw, int w, void* stream);
int san_act_bwd_coef_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                          const float* sc, const float* sh, float slope, const float* coef,
                          float* dy, int d_ctot, int d_coff, void* amax, int n, int c, int hw, void* stream);

/* One-pass norm + activation backward on WORKGROUP CLUSTERS (round 6; csrc/san_bwd.hip act_bwd_cluster_kernel): K consecutive
 * workgroups share one reduction domain, keep their chunk of the plane in registers, exchange their two partial sums through a
 * small record in `sync` and write dy -- g and y are read once instead of twice, one launch instead of two (InstanceNorm; replaces
 * san_plane_dot_stats + san_act_bwd*_amax where a plane does not fit one workgroup, i.e. the 320 x 320 levels of varnet.py:139-146's
 * backward) or three (training BatchNorm, unet.py:125: san_plane_dot_stats + san_bn_bwd_finalize + san_act_bwd_coef_amax).
 * sync: int32 scratch of san_*_sync_words(n, c, hw) words that the caller zeroes ONCE; the kernels leave it zero.  A words
 * query of 0 means: shape not covered, use the multi-launch form.  Tensors 16-byte aligned, hw % 4 == 0.
 * san_act_bwd_in: g2 (may be NULL) / g2_scale as san_act_bwd_up_amax; flags / w as san_act_bwd_ex_amax; amax may be NULL.
 * san_bn_act_bwd: dgamma[c] / dbeta[c] are accumulated into.
 * san_act_bwd_cluster_set_tuning(on, min_hw, v, bn_on): -1 / 0 / 0 / -1 leave a setting unchanged (process-wide; tests, A/B runs). */
int san_act_bwd_in_sync_words(int n, int c, int hw);
int san_act_bwd_in(const float* g, int g_ctot, int g_coff, const float* g2, int g2_ctot, int g2_coff, float g2_scale,
                   const float* y, int y_ctot, int y_coff, const float* sc, const float* sh, float slope, float* dy, int d_ctot,
                   int d_coff, void* amax, int n, int c, int hw, int w, int flags, void* sync, void* stream);
int san_bn_act_bwd_sync_words(int n, int c, int hw);
int san_bn_act_bwd(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc, const float* sh,
                   float slope, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dy, int d_ctot,
                   int d_coff, void* amax, int n, int c, int hw, void* sync, void* stream);
int san_act_bwd_cluster_set_tuning(int on, int min_hw, int v, int bn_on);
int san_conv_bf16x3_dgrad_amax(const float* dy, int dy_ctot, int dy_coff, int cin, const void* w_packed, float* dx, int dx_ctot,
                               int dx_coff, int cout, const void* amax, int n, int h, int w, int ks, void* ws, size_t ws_bytes,
                               void* stream);
int san_conv2d_wgrad_bf16x3_amax(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                 float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                                 void* scratch, const void* dy_amax, int n, int h, int w, void* stream);
int san_conv1x1_wgrad_bf16x3_amax(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                  float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                                  int transposed, void* scratch, const void* dy_amax, int n, int h, int w, void* stream);

/* Deferred weight-gradient reductions (training step: ~320 reduction launches become ~8).  san_wgrad_defer(1): the bf16x3 /
 * fp16-part weight-gradient entry points above launch only their main kernel and queue the fixed-order reduction of their
 * partial tiles (so `scratch` must stay untouched and be private to that call until the flush); san_wgrad_defer_flush(stream)
 * launches the queued reductions, up to 48 layers per launch.  Flush before queueing a second gradient of the same dw.
 * Results are bit-identical to the immediate form.  san_wgrad_defer returns the previous setting. */
int san_wgrad_defer(int on);
int san_wgrad_defer_pending(void);
int san_wgrad_defer_flush(void* stream);

/* ------------------------------------------- image-domain cascade boundary */

/* out = (i)fft along H only (ortho: scale 1/sqrt(h)) of interleaved complex [planes, h, w].  k0x = ifft_y(k0) is the
 * data term of san_dc_rows; x0 = ifft_x(k0x) = ifft2(k0). */
int san_fft_cols(const float* in, float* out, int planes, int h, int w, int inverse, void* stream);

/* One cascade boundary in the IMAGE domain.  The soft data consistency's mask depends on kx only, so with x = ifft2(k)
 * the reference's  k' = k - dc_w * where(mask, k - k0, 0) - fft2(r * S)  followed by  m' = sum_c ifft2(k')_c conj(S_c)
 * (varnet.py:508-530 and :511-512 of the next cascade) is row-local:
 *     D  = ifft_x( mask (fft_x(x) - k0x) )        x_out = x - dc_w D - r S        m_out = sum_c conj(S_c) x_out_c
 * x, sens, k0x, x_out, dk_*: interleaved complex [n, c, h, w]; r_planar [n, 2, h, w]; m_out planar into channels 0, 1
 * of [n, m_ctot, h, w]; mask fp32 [w]; dc_w one device float.  x_out may alias x.  k0x / r_planar / x_out / m_out may
 * be NULL (term absent / result not wanted).  dk_out (optional): mask (fft_x(x) - k0x), kept for the backward pass.
 * backward != 0 (the DC term is self-adjoint; k0x, r_planar, dk_out must be NULL):
 *     x_out = g - dc_w ifft_x(mask fft_x(g)),   m_out = - sum_c conj(S_c) g_c   (gradient wrt the regulariser output),
 *     dcw_part[san_dc_rows_partials(n, c, h, w)] = per-workgroup partials of Re sum conj(fft_x(g)) dk_in, whose total is
 *     -dL/d(dc_w). */
int san_dc_rows_partials(int n, int c, int h, int w);
int san_dc_rows(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, const float* dk_in,
                float* dcw_part, int backward, int n, int c, int h, int w, void* stream);
/* Round 6: the forward form that also emits the statistics the NEXT cascade's NormUnet normalises its input with
 * (NormUnet.norm, varnet.py:262-273: mean / std of the two planes of m_out): m_stats [n][2][san_dc_rows_stat_tiles(n, c, h, w)][3] =
 * (count, mean, M2) per workgroup and plane, merged by san_norm_finalize like san_plane_stats' records.  A tile count of 0 means the
 * shape's kernel does not emit them (only the 320-wide kernel does): run san_plane_stats on m_out instead. */
int san_dc_rows_stat_tiles(int n, int c, int h, int w);
int san_dc_rows_stats(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                      const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, float* m_stats, int n, int c,
                      int h, int w, void* stream);

/* -------------------------------------------------------- warp and losses */

/* out[n,c,i,j] = bilinear sample of img[n,c] at identity_grid(i,j) + offset[n,:,i,j]
 * (zeros padding, align_corners=False).  offset is NCHW [n, 2, h, w] (channel 0
 * = x, 1 = y, normalised units) = the alignment head's raw output, i.e. the
 * reference's NHWC `offset` before its permute; grid_out (optional, NHWC
 * [n,h,w,2]) receives identity + offset.  padding: 0 zeros, 1 reflection.
 * Replaces affine_grid + add + F.grid_sample (cross.py:24-34; augment.py:60). */
int san_warp_fwd(const float* img, const float* offset, float* out, float* grid_out,
                 int n, int c, int h, int w, int padding, void* stream);

/* Generic sampler: explicit NHWC grid [n, ho, wo, 2]. */
int san_grid_sample_fwd(const float* img, const float* grid, float* out,
                        int n, int c, int h, int w, int ho, int wo, int padding, void* stream);
/* Its gradient wrt the IMAGE (zeros padding): gimg [n,c,h,w] = scatter of g [n,c,ho,wo] over the four texels each output
 * read (gimg is zeroed here; float atomics -- the only ones in the library; the training step never asks for this,
 * autograd callers of SpatialTransformer.warp with img.requires_grad do; cross.py:32-34). */
int san_grid_sample_bwd_img(const float* grid, const float* g, float* gimg,
                            int n, int c, int h, int w, int ho, int wo, void* stream);
/* The same gradient without float atomics (round 5; what autograd._WarpFn calls): contributions are rounded to 64-bit fixed
 * point (2^-40 of max |g|) and summed with integer atomics, so the result is independent of the arrival order -- bit-reproducible.
 * work: san_grid_sample_bwd_img_work_bytes(n, c, h, w) bytes of scratch.  cross.py:32-34 (F.grid_sample's autograd). */
size_t san_grid_sample_bwd_img_work_bytes(int n, int c, int h, int w);
int san_grid_sample_bwd_img_det(const float* grid, const float* g, float* gimg, long long* work,
                                int n, int c, int h, int w, int ho, int wo, void* stream);

/* The same sampler on interleaved complex planes (img, out: [n,c,h,w] float2): real and imaginary
 * parts sampled with one grid, as augment.py:62-63 does with two grid_sample calls. */
int san_grid_sample_complex_fwd(const float* img, const float* grid, float* out,
                                int n, int c, int h, int w, int ho, int wo, int padding, void* stream);

/* Augmentation sampling grid (augment.py:7-48): grid [n,h,w,2] = affine_grid(affine [n,2,3],
 * align_corners=False) (+ bicubic (A=-0.75, align_corners=False) upsample of ctrl [n,2,cg,cg] when
 * ctrl != NULL; the reference uses cg = 9 and ctrl = (rand-0.5)*2/50).  Feed it to the samplers
 * above with padding = 1 (reflection). */
int san_augment_grid(const float* affine, const float* ctrl, float* grid, int n, int h, int w, int cg, void* stream);

/* loss[0] = 1 - mean SSIM(x, y), 7x7 uniform valid window (ssimloss.py:11-40).
 * ws: fp32 [san_loss_workspace_floats(n, h, w)]. */
size_t san_loss_workspace_floats(int n, int h, int w);
int san_ssim_loss_fwd(const float* x, const float* y, float* loss, int n, int h, int w,
                      float* ws, void* stream);

/* loss[0] = -mean(cross^2/(Ivar*Jvar+1e-5)), win x win zero padded box sums
 * (lnccloss.py:7-56). */
int san_lncc_loss_fwd(const float* i, const float* j, float* loss, int n, int h, int w, int win,
                      float* ws, void* stream);

/* Backward of san_lncc_loss_fwd (replaces autograd through lnccloss.py:7-56): gi = gscale [* gscale_dev[0]] * d loss / d i,
 * gj alike (either may be NULL; accumulate != 0 adds to what they hold).  Two launches (window coefficients, LDS-tiled
 * gather), no float atomics.  ws: fp32 [san_lncc_bwd_workspace_floats(n, h, w)] = five coefficient planes. */
size_t san_lncc_bwd_workspace_floats(int n, int h, int w);
int san_lncc_loss_bwd(const float* i, const float* j, float* gi, float* gj, float gscale, const float* gscale_dev,
                      int accumulate, int n, int h, int w, int win, float* ws, void* stream);

/* y [planes, h/2, w/2] = avg_pool2(conv2d(x, kern[ksize x ksize], zero pad ksize/2)):
 * the Gaussian(sigma=3, 13 taps) + 2x average-pool step between the scales of
 * ms_lncc_loss (lnccloss.py:58-65, miloss.py:6-24).  kern: device fp32 [ksize*ksize]. */
int san_smooth_pool_fwd(const float* x, const float* kern, float* y, int planes, int h, int w, int ksize,
                        void* stream);
/* Its adjoint: gx [planes, h, w] (+)= d/dx of <gy, smooth_pool(x)> (the chain between the scales of ms_lncc_loss,
 * lnccloss.py:58-65). */
int san_smooth_pool_bwd(const float* gy, const float* kern, float* gx, int accumulate, int planes, int h, int w, int ksize,
                        void* stream);

/* loss[0] = (mean(dW^2) + mean(dH^2))/2 of an offset field given as NCHW
 * [n, 2, h, w] (model.py:21-28 on the permuted view). */
int san_gradient_loss_fwd(const float* offset, float* loss, int n, int h, int w, float* ws, void* stream);

/* ---- 3x3 convolution on the bf16 matrix cores with fp32-level accuracy ("bf16x3": operands split into
 * three bf16 parts, six products accumulated in fp32; csrc/san_conv_bf16.hip).  Same contract as
 * san_conv2d_fwd for ks = 3 without the output affine; used where san_conv_bf16x3_eligible() says so
 * (channel counts that fill 16-wide MFMA tiles).  Weights: san_conv_bf16x3_pack into a buffer of
 * san_conv_bf16x3_packed_bytes(cout, cin) bytes -- mode 0: forward weight [cout,cin,3,3]; mode 2: data
 * gradient, where (cout, cin) are those of the data-gradient convolution (= forward cin, cout) and w is
 * still the forward weight.  Statistics tiles: san_conv_bf16x3_stat_tiles(n, h, w) per (n, channel).
 * san_conv_bf16x3_pack_job / _pack_batch: the batched form (host table entry, one launch), as for
 * san_conv_pack_job / san_conv_pack_batch. */
int san_conv_bf16x3_eligible(int cin, int cout, int h, int w, int ks);
/* Persistent ("stream") form of the fp16-part 3x3 convolution (csrc/san_conv_stream.hip, round 4): workgroups walk a sequence of
 * 32 x 8 tiles with the packed weights resident in LDS and the next tile's input requested a whole K-loop ahead.  Taken
 * automatically by san_conv2d_bf16x3_fwd / _fwd_ws / _fwd_ws_in / san_conv_bf16x3_dgrad_amax (same arguments, same statistics
 * records) where san_conv_stream_eligible() says 1: fp16-format weights, h % 8 == 0, w % 32 == 0, h * w >= 160^2, cin <= 96,
 * cout in {16, 18, 32, 36, 48} (round 6: also 2 / 3 -- the data gradient of a cascade's input convolution, varnet.py:139-146 with
 * in_chans = 2 / 3 -- as a partial channel block alone; the caller asks san_conv_stream_eligible first: the one-tile kernel does not
 * take such a layer).  Replaces nn.Conv2d(3x3, padding 1) of varnet.py:139-146 and its autograd data gradient on the
 * 320^2 / 160^2 levels.  san_conv_stream_set_tuning(0) sends every layer to the one-tile-per-workgroup kernel (tests). */
int san_conv_stream_eligible(int n, int h, int w, int cin, int cout, int x_ctot);
int san_conv_stream_set_tuning(int on);
int san_conv_bf16x3_debug_timeline(void* buf);   /* tuning: per-workgroup clock marks of the next launches (8 x u64 each; NULL = off), scratch/conv_timeline.py */
int san_conv_bf16x3_set_tuning(int wd, int mb);   /* tests / tuning: weights-direct form (-1 auto, 0, 1), channel blocks per workgroup (-1 auto, 2..5) */
int san_conv_bf16x3_tile_set_tuning(int nbw, int wd_cold);   /* tests / tuning (round 6): blocks per wave of the full-width tiles (0 auto, 3, 4); LDS-staged weights for launches of <= 256 workgroups (1 default, 0 off) */
size_t san_conv_bf16x3_packed_bytes(int cout, int cin);
int san_conv_bf16x3_stat_tiles(int n, int h, int w);
/* Statistics tiles of the 3x3 entry points below for THIS layer (round 6): the tile geometry of a 3x3 launch may depend on the batch and
 * the channel counts (short full-width tiles where the default geometry leaves compute units idle), f16_format = 1 when the packed
 * weights given to the launch hold two fp16 parts (mode + 16).  Same layout as san_conv_bf16x3_stat_tiles, which stays the count of the
 * 1x1 / transposed forms.  (nn.Conv2d(3x3) + InstanceNorm2d statistics, varnet.py:139-146) */
int san_conv3x3_bf16x3_stat_tiles(int n, int h, int w, int cin, int cout, int f16_format);
int san_conv_bf16x3_pack(const float* w, void* packed, int cout, int cin, int mode, void* stream);
int san_conv_bf16x3_pack_job(long long* job8, const float* w, void* packed, int cout, int cin, int mode);
int san_conv_bf16x3_pack_batch(const long long* jobs_dev, int njobs, void* stream);
/* NOTE (round 5): the batched forms write only the units that can be non-zero; an image must have been zero-filled (or packed once by
 * san_conv_bf16x3_pack_ks, which writes everything) before its first batched pack.
 * The same with `blocks` (1..64) workgroups per job; fp8 != 0: run the per-tensor fp8 scale pass first (needed only when a job of
 * the run has the fp8 format).  Callers sort their jobs by size and pack each size class with a fitting grid. */
int san_conv_bf16x3_pack_batch_grid(const long long* jobs_dev, int njobs, int blocks, int fp8, void* stream);
int san_conv2d_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin,
                          const float* in_scale, const float* in_shift, float in_slope,
                          const void* w_packed, const float* bias,
                          float* y, int y_ctot, int y_coff, int cout, float* part_stats,
                          int n, int h, int w, void* stream);

/* Split-K form for deep-K layers on small images (288 -> 288 @20^2: 64 tiles for 256 CUs and 12 serial
 * chunks per tile): up to 4 workgroups share an output tile, each over a range of the input channels,
 * and a second launch adds their partial outputs in a fixed order (deterministic), applies the bias,
 * stores and takes the statistics.  ws: san_conv_bf16x3_ws_bytes(...) bytes (0 = this layer is not
 * split; then the plain entry point does the same work). */
size_t san_conv_bf16x3_ws_bytes(int n, int h, int w, int cin, int cout, int ks);
int san_conv2d_bf16x3_fwd_ws(const float* x, int x_ctot, int x_coff, int cin,
                             const float* in_scale, const float* in_shift, float in_slope,
                             const void* w_packed, const float* bias,
                             float* y, int y_ctot, int y_coff, int cout, float* part_stats,
                             int n, int h, int w, void* ws, size_t ws_bytes, void* stream);
/* The same for a layer followed by InstanceNorm2d (varnet.py:141,144): when the launch is split over K, its second pass sees
 * every (sample, channel) plane whole and writes the lazy affine itself -- scale / shift views [n, sc_ctot] at channel offset
 * sc_coff, exactly san_norm_finalize(SAN_NORM_INSTANCE)'s values -- and sets *finalised = 1 (host int; part_stats untouched):
 * the caller skips its san_norm_finalize launch.  Otherwise *finalised = 0 and part_stats holds the partials as usual. */
int san_conv2d_bf16x3_fwd_ws_in(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                                int cout, float* part_stats, int n, int h, int w, void* ws, size_t ws_bytes, float* scale,
                                float* shift, int sc_ctot, int sc_coff, float eps, int* finalised, void* stream);

/* Per-tensor power-of-two scale of fp16-format weight images (round 6): on = 1 / 0 switches it for images packed FROM NOW ON
 * (returns the previous setting; on < 0 only queries).  With it the packed image holds w * S_w (max |w| S_w in [2^13, 2^14)), {S_w,
 * 1 / S_w} sit behind the image and the convolution kernels multiply their accumulators by 1 / S_w: weights of any magnitude keep 22
 * mantissa bits (unscaled: fp16's range as it is -- a 0.1-sized weight loses 3e-7 to the denormal floor of its second part).  Off
 * by default (a max pass over every weight per optimiser step); the caller initialises the 8 bytes behind a new image to {1.f, 1.f}.
 * Replaces nothing in the reference (its weights are fp32, varnet.py:139-146); it is part of this library's operand format. */
int san_conv_f16_wscale_enable(int on);

/* The same kernel as a 1x1 convolution (the alignment net's 1x1 layers, unet.py:64-77; the data gradient of
 * the transposed convolutions): weights packed with the _ks entry points (ks = 1 or 3; the plain ones are
 * ks = 3), statistics tiles as san_conv_bf16x3_stat_tiles. */
size_t san_conv_bf16x3_packed_bytes_ks(int cout, int cin, int ks);
int san_conv_bf16x3_pack_ks(const float* w, void* packed, int cout, int cin, int mode, int ks, void* stream);
int san_conv_bf16x3_pack_job_ks(long long* job8, const float* w, void* packed, int cout, int cin, int mode, int ks);
int san_conv1x1_bf16x3_eligible(int cin, int cout, int h, int w);
int san_conv1x1_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin,
                           const float* in_scale, const float* in_shift, float in_slope,
                           const void* w_packed, const float* bias,
                           float* y, int y_ctot, int y_coff, int cout, float* part_stats,
                           int n, int h, int w, void* stream);

/* ConvTranspose2d 2x2 stride 2 (varnet.py:159-192) on the same kernel: 1x1 convolution to 4*cout virtual
 * channels + pixel shuffle in the epilogue.  x [n, x_ctot, h, w] -> y [n, y_ctot, 2h, 2w] (8-byte aligned);
 * weights: the [Cin, Cout, 2, 2] tensor packed with san_conv_bf16x3_pack_ks(w, packed, 4*cout, cin, 2, 1);
 * part_stats [n, cout, 4 * san_conv_bf16x3_stat_tiles(n, h, w), 3] or NULL. */
int san_tconv2x2_bf16x3_eligible(int cin, int cout, int h, int w);
int san_tconv2x2_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin,
                            const float* in_scale, const float* in_shift, float in_slope,
                            const void* w_packed, float* y, int y_ctot, int y_coff, int cout,
                            float* part_stats, int n, int h, int w, void* stream);

/* 3x3 weight gradient on the bf16 matrix cores, fp32-level accuracy (csrc/san_wgrad_bf16.hip): same
 * contract as san_conv2d_wgrad for ks = 3 (backward of F.conv2d at varnet.py:140,143 / unet.py:119-140
 * w.r.t. the weight).  _supported(): the kernel can run the layer; _eligible(): it is also the faster
 * choice (what the dispatcher asks).  scratch: 16-byte aligned
 * device buffer of san_conv_wgrad_bf16x3_scratch_bytes(n,h,w,cin,cout) bytes (split bf16 planes of x
 * and dy + per-workgroup partial tiles); nothing is kept in it after the call returns. */
int san_conv_wgrad_bf16x3_supported(int n, int h, int w, int cin, int cout, int ks);
int san_conv_wgrad_bf16x3_set_mode(int mode);   /* -1 automatic, 0 split-plane form, 1 direct form (tests / tuning) */
int san_conv_wgrad_bf16x3_eligible(int n, int h, int w, int cin, int cout, int ks);
size_t san_conv_wgrad_bf16x3_scratch_bytes(int n, int h, int w, int cin, int cout);
int san_conv2d_wgrad_bf16x3(const float* x, int x_ctot, int x_coff, int cin,
                            const float* in_scale, const float* in_shift, float in_slope,
                            const float* dy, int dy_ctot, int dy_coff, int cout,
                            float* dw, int accumulate, void* scratch,
                            int n, int h, int w, void* stream);

/* 1x1 weight gradient on the bf16 matrix cores (same file): san_conv2d_wgrad's contract for ks = 1 -- the
 * 2x2 transposed convolutions (varnet.py:159-192, run as 1x1 convolutions to 4 cout + pixel shuffle) and the
 * alignment net's 1x1 layers (unet.py:119-140).  Needs h*w % 4 == 0 and 16-byte aligned x / dy.
 * transposed = 1 writes dw as [cin][cout] (the ConvTranspose2d weight layout [Cin, Cout*4]) instead of [cout][cin]. */
int san_conv1x1_wgrad_bf16x3_eligible(int n, int h, int w, int cin, int cout);
size_t san_conv1x1_wgrad_bf16x3_scratch_bytes(int n, int h, int w, int cin, int cout);
int san_conv1x1_wgrad_bf16x3(const float* x, int x_ctot, int x_coff, int cin,
                             const float* in_scale, const float* in_shift, float in_slope,
                             const float* dy, int dy_ctot, int dy_coff, int cout,
                             float* dw, int accumulate, int transposed, void* scratch,
                             int n, int h, int w, void* stream);

/* Batched weight packing for training, where every weight changes every step: san_conv_pack_job
 * fills one HOST table entry (8 x int64) for a weight/packed-buffer pair -- mode 0: Conv2d forward
 * (san_conv_pack_weights_fwd), 1: ConvTranspose2d 2x2 (san_conv_pack_weights transposed), 2: Conv2d
 * data gradient (san_conv_pack_weights_dgrad); packed buffer sizes as for those calls -- and
 * san_conv_pack_batch runs a device copy of the table [njobs][8] in ONE launch. */
int san_conv_pack_job(long long* job8, const float* w, float* packed, int cout, int cin, int ks, int mode);
int san_conv_pack_batch(const long long* jobs_dev, int njobs, void* stream);

/* Validation metrics of an image batch (metrics.py:23-35, 55-69), one launch: out [n][4] doubles =
 * per image { sum (gt-pred)^2, sum |gt-pred|, sum gt^2, mutual information from a bins x bins joint
 * histogram over [0,1]^2 }.  mse / mae / nmse / psnr / mi follow on the host from these sums. */
int san_image_metrics(const float* gt, const float* pred, double* out, int n, int hw, int bins, void* stream);

/* One AdamW step over flat fp32 buffers (replaces torch.optim.AdamW(lr, weight_decay) over every
 * parameter tensor, model.py:72-87; decoupled weight decay, no amsgrad):
 *   g' = grad_scale*g;  p *= 1 - lr*wd;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).   step counts from 1.
 * All four buffers: count floats, 16-byte aligned. */
int san_adamw_step(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same step with the count kept in device memory (one int64: steps taken so far; the call uses *step_dev + 1 and
 * then advances it), so that a captured hipGraph of the training step replays correctly. */
int san_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                       float beta2, float eps, float weight_decay, long long* step_dev, float grad_scale, void* stream);
/* ... and with (lr, weight_decay, grad_scale) read from device memory too when hyper_dev (fp32 [3]) is given: a captured
 * step then follows a learning-rate schedule (the caller refreshes the three floats between replays; param_groups['lr'] of
 * torch.optim.AdamW, model.py:72-81).  hyper_dev == NULL: the scalar arguments, as san_adamw_step_dev. */
int san_adamw_step_hyper(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                         float beta2, float eps, float weight_decay, long long* step_dev, float grad_scale,
                         const float* hyper_dev, void* stream);

/* ------------------------------------------------------------- gradient exchange on RCCL (round 5) */

/* The data-parallel step's all-reduces as C-ABI calls (csrc/san_rccl.cpp), so that a recorded step's tape walks them like kernel
 * launches.  The reference has no data parallelism; these replace torch.distributed.all_reduce on the flat
 * gradient buffers inside CSModel.update().  RCCL is not linked: san_rccl_load binds the shared object at `path` (the copy the
 * process already has -- torch's; NULL = "librccl.so.1") and returns its version code in *version (may be NULL).
 * san_rccl_unique_id fills 128 bytes on rank 0; every rank passes them to san_rccl_comm_init (collective, blocking; the calling
 * thread's HIP device is the rank's GPU), which returns a communicator handle.  san_rccl_allreduce_sum_f32 sums buf[0 .. count)
 * in place over the ranks, stream-ordered on `stream`. */
int san_rccl_load(const char* path, int* version);
int san_rccl_unique_id(void* id128);
int san_rccl_comm_init(const void* id128, int world, int rank, int* handle);
int san_rccl_allreduce_sum_f32(int handle, float* buf, size_t count, void* stream);
int san_rccl_comm_destroy(int handle);
/* The same sum as two collectives (round 6, SAN_GRAD_EXCHANGE=rs_ag): rank r receives the sum of every rank's chunk r of `send`
 * (world chunks of recv_count floats) in recv -- ncclReduceScatter -- and san_rccl_allgather_f32 hands every rank's send_count
 * floats to everybody (recv = world * send_count floats, rank order).  On a fully connected xGMI node both phases use all seven
 * links at once; in-place forms: recv == send + rank * recv_count / send == recv + rank * send_count. */
int san_rccl_reduce_scatter_sum_f32(int handle, const float* send, float* recv, size_t recv_count, void* stream);
int san_rccl_allgather_f32(int handle, const float* send, float* recv, size_t send_count, void* stream);

/* ------------------------------------------------------------- recorded steps */

/* Walks a "tape" of recorded calls on the host, in order (csrc/san_replay.cpp; the host-side mirror of the reference's train loop
 * body, train.py:212-217, once CSModel.update() has recorded it -- the reference itself has no such layer: its step is Python).
 * tape: host array of n_words 64-bit words; an entry = one head word  code | flags << 16 | nargs << 24  followed by nargs
 * argument words (pointers / integers by value, float / double as bit patterns).  code < 0x8000: the index, in the order of
 * this header, of an int-returning function declared here, called with those arguments; 0x8000: hipEventRecord(event, stream);
 * 0x8001: hipStreamWaitEvent(stream, event).  flags bit 0: the call's return value is not an error code; bit 1: a weight-packing
 * launch, skipped when skip_packs != 0.  Stops at the first failing call: returns its code and stores the index of the entry's
 * head word in *failed_word (host, may be NULL).  Entries are not validated beyond their length: a tape is built by
 * spatialalignmentnetwork_amd/model.py from calls that already ran once. */
int san_replay_run(const void* tape, size_t n_words, int skip_packs, long long* failed_word);

/* Events for the stream hand-offs of a recorded step (round 6; csrc/san_core.cpp).  san_event_create makes a hipEvent_t on
 * `device` (the calling thread's current device is restored) without timing; light != 0 also drops the system-scope fence
 * (hipEventDisableSystemFence: ~1.2 instead of ~3.2 us per record on the recording stream) -- for dependencies between streams
 * of ONE GPU only; anything a peer GPU or the host observes (the gradient exchange's communication stream) takes light = 0.
 * *event (host) receives the handle; san_event_destroy releases it.  These replace torch.cuda.Event objects inside replay tapes
 * (train.py:212-217 has no counterpart: the reference's step is eager Python). */
int san_event_create(int device, int light, void** event);
int san_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* SAN_HIP_H */
