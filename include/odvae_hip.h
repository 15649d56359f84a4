/* libodvae_hip.so -- C ABI of the MI355X (gfx950) kernels behind the OD-VAE autoencoder training step.
 *
 * The reference (tanushreebanerjee/generative-detection) has no FFI of its own: its hot path calls
 * torch ops from Python ([UPSTREAM] ldm/modules/diffusionmodules/model.py through
 * src/modules/autoencodermodules/feat_encoder.py:4, feat_decoder.py:4; src/models/autoencoder.py;
 * src/util/distributions.py; src/modules/losses/contperceptual.py).  Each entry point below names the
 * torch call it stands in for.  The host side binds these with ctypes
 * (generative-detection_amd/lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP global memory) unless it says "host";
 *   - tensors are f32, activations are NHWC ([N][H][W][C], channels contiguous), conv weights OIHW;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work, they
 *     never synchronise the device and never allocate;
 *   - scratch comes from the caller: `workspace`/`workspace_bytes`, sized by the *_workspace_bytes query;
 *   - return value: 0 on success, ODVAE_ERR_* otherwise; odvae_last_error() holds the message.
 */
#ifndef ODVAE_HIP_H
#define ODVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODVAE_ABI_VERSION 4            /* odvae_abi_version() of a library built from this header */
#define ODVAE_OK 0
#define ODVAE_ERR_ARG 1
#define ODVAE_ERR_WORKSPACE 2
#define ODVAE_ERR_HIP 3

/* ---- runtime.cpp ------------------------------------------------------------------------------- */
const char* odvae_last_error(void);   /* host string, thread-local */
int odvae_abi_version(void);          /* == ODVAE_ABI_VERSION of this header; bumped whenever the exported surface changes */
const char* odvae_target_arch(void);  /* "gfx950" */

/* ---- gemm_f32.hip: torch.nn.Conv2d(k=1) / torch.bmm ---------------------------------------------
 * C[b] = alpha * op(A[b]) * op(B[b]) (+ bias[col]) (+ residual[b]); row-major;
 * transA=0: A is [M][K]; transA=1: A is stored [K][M]; transB=0: B is [K][N]; transB=1: B is stored [N][K].
 * Stands in for the 1x1 convolutions (nin_shortcut, AttnBlock q/k/v/proj_out, quant_conv_obj/pose,
 * post_quant_conv: src/models/autoencoder.py:88-90,179-180) and AttnBlock's torch.bmm products. */
size_t odvae_gemm_f32_workspace_bytes(int M, int N, int K, int batch);
/* operand staging of the two GEMM entry points: -1 per shape (default: LDS-DMA where A is row-contiguous), 0 through registers,
 * 1 by LDS-DMA (`buffer_load ... lds`, two stages of 32-wide steps), 2 the same with 16-wide steps (four blocks per CU); identical
 * results; returns the previous setting.  Env preset: ODVAE_GEMM_DMA. */
int odvae_gemm_select_staging(int mode);
int odvae_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                   const float* A, int lda, int64_t strideA,
                   const float* B, int ldb, int64_t strideB,
                   float* C, int ldc, int64_t strideC,
                   const float* bias, const float* residual, int batch,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- conv3x3_f32.hip: torch.nn.Conv2d(k=3) in ResnetBlock / Downsample / Upsample / conv_in / conv_out
 * Weight packs: reduction axis padded to odvae_conv3x3_pack_reduce_pad(), output axis to
 * odvae_conv3x3_pack_out_pad(); a pack has odvae_conv3x3_pack_floats(c_reduce, c_out) floats.
 * odvae_conv3x3_pack_f32 turns OIHW into the forward pack (reduce=Cin,out=Cout) and/or the
 * data-gradient pack (reduce=Cout,out=Cin, taps flipped); either output may be NULL. */
int odvae_conv3x3_pack_reduce_pad(int c_reduce);
int odvae_conv3x3_pack_out_pad(int c_out);
size_t odvae_conv3x3_pack_floats(int c_reduce, int c_out);
int odvae_conv3x3_pack_f32(const float* w_oihw, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream);
/* 16-tap packs of an Upsample conv (modes 5 / 6): odvae_conv3x3_up_pack_floats(c_reduce, c_out) floats each */
size_t odvae_conv3x3_up_pack_floats(int c_reduce, int c_out);
int odvae_conv3x3_pack_up_f32(const float* w_oihw, int Cout, int Cin, float* fwd16, float* dgrad16, void* stream);
/* mode 0: stride 1 pad 1 (ResnetBlock.conv1/conv2, conv_in, conv_out)
 * mode 1: F.pad(x,(0,1,0,1)) + stride 2 pad 0 (Downsample.forward)      Ho = Hi/2
 * mode 2: F.interpolate(scale 2, nearest) + stride 1 pad 1 (Upsample)   Ho = 2*Hi
 * mode 3: data gradient of mode 1 (x = dy, y = dx, wpk = dgrad pack)    Ho = 2*Hi
 * mode 5: mode 2 evaluated per output parity class on the low-res input with the taps that hit the same input pixel
 *         pre-summed (wpk = fwd16 of odvae_conv3x3_pack_up_f32; 16 instead of 36 tap-products per input pixel)  Ho = 2*Hi
 * mode 6: data gradient of mode 2 / 5 = 4x4-tap stride-2 pad-1 conv over dy (x = dy, y = dx, wpk = dgrad16)   Ho = Hi/2
 * The data gradient of mode 0 is mode 0 with the dgrad pack. */
int odvae_conv3x3_f32(int mode, const float* x, int N, int Hi, int Wi, int Cin,
                      const float* wpk, int Cout, const float* bias, const float* residual,
                      float* y, int Ho, int Wo, int act /* 0 none, 1 ReLU */, void* stream);

/* ---- conv3x3_wino_f32.hip: the stride-1 pad-1 3x3 convolution (mode 0 above and its data gradient) by Winograd
 * F(2x2, 3x3) with fused input / output transforms: f32 throughout, 2.25x fewer multiply-adds, summation order differs
 * from the direct form (~1e-6 relative).  Packs: U = G g G^T per (ci, co), [16][reduce_pad/4][out_pad][4] floats;
 * fwd_pack (reduce = Cin) and/or dgrad_pack (reduce = Cout, taps flipped), either may be NULL.  Needs even H, W and
 * Cin % 4 == 0. */
int odvae_conv3x3_wino_reduce_pad(int c_reduce);
int odvae_conv3x3_wino_out_pad(int c_out);
size_t odvae_conv3x3_wino_pack_floats(int c_reduce, int c_out);
int odvae_conv3x3_pack_wino_f32(const float* w_oihw, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream);
int odvae_conv3x3_wino_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                           const float* bias, const float* residual, float* y, int act /* 0 none, 1 ReLU */, void* stream);

/* ---- conv3x3_wino4_f32.hip: the same convolution by Winograd F(4x4, 3x3): 4x fewer multiply-adds than the direct form
 * (1.78x fewer than F(2x2)), f32 throughout; the wider interpolation points cost accuracy (max deviation from an f64 direct
 * convolution ~4e-6 of max|y| at Cin = 128, F(2x2) / direct f32: 2e-7).  Packs: [36][reduce_pad/4][out_pad][4] floats.
 * Needs H, W in multiples of 4 and Cin % 8 == 0; `supported` names the shapes the step routes here (forward and data
 * gradient both qualify). */
int odvae_conv3x3_wino4_reduce_pad(int c_reduce);
int odvae_conv3x3_wino4_out_pad(int c_out);
size_t odvae_conv3x3_wino4_pack_floats(int c_reduce, int c_out);
int odvae_conv3x3_wino4_supported(int H, int W, int Cin, int Cout);
int odvae_conv3x3_pack_wino4_f32(const float* w_oihw, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream);
/* Batched packs: ONE launch for n weights (a training step repacks every conv weight after each optimizer step).  items = device array of
 * n records {const float* w; void* fwd_pack; void* dgrad_pack; int Cout, Cin, taps, reserved;} (40 bytes, natural alignment; either pack
 * may be NULL), channel counts that need no padding (c == *_reduce_pad(c) == *_out_pad(c)).  (The bf16 packs stay per weight: batching them
 * was measured slower -- a pack made right before its conv is still in L2 when the conv reads it.) */
int odvae_conv3x3_pack_wino_batch(const void* items, int n, void* stream);
int odvae_conv3x3_pack_wino4_batch(const void* items, int n, void* stream);

int odvae_conv3x3_wino4_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                            const float* bias, const float* residual, float* y, int act /* must be 0: no fused activation */, void* stream);
/* The same, and the output transform also leaves the GroupNorm statistics of y for the layer that reads it (GroupNorm row:
 * "statistics from the producing conv's epilogue"): gn_partial [N][odvae_conv3x3_wino4_stats_chunks(H, W)][gn_groups][2] = (sum, sum of
 * squares) of y per output tile and channel group, every slot written by exactly one block (deterministic).  Cout / gn_groups must be a
 * power of two <= 32.  Feed it to odvae_groupnorm_fwd_partials_f32. */
int odvae_conv3x3_wino4_stats_chunks(int H, int W);
int odvae_conv3x3_wino4_stats_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                                  const float* bias, const float* residual, float* y, float* gn_partial, int gn_groups, void* stream);
/* The Upsample conv (nearest 2x, then 3x3 stride 1 pad 1) on the same kernel: x [N][H/2][W/2][Cin] low resolution, y [N][H][W][Cout];
 * upk = the forward pack of odvae_conv3x3_pack_wino4_f32; gn_partial / gn_groups as above or NULL / 0.  Shapes:
 * odvae_conv3x3_wino4_supported(H, W, Cin, Cout) on the OUTPUT size. */
int odvae_conv3x3_wino4_up_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout, const float* bias,
                               const float* residual, float* y, float* gn_partial, int gn_groups, void* stream);
/* Data gradient of the Upsample conv w.r.t. its LOW-resolution input ([UPSTREAM] Upsample.forward: interpolate(2x, nearest), conv): x = the
 * conv's dy [N][H][W][Cin], upk = its data-gradient pack, y [N][H/2][W/2][Cout] = the 2x2 sums of the full-resolution gradient, formed in the
 * output transform (that gradient is never stored; odvae_upsample2x_bwd_f32 is not needed behind it). */
int odvae_conv3x3_wino4_pool_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout, float* y, void* stream);
/* Data gradient of a conv whose INPUT was a = swish(GroupNorm(gn_x)) ([UPSTREAM] ResnetBlock: `h = conv(nonlinearity(norm(h)))`), and the
 * first pass of that GroupNorm's backward out of the same output transform: x = the conv's dy [N][H][W][Cin], upk = its data-gradient
 * pack, y = da [N][H][W][Cout]; gn_partial [N][odvae_conv3x3_wino4_stats_chunks(H, W)][2][Cout] = per output tile and channel
 * (sum du * xhat, sum du), du = da * swish'(xhat * gamma + beta) -- every slot written by exactly one block.  Feed it to
 * odvae_groupnorm_bwd_partials_f32: da and gn_x are then not streamed a second time for the sums. */
int odvae_conv3x3_wino4_gnbwd_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout, float* y,
                                  const float* gn_x, const float* gn_mean, const float* gn_rstd, const float* gn_gamma, const float* gn_beta,
                                  int gn_groups, float* gn_partial, void* stream);

/* ---- conv3x3_wgrad_f32.hip: weight/bias gradient autograd computes for those convolutions (modes 0-2; mode 5 =
 * mode 2 accumulated per output parity class, 16 instead of 36 tap-products per input pixel, same dw)
 * dw is OIHW [Cout][Cin][3][3], overwritten; dbias [Cout] or NULL. */
size_t odvae_conv3x3_wgrad_workspace_bytes(int mode, int N, int Ho, int Wo, int Cin, int Cout);
int odvae_conv3x3_wgrad_f32(int mode, const float* x, const float* dy, int N, int Hi, int Wi, int Cin,
                            int Ho, int Wo, int Cout, float* dw, float* dbias,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- conv3x3_wgrad_wino_f32.hip: the same weight/bias gradient for mode 0 (stride 1, pad 1) in the Winograd
 * F(2x2,3x3) domain: dU[xi] = sum over 2x2 tiles of (B^T d B)[xi]^T (A dY A^T)[xi], dw = G^T dU G -- 16 instead of 36
 * multiply-adds per tile.  Serves even H, W, channel counts in multiples of 128 and tensors below 4 GiB
 * (..._supported() == 1); everything else stays on odvae_conv3x3_wgrad_f32.  dw OIHW overwritten; dbias [Cout] or NULL.
 * Replaces autograd's weight gradient of F.conv2d(x, w, b, stride=1, padding=1) ([UPSTREAM] ldm ResnetBlock.conv1/conv2,
 * Encoder/Decoder convs; reference call sites src/modules/autoencodermodules/feat_encoder.py:2, feat_decoder.py:2). */
int odvae_conv3x3_wgrad_wino_supported(int N, int H, int W, int Cin, int Cout);
size_t odvae_conv3x3_wgrad_wino_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int odvae_conv3x3_wgrad_wino_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout,
                                 float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* ---- groupnorm.hip: Normalize = GroupNorm(32, C, eps=1e-6) followed by x*sigmoid(x) -----------------
 * x,y: [N][HW][C]; mean,rstd: [N][G]; swish: 0 identity, 1 x*sigmoid(x). */
size_t odvae_groupnorm_workspace_bytes(int N, int HW, int C, int G);
int odvae_groupnorm_fwd_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                            float eps, int swish, float* y, float* mean, float* rstd,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the same without the statistics pass: partial [N][chunks][G][2] = (sum, sum of squares) of x per chunk and channel group, as the kernel
 * that produced x left them (odvae_conv3x3_wino4_stats_f32); finalize (f64, fixed order) + apply, x read once */
int odvae_groupnorm_fwd_partials_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                                     float eps, int swish, float* y, float* mean, float* rstd,
                                     const float* partial, int chunks, void* stream);
/* the apply pass alone from a forward call's mean / rstd: y = act(GroupNorm(x)) re-made from x (the recompute of the "norm" checkpoint policy:
 * the conv that consumed y keeps (x, mean, rstd) instead of y); odvae_groupnorm_apply_bf16 is its bf16 twin */
int odvae_groupnorm_apply_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, int swish, float* y, void* stream);
int odvae_groupnorm_apply_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, int swish, void* y, void* stream);
int odvae_groupnorm_bwd_f32(const float* x, const float* dy, int N, int HW, int C, int G,
                            const float* gamma, const float* beta, const float* mean, const float* rstd, int swish,
                            float* dx, float* dgamma, float* dbeta, const float* dx_add,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the same with the sums of the first pass given: partial [N][chunks][2][C] as odvae_conv3x3_wino4_gnbwd_f32 leaves it; finalize (f64,
 * fixed order) + apply.  workspace: (N * 2 * C + N * G * 2) floats (odvae_groupnorm_workspace_bytes covers it). */
int odvae_groupnorm_bwd_partials_f32(const float* x, const float* dy, int N, int HW, int C, int G,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd, int swish,
                                     float* dx, float* dgamma, float* dbeta, const float* dx_add, const float* partial, int chunks,
                                     void* workspace, size_t workspace_bytes, void* stream);
/* backward form of odvae_groupnorm_bwd_f32: -1 (default) the read-once kernel where ONE block holds a (sample, 32-channel slab) in its
 * registers (HW <= 256; C % 32 == 0, whole groups per slab), reduce + apply (x and dy read twice) elsewhere; 0 always reduce + apply;
 * 1 the read-once kernel on every shape it takes (teams of ceil(HW / 256) resident blocks that meet at a per-item barrier in L2 --
 * measured slower than the two-kernel form on this chip for HW >= 1024, kept for A/B) or ODVAE_ERR_ARG.  Returns the previous setting. */
int odvae_groupnorm_select_backward(int mode);
/* team-barrier waits of the read-once kernels that gave up (0.2 s) since the library was loaded; synchronises the device.  Must be 0. */
int odvae_groupnorm_fused_timeouts(void);
/* both device-side health counters (synchronises the device; call where the host waits anyway: validation, checkpoint, end of fit):
 * *gn_timeouts != 0 -> a GroupNorm backward's team barrier gave up and its gradients are WRONG (stop the run); *attn_fallbacks = exact-softmax
 * fallbacks of the folded attention softmax taken so far (correct results, three extra passes each).  inject_* > 0: test hooks that bump the
 * counters first (inject_gn_timeouts < 0 clears that counter).  Mirrors nothing in the reference (its torch ops cannot fail this way); generative-detection_amd/trainer.py reads it. */
int odvae_device_health(int* gn_timeouts, int* attn_fallbacks, int inject_gn_timeouts, int inject_attn_fallbacks);
int odvae_attn_softmax_fallbacks(int add);
/* dx_add (nullable, same shape as x): a second gradient reaching x (the ResnetBlock / AttnBlock skip connection,
   [UPSTREAM] model.py `return x + h`), summed into dx in the same pass instead of autograd's separate add kernel */

/* ---- elementwise.hip ------------------------------------------------------------------------------ */
/* torch.nn.functional.softmax(scale * x, dim=-1) over rows (AttnBlock); y may alias x */
int odvae_softmax_rows_f32(const float* x, float* y, int64_t rows, int cols, float scale, void* stream);
int odvae_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, void* stream);
/* Attention backward with the softmax backward folded into the product dP = dO V^T ([UPSTREAM] ldm AttnBlock.forward under
 * autograd): rowdot[i] = dO[i] . O[i] (= sum_j P[i][j] dP[i][j]); dS = alpha * P .* (A B^T - rowdot[row]).
 * A [M][K], B [N][K]; P, dS [M][N] (leading dimension ldc, batch stride strideC; dS may alias P); rowdot [batch][M]. */
int odvae_rowdot_f32(const float* a, const float* b, int64_t rows, int cols, float* out, void* stream);
int odvae_gemm_softmax_bwd_f32(int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA,
                               const float* B, int ldb, int64_t strideB, const float* P, const float* rowdot,
                               int64_t strideRow, float* dS, int ldc, int64_t strideC, int batch, void* stream);
/* Attention forward with the row softmax folded into the two products ([UPSTREAM] ldm AttnBlock.forward: bmm -> softmax(dim=2) -> bmm).
 * softmax is invariant to the shift it subtracts, so the QK^T epilogue can write E = exp(scale * (q_i . k_j - bound_i)) against any bound_i >=
 * max_j q_i . k_j -- |q_i| * max_j |k_j| -- instead of waiting for the row maximum, and the PV product sums l_i = sum_j E_ij beside its
 * multiplies and divides by it: P = E / l is never formed, the T x T tensor is written once and read once (the separate softmax pass read and
 * rewrote it: 3.9 of 232 ms per step).  A bound far above the maximum underflows a whole row; odvae_gemm_rownorm_f32 then raises *flag and
 * the caller's PREDICATED fallback launches (odvae_gemm_pred_f32, odvae_softmax_rows_pred_f32: no-ops unless *flag != 0) redo the block with
 * the row maximum -- no host synchronisation either way.
 *   odvae_attn_row_bound_f32: qkv [N][T][3C] (q | k | v per token) -> bound [N*T] (unscaled), *flag = 0; nk_scratch [N*T]
 *   odvae_gemm_exp_bound_f32: E = exp(alpha * (A B^T - rowbound[row])); A [M][K], B [N][K]
 *   odvae_gemm_rownorm_f32:   C = (A B) / l[row], l = row sums of A [M][K]; B [K][N]; rinv [batch][M] = 1 / l; *flag |= (some l < 1e-30 or not finite)
 *   odvae_gemm_pred_f32:      odvae_gemm_f32 (unsplit shapes, no bias / residual) under the predicate
 *   odvae_softmax_rows_pred_f32: odvae_softmax_rows_f32 under the predicate, and ones[row] = 1
 * Backward with P given as (E, rinv): odvae_rowdot_scale_f32 (out[i] = a_i . b_i, a_scaled[i][:] = a[i][:] * row_scale[i]: D_i and dO_i / l_i),
 * odvae_gemm_softmax_bwd_scaled_f32 (dS = alpha * E .* rinv[row] .* (A B^T - rowdot[row])). */
int odvae_attn_row_bound_f32(const float* qkv, int N, int T, int C, float* bound, float* nk_scratch, int* flag, void* stream);
int odvae_gemm_exp_bound_f32(int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                             const float* rowbound, int64_t strideRow, float* E, int ldc, int64_t strideC, int batch, void* stream);
int odvae_gemm_rownorm_f32(int M, int N, int K, const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                           float* C, int ldc, int64_t strideC, float* rinv, int64_t strideRow, int* flag, int batch, void* stream);
int odvae_gemm_pred_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA,
                        const float* B, int ldb, int64_t strideB, float* C, int ldc, int64_t strideC, int batch, const int* pred, void* stream);
int odvae_softmax_rows_pred_f32(const float* x, float* y, int64_t rows, int cols, float scale, const int* pred, float* ones, void* stream);
int odvae_rowdot_scale_f32(const float* a, const float* b, const float* row_scale, int64_t rows, int cols, float* out, float* a_scaled, void* stream);
int odvae_gemm_softmax_bwd_scaled_f32(int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA,
                                      const float* B, int ldb, int64_t strideB, const float* E, const float* rowdot, const float* rinv,
                                      int64_t strideRow, float* dS, int ldc, int64_t strideC, int batch, void* stream);
/* backward of F.interpolate(scale_factor=2, mode="nearest"): dx[N][H][W][C] from du[N][2H][2W][C] */
int odvae_upsample2x_bwd_f32(const float* du, float* dx, int N, int H, int W, int C, void* stream);
/* PoseAutoencoder._rescale (src/models/autoencoder.py:434-436): NCHW in, NHWC out; workspace >= 8 KiB */
int odvae_rescale_minmax_f32(const float* x_nchw, float* y_nhwc, int N, int C, int HW, float* minmax_out,
                             void* workspace, size_t workspace_bytes, void* stream);
/* DiagonalGaussianDistribution (src/util/distributions.py:5-41): moments [N][HW][2*Cz] */
int odvae_gaussian_sample_f32(const float* moments, const float* eps, float* z, int N, int HW, int Cz, void* stream);
int odvae_gaussian_kl_f32(const float* moments, float* kl, int N, int HW, int Cz, void* stream);
int odvae_gaussian_bwd_f32(const float* moments, const float* eps, const float* dz, const float* dkl,
                           float* dmoments, int N, int HW, int Cz, void* stream);
/* PoseLoss._get_rec_loss pixel term with mask_2d_bbox applied (contperceptual.py:137,252-255):
 * out[n] = sum |x*m - xr*m|; workspace >= N*1 KiB */
int odvae_l1_masked_sum_f32(const float* x, const float* xr, const float* mask, float* out, int N, int HW, int C,
                            void* workspace, size_t workspace_bytes, void* stream);
int odvae_l1_masked_bwd_f32(const float* x, const float* xr, const float* mask, const float* g, float* dxr,
                            int N, int HW, int C, void* stream);
/* bias gradient of a 1x1 convolution: out[c] = sum_rows x[row][c] */
size_t odvae_colsum_workspace_bytes(int64_t rows, int C);
int odvae_colsum_f32(const float* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.utils.clip_grad_norm_ over one flat arena: out[0] = norm, out[1] = clip coefficient; workspace >= 4 KiB */
int odvae_grad_norm_f32(const float* g, int64_t n, float max_norm, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* torch.optim.Adam step (src/models/autoencoder.py:365-377) over flat arenas; clip = odvae_grad_norm_f32 output or NULL */
int odvae_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, int step, const float* clip, void* stream);
int odvae_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, void* stream);
/* y = x * mask[n][hw] broadcast over channels (inputs*mask_2d_bbox, contperceptual.py:252-255); x,y [npix][C] */
int odvae_mul_mask_f32(const float* x, const float* mask, float* y, int64_t npix, int C, void* stream);
/* out = z*mask + add, mask/add may be NULL (dropout on z_obj, +noise, +enc_pose: src/models/autoencoder.py:233-253) */
int odvae_latent_combine_f32(const float* z, const float* mask, const float* add, float* out, int64_t n, void* stream);

/* ---- gan_f32.hip: PatchGAN NLayerDiscriminator ([UPSTREAM] taming discriminator; contperceptual.py:285,355-356) ----
 * Conv2d(k=4, pad=1, stride 1|2) = im2col + odvae_gemm_f32 against the weight reordered to [Cout][(kh,kw,ci)];
 * cols is [N*Ho*Wo][16*C]; col2im is its adjoint (data gradient). */
int odvae_im2col4x4_f32(const float* x, float* cols, int N, int Hi, int Wi, int C, int Ho, int Wo, int stride, void* stream);
int odvae_col2im4x4_f32(const float* dcols, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int stride, void* stream);
int odvae_weight4x4_reorder_f32(const float* src, float* dst, int Cout, int Cin, int to_gemm, void* stream);
/* torch.nn.BatchNorm2d (batch statistics when train=1, running-stat update with momentum) + LeakyReLU(slope), x [rows][C] */
size_t odvae_batchnorm_workspace_bytes(int64_t rows, int C);
int odvae_batchnorm_lrelu_fwd_f32(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float slope, int train, float* mean, float* rstd,
                                  float* running_mean, float* running_var, float* y,
                                  void* workspace, size_t workspace_bytes, void* stream);
int odvae_batchnorm_lrelu_bwd_f32(const float* x, const float* dy, int64_t rows, int C, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, float slope, int train,
                                  float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.LeakyReLU(slope); the backward with slope 0 is the ReLU backward */
int odvae_leaky_relu_f32(const float* x, float* y, float slope, int64_t n, void* stream);
int odvae_leaky_relu_bwd_f32(const float* x, const float* dy, float* dx, float slope, int64_t n, void* stream);

/* ---- lpips_f32.hip: LPIPS-style perceptual distance ([UPSTREAM] taming lpips.py; contperceptual.py:143) ------------- */
/* ScalingLayer (x - shift[c]) / scale[c]; backward=1 computes x / scale[c] */
int odvae_scaling_layer_f32(const float* x, const float* shift, const float* scale, float* y, int64_t npix, int C, int backward, void* stream);
/* torch.nn.MaxPool2d(2, 2): x [N][2Ho][2Wo][C] -> y [N][Ho][Wo][C] */
int odvae_maxpool2x2_f32(const float* x, float* y, int N, int Ho, int Wo, int C, void* stream);
int odvae_maxpool2x2_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int Ho, int Wo, int C, void* stream);
/* out[n] = spatial mean of lin_w . (normalize_tensor(f0) - normalize_tensor(f1))^2 ; gradient w.r.t. f1 only */
int odvae_lpips_distance_f32(const float* f0, const float* f1, const float* w, float* out, int N, int HW, int C,
                             void* workspace, size_t workspace_bytes, void* stream);
int odvae_lpips_distance_bwd_f32(const float* f0, const float* f1, const float* w, const float* g, float* df1,
                                 int N, int HW, int C, void* stream);

/* ---- patch_u8.hip: object-patch extraction (src/data/datasets/nuscenes.py:159-192) ------------ */
/* Replaces img_pil.crop(box) (:159) -> patch.resize((S, S), BILINEAR, reducing_gap=1.0) (:176) -> ToTensor (:190-191) and
   the NEAREST-resized 2-d box mask (:178-192), bit-identical to Pillow's 8-bit fixed-point resampler.
   d_images [B] device pointers to u8 HWC RGB camera images; d_geom [B][8] int32 {img_h, img_w, crop_x1, crop_y1, crop_size,
   table_slot, 0, 0}; d_mask_rect [B][4] int32 {x_start, x_stop, y_start, y_stop} in crop coordinates; d_tables
   [n_slots][S][8] int32 per crop size {k0..k4 (coefficients * 2^22), first source index, taps, nearest source index};
   patch [B][S][S][3] f32 in [0,1], mask [B][S][S] f32 in {0,1}.  odvae_patch_table_ints(S) = ints per table slot. */
int odvae_patch_table_ints(int S);
int odvae_patch_crop_resize_u8(const void* d_images, const void* d_geom, const void* d_mask_rect, const void* d_tables,
                               int n_slots, int B, int S, void* patch, void* mask, void* stream);

/* ---- pose_f32.hip: every pose-head loss term in one launch (src/modules/losses/contperceptual.py:111-132,176-212) ------------- */
/* dec_pose [B][8+NC] (pose 4 | lhw 3 | fill 1 | class logits), moments [B][16] (box posterior mean | raw logvar), prior [L][3][8]
   (mean | var | logvar per label), prior_idx [B] (< 0 = label "background").  out [9] = pose, class, bbox, fill, kl_bbox, mean t1, t2,
   t3, v3; jac_pose [4][B][8+NC] and jac_mom [B][16] = Jacobians of out[0..3] / out[4] for odvae_pose_losses_bwd_f32 */
int odvae_pose_losses_f32(const float* dec_pose, const float* pose_gt, const float* bbox_gt, const float* fill_gt, const int64_t* class_gt,
                          const float* moments, const float* prior, const int* prior_idx, int B, int NC, int L, int background_class_idx,
                          int pose_loss_l2, int train_on_yaw, float gamma, float alpha, float* out, float* jac_pose, float* jac_mom,
                          void* stream);
int odvae_pose_losses_bwd_f32(const float* g, const float* jac_pose, const float* jac_mom, int B, int NC, float* d_dec_pose,
                              float* d_moments, void* stream);

/* ---- linear_f32.hip: small-batch dense layers of the pose-head MLPs (pose_encoder.py:59-131, pose_decoder.py:60-97) ------------ */
size_t odvae_linear_workspace_bytes(int M, int N, int K);
/* y [M][N] = act(x [M][K] . w [N][K]^T + bias); act 0 none | 1 tanh | 2 swish | 3 relu; pre (optional) = the pre-activation */
int odvae_linear_fwd_f32(const float* x, const float* w, const float* bias, int M, int N, int K, int act, float* y, float* pre,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dpre [M][N] = dy * act'(pre); dx [M][K] = dpre . w (or NULL); dw [N][K] = dpre^T . x (or NULL); bias gradient = column sum of dpre */
int odvae_linear_bwd_f32(const float* x, const float* w, const float* pre, const float* dy, int M, int N, int K, int act, float* dpre,
                         float* dx, float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* ==== bf16 mixed-precision path (BASELINE.json configs[4]; reference knobs: configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:139
 * `precision`, train.py:521).  Activations bf16 NHWC in HBM, master weights / weight gradients / statistics f32, accumulation f32
 * on v_mfma_f32_32x32x16_bf16.  Every `void*` activation pointer below is bf16 unless the comment says otherwise. ================ */

/* ---- conv_bf16.hip: 3x3 / 1x1 convolutions ([UPSTREAM] ldm model.py via feat_encoder.py:4, feat_decoder.py:4) ---------------- */
int odvae_conv_bf16_reduce_pad(int c);
int odvae_conv_bf16_out_pad(int c);
size_t odvae_conv_bf16_pack_elems(int reduce_c, int out_c, int taps);
/* OIHW f32 weights [Cout][Cin][k][k] (taps = k*k in {1, 9}) -> bf16 MFMA-fragment packs: fwd (reduce Cin, rows Cout) and dgrad
   (reduce Cout, rows Cin, taps flipped); either may be NULL */
int odvae_conv_pack_bf16(const float* w, int Cout, int Cin, int taps, void* fwd_pack, void* dgrad_pack, void* stream);
/* mode 0 stride 1 pad 1 | 1 Downsample pad(0,1,0,1)+stride 2 | 2 nearest-2x + stride 1 | 3 data gradient of mode 1 | 4 = 1x1 with the
   pixels flattened to [N][Hi][Wi] = [1][M/16][16].  x [N][Hi][Wi][Cin] (Cin % 8 == 0), bias f32 or NULL, residual bf16
   [N][Ho][Wo][Cout] or NULL, y bf16 (out_f32 = 0, Cout % 4 == 0) or f32 (out_f32 = 1) */
int odvae_conv_bf16(int mode, const void* x, int N, int Hi, int Wi, int Cin, const void* pack, int Cout, const float* bias,
                    const void* residual, void* y, int Ho, int Wo, int out_f32, void* stream);
/* The stride-1 3x3 conv (mode 0, bf16 output) whose epilogue also leaves the GroupNorm statistics of y -- (sum, sum of squares) of the
 * bf16-rounded values per output tile and channel group, gn_partial [N][odvae_conv_bf16_stats_chunks(H, W)][gn_groups][2], every slot
 * written by exactly one block -- for odvae_groupnorm_fwd_partials_bf16.  odvae_conv_bf16_stats_supported: Cout > 64, groups of whole
 * 4-channel runs that do not straddle a 128-channel block. */
int odvae_conv_bf16_stats_chunks(int H, int W);
int odvae_conv_bf16_stats_supported(int Cout, int gn_groups);
int odvae_conv_bf16_stats(const void* x, int N, int H, int W, int Cin, const void* pack, int Cout, const float* bias, const void* residual,
                          void* y, float* gn_partial, int gn_groups, void* stream);
/* ---- conv_wgrad_bf16.hip: weight gradient, dw f32 OIHW, modes 0 / 1 / 2 / 4 as above; deterministic ---------------------------- */
size_t odvae_conv_wgrad_bf16_workspace_bytes(int mode, int N, int Ho, int Wo, int Cin, int Cout);
/* db f32 [Cout] (bias gradient = per-channel sum of dy) or NULL, produced in the same pass */
int odvae_conv_wgrad_bf16(int mode, const void* x, const void* dy, int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout,
                          float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);
/* ---- flash_attn_bf16.hip: fused single-head attention ([UPSTREAM] AttnBlock.forward), scores never in HBM --------------------- */
int odvae_flash_attn_supported(int N, int T, int C);
/* qkv [N][T][3C] (q | k | v) -> o [N][T][C], lse2 f32 [N][T] = log2 sum_j exp(score_ij * scale) */
int odvae_flash_attn_fwd_bf16(const void* qkv, int N, int T, int C, float scale, void* o, float* lse2, void* stream);
/* dqkv [N][T][3C] from d_o, o, lse2; delta_ws f32 [N*T] scratch */
int odvae_flash_attn_bwd_bf16(const void* qkv, const void* o, const void* d_o, const float* lse2, int N, int T, int C, float scale,
                              void* dqkv, float* delta_ws, void* stream);
/* ---- bf16_ops.hip: GroupNorm(32, eps 1e-6) + swish with bf16 activations / f32 statistics, dtype hand-offs ---------------------- */
size_t odvae_groupnorm_bf16_workspace_bytes(int N, int HW, int C, int G);
int odvae_groupnorm_fwd_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta, float eps, int swish,
                             void* y, float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream);
/* odvae_groupnorm_fwd_bf16 without its statistics pass: partial [N][chunks][G][2] as odvae_conv_bf16_stats left them */
int odvae_groupnorm_fwd_partials_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta, float eps, int swish,
                                      void* y, float* mean, float* rstd, const float* partial, int chunks, void* stream);
int odvae_groupnorm_bwd_bf16(const void* x, const void* dy, int N, int HW, int C, int G, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, int swish, void* dx, float* dgamma, float* dbeta, const void* dx_add,
                             void* workspace, size_t workspace_bytes, void* stream);
/* y bf16 [rows][CP] = x f32 [rows][C], channels C..CP-1 zero (CP % 8 == 0) */
int odvae_cast_pad_bf16(const float* x, int64_t rows, int C, int CP, void* y, void* stream);
int odvae_cast_f32_from_bf16(const void* x, int64_t n, float* y, void* stream);
/* dx [N][H][W][C] = 2x2 sum-pool of du [N][2H][2W][C]: data gradient of F.interpolate(scale 2, nearest) */
int odvae_upsample2x_bwd_bf16(const void* du, void* dx, int N, int H, int W, int C, void* stream);
size_t odvae_colsum_bf16_workspace_bytes(int64_t rows, int C);
/* out f32 [C] = column sums of x bf16 [rows][C] (bias gradients) */
int odvae_colsum_bf16(const void* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODVAE_HIP_H */
