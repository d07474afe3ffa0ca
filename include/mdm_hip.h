/* libmdm_hip -- C ABI of the MI355X (gfx950) U-Net denoiser hot path.
 *
 * This is the drop-in boundary of the repository: every arithmetic step that
 * apple/ml-mdm's UNet / NestedUNet forward+backward dispatches to a vendor
 * library through torch.nn / torch.nn.functional is one entry point here.
 * The reference has no native boundary of its own (it is pure Python on top of
 * ATen); the "interface replaced" column therefore cites the Python call site
 * (paths relative to ml-mdm-matryoshka/ml_mdm/).
 *
 * Rules common to all entry points
 *   - plain pointers + sizes only; device pointers unless stated; no torch types
 *   - no allocation, no host synchronisation; caller owns every buffer including
 *     workspaces (sizes come from the *_plan functions, which are host-only)
 *   - `stream` is a hipStream_t passed as void* (0 = default stream)
 *   - `dtype` selects the storage type of activations / packed weights:
 *       MDM_F32 (0)  exact-fp32 MFMA path (v_mfma_f32_16x16x4_f32)  -- parity mode
 *       MDM_BF16 (1) bf16 storage, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
 *       MDM_F32_SPLIT (2; mdm_conv_fwd, mdm_conv_fwd_ws, mdm_attn_fwd only) fp32 storage and accumulators, every product
 *                    as three bf16 MFMAs on bf16 hi + lo halves of the fp32 operands (x*y to ~2^-16 relative): the
 *                    reference samples in fp32 (diffusion.py:181-197, clis/generate_sample.py:230-256); this is that
 *                    precision class (1e-3 sampling gate) at about a third of the bf16 rate instead of 1/16
 *       MDM_F32_SPLIT_W (3; mdm_conv_fwd, mdm_conv_fwd_ws only) the same arithmetic, bit for bit, with w_packed already
 *                    split into hi / lo planes by mdm_split_weight_planes (weights are constant over the 50-250 denoise
 *                    iterations of a sampling run: half the conversion work leaves the k-loop); needs k*k*Cin % 8 == 0
 *     parameters, their gradients, normalisation statistics and LSEs are always fp32
 *   - activations are NHWC: [N, H, W, C] (a linear layer is N=rows, H=W=1)
 *   - channel counts must be multiples of the 16-byte chunk: 4 (fp32) / 8 (bf16)
 *   - return value: 0 ok, <0 invalid argument (mdm_last_error() describes it),
 *     >0 the hipError_t of a failed launch
 */
#ifndef MDM_HIP_H_
#define MDM_HIP_H_
#include <stddef.h>

/* Bumped whenever an exported signature changes; mdm_abi_version() returns the value the library was built with.
 *   1  round 1
 *   2  mdm_sumsq / mdm_adamw_ema_step gained parameters, mdm_gn_bwd a third accumulate mode (round 2)
 *   4  MDM_F32_SPLIT dtype code (mdm_conv_fwd*, mdm_attn_fwd), mdm_dropout (round 4)
 *   3  round 3
 *   5  bf16 tensors: y_pre / aux of mdm_conv_fwd* hold one byte per element, the code of gelu'(pre) (round 6) */
#define MDM_HIP_ABI_VERSION 5

#ifdef __cplusplus
extern "C" {
#endif

enum { MDM_F32 = 0, MDM_BF16 = 1, MDM_F32_SPLIT = 2, MDM_F32_SPLIT_W = 3 };
enum { MDM_ACT_NONE = 0, MDM_ACT_GELU = 1, MDM_ACT_DGELU_AUX = 2 };

int mdm_abi_version(void);
const char* mdm_last_error(void);

/* ---- convolution / linear (implicit GEMM on MFMA) --------------------------------------
 * replaces nn.Conv2d 3x3 s1/s2 p1 and 1x1 (models/unet.py:199-221, 260-271, 514-532, 632, 751;
 * models/nested_unet.py:109-128) and nn.Linear (unet.py:206, 264, 605-609, 625-626, 763).
 *
 * mdm_pack_weight: reference layout (Cout, Cin, k, k) fp32  ->  w_fwd [Cout][k*k][Cin_pad] and
 *   (optional) w_dgrad [Cin][k*k flipped][Cout_pad], both of `dtype`.  kblock_* != 0 selects the
 *   channel-block-major reduction order for a 3x3 pack (k = (c / B) * 9B + tap * B + c % B, B = 64 for bf16,
 *   32 for fp32; needs C % B == 0): the 9 taps of a channel block become consecutive k-tiles, which keeps the
 *   shifted re-reads of the activation in L1/L2.  The same value must be passed to mdm_conv_fwd as `kblock`.
 * mdm_conv_fwd:  y = epilogue(conv(x, w_packed)).  epilogue = +bias, act, +res (in this order);
 *   act == MDM_ACT_GELU also leaves, in y_pre when non-null, what the backward through the GELU needs (unet.py:270):
 *     fp32 tensors: the pre-activation (same shape and dtype as y);
 *     bf16 tensors: ONE BYTE per element, q = round(196 gelu'(pre)) + 28 (gelu' = (q - 28) / 196: 0 and 1 exact, step
 *     5.1e-3) -- the backward uses the pre-activation only through gelu', and the launch is store-bound;
 *   act == MDM_ACT_DGELU_AUX multiplies by gelu'(aux) for fp32 tensors, by the decoded byte aux[m * Cout + n] for bf16
 *     tensors (aux = the y_pre of the MDM_ACT_GELU launch; backward through the FFN GELU); res must be NULL with it.
 *   The same entry computes the input gradient: pass dy as x and w_dgrad as w_packed
 *   (transposed = 1 for the gradient of a stride-2 convolution: Ho = 2H, Wo = 2W).
 * mdm_conv_wgrad + mdm_conv_wgrad_reduce: dw (Cout, Cin, k, k) fp32 = sum_m dy[m, :] (x) im2col(x)[m, :].
 *   The first call runs the split GEMM into fp32 slabs in ws (size from mdm_conv_wgrad_plan); the second sums the
 *   slabs into the reference OIHW layout.  want_bias / dbias != NULL additionally produce the bias gradient
 *   sum_m dy[m, :] (the bf16 kernels get it from the dY fragments they already hold -- dot products against ones, dealt
 *   out over the waves and k-tile blocks that hold the same fragments; partial rows + a row count live behind the slabs).
 * mdm_colsum: out[c] = sum_m x[m, c]  (bias gradients); ws from mdm_colsum_plan.
 *   `accumulate` != 0 (here and in mdm_gn_bwd / mdm_ln_bwd) adds the parameter gradient into the destination
 *   instead of overwriting it: the caller points it at the parameter's slot of a flat gradient arena, which
 *   removes the per-parameter "grad += new" kernels of the autograd engine.
 */
int mdm_pack_weight(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int ksize, int Cin_pad,
                    int Cout_pad, int kblock_fwd, int kblock_dgrad, int dtype, void* stream);
/* planes for MDM_F32_SPLIT_W: the n fp32 values (n % 8 == 0) of a packed weight (either output of mdm_pack_weight with
 * dtype MDM_F32) -> `planes` (4 n bytes, a different buffer): run r of 8 values becomes [8 bf16 hi | 8 bf16 lo],
 * hi = bf16(v), lo = bf16(v - hi) -- the split the MDM_F32_SPLIT k-loop performs on the fly (no reference counterpart: the
 * reference samples in plain fp32, diffusion.py:181-197) */
int mdm_split_weight_planes(const float* w_packed, void* planes, size_t n, void* stream);
/* every kernel-layout weight of a model in ONE launch (they all go stale at each optimizer step).  `table`: DEVICE
 * array of n 48-byte descriptors {const float* w; void* w_fwd; void* w_dgrad (or NULL); int Cout, Cin, taps (1 | 9),
 * kblock_fwd, kblock_dgrad, first_block}; first_block = running sum of (Cout/32)*(Cin/32); total_blocks = the final
 * sum.  Needs Cout % 32 == 0 and Cin % 32 == 0 (no channel padding) for every entry. */
int mdm_pack_weights_multi(const void* table, int n, int total_blocks, int dtype, void* stream);
/* mdm_conv_fwd picks the kernel: the implicit-GEMM tiles (256x256 / 256x192 / 128x128, by tile count), or -- bf16 3x3
 * stride 1 with 32 / 64 channels on both sides, plain epilogue, >= 65536 pixels, H % 8 == 0, W % 64 == 0: the narrow outer
 * levels of the nested models (nested_unet.py:109-128, resolution_channels [32, 32, 64] / [64, 128, ...]) -- a direct
 * convolution that stages each halo tile once (conv3x3_direct_kernel); same arguments, same rounding points. */
int mdm_conv_fwd(const void* x, const void* w_packed, const float* bias, const void* res, const void* aux, void* y,
                 void* y_pre, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int ksize, int stride,
                 int transposed, int act, int kblock, int dtype, void* stream);
/* Small problems (sampling at batch 1-4): when the output tiles cannot fill the chip the reduction is split over more
 * blocks -- partial fp32 tiles in `ws`, then one sum + epilogue launch (same rounding points as the fused epilogue).
 * mdm_conv_fwd_plan: splits (1 = none) and the workspace bytes for a [M, Cout] output with reduction length K;
 * mdm_conv_fwd_ws: mdm_conv_fwd with that workspace (NULL / too small = never split). */
int mdm_conv_fwd_plan(int M, int Cout, int K, int dtype, int* splits, size_t* ws_bytes);
int mdm_conv_fwd_ws(const void* x, const void* w_packed, const float* bias, const void* res, const void* aux, void* y,
                    void* y_pre, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int ksize, int stride,
                    int transposed, int act, int kblock, int dtype, float* ws, size_t ws_bytes, void* stream);
/* Sub-pixel forms of the resampling convolutions (ABI 3; bf16).  A stride-2 3x3 convolution (models/unet.py:514-522) and
 * a 3x3 convolution of a nearest-2x-upsampled image (unet.py:567-569) touch, per 2x2 block of the high-resolution side,
 * 2x2 low-resolution pixels per phase: over 2x2-blocked tensors they are 2x2 correlations -- 16 weight blocks instead of
 * 36, 2.25x fewer multiply-adds -- and the upsampled tensor never exists.
 *   mdm_conv_s2_dgrad     dx [N, 2Ho, 2Wo, Cin] of the stride-2 conv from dy [N, Ho, Wo, Cout]; w_sel [4 Cin][Cout/64][4][64]
 *                         (row (ph, pw, ci), tap j = 2 dh + dw reads dy[b + d]; zeros where a phase has no tap).
 *                         Cout % 64 == 0, Cin % 128 == 0, even H / W.  mdm_conv_fwd(transposed = 1) stays for the rest.
 *   mdm_s2dgrad_pack      w (Cout, Cin, 3, 3) fp32 -> that w_sel (one launch; re-run after every optimizer step).
 *   mdm_upconv_pack       w (Cout, Cin, 3, 3) fp32 -> w_ph [4 Cout][Cin/64][4][64] (forward) and w_t [Cin][4][Cout/64][4][64]
 *                         (input gradient): the 3x3 taps that fall on one low-resolution pixel of a phase, summed.
 *   mdm_conv_up_fwd       y [N, 2H, 2W, Cout] = conv3x3(upsample2x(x)) + bias from x [N, H, W, Cin]; bias4 = bias x 4 phases.
 *   mdm_space_to_depth2x  dyb [N, H, W, 4C] (channel (ph, pw, c)) from dy [N, 2H, 2W, C]: operand of the two gradients
 *   mdm_conv_up_dgrad     dx [N, H, W, Cin] from dyb and w_t.
 *   mdm_conv_wgrad_blocked + mdm_conv_wgrad_reduce(Cout = 4 Cout, ksize 3) + mdm_upconv_wfold
 *                         dW (Cout, Cin, 3, 3) (+)=, dbias (+)=: the split GEMM over x and dyb computes only the 16 needed
 *                         (phase, tap) blocks of dwb (4 Cout, Cin, 3, 3); the fold adds them into the 3x3 taps.
 *                         ws as mdm_conv_wgrad_plan(N H W, 4 Cout, 9 Cin).  Cin % 256 == 0, Cout % 256 == 0, H / W powers of 2. */
/* mdm_conv_fwd_gn (ABI 3): convolution (1x1 or 3x3, stride 1) + bias (+ residual) AND the GroupNorm of its output in
 *   ONE launch: y as mdm_conv_fwd, plus y_norm = act(GroupNorm(y)), stats [N][G][2], coef [N][Cout][2] -- exactly what
 *   mdm_gn_fwd would produce from y (gn_act: 0 none, 1 SiLU).  Replaces nn.Conv2d followed by nn.GroupNorm where a
 *   256-row tile is one sample (16x16 images) and a 192-column tile 8 whole groups of 24 channels (unet.py:310-311
 *   proj_out -> ffn[0]; the last conv of a layer -> the next layer's first norm).  mdm_conv_fwd_gn_ok (host-only) tells. */
int mdm_conv_fwd_gn_ok(int N, int H, int W, int Cin, int Cout, int ksize, int kblock, int groups, int dtype);
int mdm_conv_fwd_gn(const void* x, const void* w_packed, const float* bias, const void* res, void* y, int N, int H, int W,
                    int Cin, int Cout, int ksize, int kblock, const float* gamma, const float* beta, int groups, float eps,
                    int gn_act, void* y_norm, float* stats, float* coef, int dtype, void* stream);
int mdm_conv_s2_dgrad(const void* dy, const void* w_sel, void* dx, int N, int Ho, int Wo, int Cout, int Cin, int dtype,
                      void* stream);
/* the same + res [N, 2Ho, 2Wo, Cin] (or NULL) added to dx in the epilogue: the gradient of a skip connection that taps the
 * tensor a down-sampling convolution reads (unet.py:566-567, 883-897), instead of a separate accumulation pass */
int mdm_conv_s2_dgrad_res(const void* dy, const void* w_sel, const void* res, void* dx, int N, int Ho, int Wo, int Cout,
                          int Cin, int dtype, void* stream);
int mdm_s2dgrad_pack(const float* w_oihw, void* w_sel, int Cout, int Cin, void* stream);
int mdm_upconv_pack(const float* w_oihw, void* w_ph, void* w_t, int Cout, int Cin, void* stream);
int mdm_conv_up_fwd(const void* x, const void* w_ph, const float* bias4, void* y, int N, int H, int W, int Cin, int Cout,
                    int dtype, void* stream);
int mdm_space_to_depth2x(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int mdm_conv_up_dgrad(const void* dyb, const void* w_t, void* dx, int N, int H, int W, int Cout, int Cin, int dtype,
                      void* stream);
int mdm_conv_wgrad_blocked(const void* x, const void* dyb, int want_bias, float* ws, int N, int H, int W, int Cin, int Cout,
                           int dtype, void* stream);
int mdm_upconv_wfold(const float* dwb, float* dw, const float* dbias4, float* dbias, int Cout, int Cin, int accumulate,
                     void* stream);
int mdm_conv_wgrad_plan(int M, int Cout, int K, int dtype, int* splits_out, size_t* ws_bytes);
int mdm_conv_wgrad(const void* x, const void* dy, int want_bias, float* ws, int N, int H, int W, int Cin, int Ho,
                   int Wo, int Cout, int ksize, int stride, int dtype, void* stream);
int mdm_conv_wgrad_reduce(const float* ws, float* dw_oihw, float* dbias, const void* dy, int M, int Cin, int Cout,
                          int ksize, int accumulate, int dtype, void* stream);
/* Grouped weight gradient of `groups` (<= 32) same-shape 1x1 convolutions / linear layers, bf16: ONE launch, no split
 * of the pixel reduction, results ADDED into dw[g] (Cout, Cin) / dbias[g] (Cout, may be NULL) -- the parameters' slots
 * of a flat gradient arena.  x[g] [M, Cin], dy[g] [M, Cout]; the pointer arrays are HOST arrays of device pointers.
 * mdm_conv_wgrad_group_plan: *tile_out = 128 / 256 when grouping fills the chip without a split, 0 when the caller
 * should launch the problems one by one (mdm_conv_wgrad). */
int mdm_conv_wgrad_group_plan(int M, int Cout, int K, int dtype, int groups, int* tile_out);
int mdm_conv_wgrad_grouped(const void* const* x, const void* const* dy, float* const* dw, float* const* dbias,
                           int groups, int M, int Cin, int Cout, int dtype, void* stream);
/* y[g] [M, Cout] = x[g] [M, Cin] w_packed[g]^T + bias[g] for `groups` (<= 32) linear layers of ONE shape in one launch
 * (bf16, Cin % 64 == 0): the key / value projections of the text states of every attention layer (unet.py:264).  HOST
 * arrays of device pointers; bias may be NULL.  With the dgrad packs and the output gradients: the input gradients. */
int mdm_linear_grouped(const void* const* x, const void* const* w_packed, const float* const* bias, void* const* y,
                       int groups, int M, int Cin, int Cout, int dtype, void* stream);
int mdm_colsum_plan(int M, int C, int* nblocks, size_t* ws_bytes);
int mdm_colsum(const void* x, float* out, float* ws, int M, int C, int accumulate, int dtype, void* stream);

/* ---- GroupNorm (+FiLM) (+SiLU), LayerNorm -----------------------------------------------
 * replaces nn.GroupNorm + F.silu + the FiLM modulation `norm2(h) * (1 + ta) + tb`
 * (unet.py:224, 233, 259, 268, 878) and nn.LayerNorm on the text states (unet.py:263, 304).
 *   y = act(GN(x; gamma, beta) * (1 + film[:, :C]) + film[:, C:]),  act: 0 none, 1 SiLU
 *   stats [N][G][2] = (mean, rstd), coef [N][C][2]: saved by forward, consumed by backward.
 *   mdm_gn_bwd writes dx, dgamma[C], dbeta[C] (fp32 atomic accumulation of one term per sample: `accumulate` == 0
 *   zero-fills them first, 1 adds into them) and dfilm [N][2C] (when film != NULL).  `accumulate` == 2: dgamma / dbeta
 *   are per-sample rows [N][C] filled with plain stores (no atomics -- at the 16x16 level the atomics of 64 samples
 *   into the same 1536 addresses cost as much as the kernel's own traffic); mdm_gn_param_reduce_multi then adds the
 *   rows of many layers into their gradient slots in one launch (table: DEVICE array of 48-byte descriptors
 *   {const float* pg, pb; float* dgamma, dbeta; int N, C, first_block, pad}, first_block = prefix sum of
 *   ceil(C / 256); total_blocks = the sum).  dres (same shape as x, may be
 *   NULL) is added into dx: the gradient that reaches x through the residual branch of the block the norm opens
 *   (h = x + f(norm(x)), unet.py:238, 309, 312) -- saves the separate accumulation kernel of the autograd engine.
 *   dres2 (ABI 3, may be NULL): a second such gradient -- x is also a skip activation consumed by the up path
 *   (unet.py:883-897, 545-547), whose gradient would otherwise be added to dx by one more elementwise kernel.
 *   ws (fp32) size from mdm_gn_plan (valid for both directions).
 */
int mdm_gn_plan(int N, int HW, int C, int G, size_t* ws_bytes);
int mdm_gn_fwd(const void* x, const float* gamma, const float* beta, const void* film, void* y, float* stats,
               float* coef, float* ws, int N, int HW, int C, int G, float eps, int act, int dtype, void* stream);
int mdm_gn_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const void* film,
               const float* stats, const float* coef, const void* dres, const void* dres2, void* dx, float* dgamma,
               float* dbeta, void* dfilm, float* ws, int N, int HW, int C, int G, int act, int accumulate, int dtype,
               void* stream);
int mdm_gn_param_reduce_multi(const void* table, int n, int total_blocks, void* stream);
int mdm_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int R, int D, float eps,
               int dtype, void* stream);
/* ws: fp32 [ceil(R/64)][D][2] */
int mdm_ln_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, float* dgamma,
               float* dbeta, float* ws, int R, int D, int accumulate, int dtype, void* stream);

/* One input, L (<= 32) LayerNorms: every attention layer normalises the SAME text states with its own gamma / beta
 * before its key / value projection (unet.py:263-264, 304).  mdm_ln_multi_fwd: statistics once, L affine outputs;
 * mdm_ln_multi_bwd: dx = the SUM of the L input gradients, dgamma[l] / dbeta[l] (+)= by fp32 atomics (`accumulate` == 0
 * zero-fills them first).  gamma / beta / y / dy / dgamma / dbeta are HOST arrays of device pointers.  D <= 4096. */
int mdm_ln_multi_fwd(const void* x, const float* const* gamma, const float* const* beta, void* const* y, int L,
                     float* stats, int R, int D, float eps, int dtype, void* stream);
int mdm_ln_multi_bwd(const void* const* dy, const void* x, const float* const* gamma, const float* stats, void* dx,
                     float* const* dgamma, float* const* dbeta, int L, int R, int D, int accumulate, int dtype,
                     void* stream);

/* ---- fused self + text cross attention ---------------------------------------------------
 * replaces SelfAttention.attention x2 + the sum (unet.py:276-307): einsum QK^T, fp32 softmax, einsum PV.
 *   qkv [B, L, 3C] = (q | k | v), kvc [B, S, 2C] = (k_c | v_c) or NULL, mask [B, S] (0/1 floats) or NULL
 *   out = softmax(q k^T / sqrt(d)) v + softmax(q k_c^T / sqrt(d) masked) v_c          [B, L, C], C = H * d
 *   out_cross (required with kvc: the kernel stages the cross term through it, and the backward reads it) = the cross
 *   term alone; lse_* [B, H, L] fp32 (optional in the forward).
 * mdm_attn_bwd overwrites dqkv [B, L, 3C] and dkvc [B, S, 2C]; delta_* are fp32 [B, H, L] scratch.
 * d must be one of 32, 64, 96, 128.
 */
int mdm_attn_fwd(const void* qkv, const void* kvc, const float* mask, void* out, void* out_cross, float* lse_self,
                 float* lse_cross, int B, int L, int S, int H, int d, int dtype, void* stream);
int mdm_attn_bwd(const void* qkv, const void* kvc, const float* mask, const void* out, const void* out_cross,
                 const void* dout, const float* lse_self, const float* lse_cross, float* delta_self,
                 float* delta_cross, void* dqkv, void* dkvc, int B, int L, int S, int H, int d, int dtype,
                 void* stream);

/* ---- streaming helpers ---------------------------------------------------------------------
 * layout conversion at the model boundary (the reference is NCHW fp32, unet.py:971-987),
 * torch.cat skip connections (unet.py:545-547; dir = 1 is the backward split),
 * F.interpolate(scale_factor=2) nearest (unet.py:567-569) and its adjoint,
 * F.silu on the time embedding (unet.py:227, 844), sin/cos timestep embedding (unet.py:834-836),
 * the masked mean over text tokens (unet.py:854-861), dtype casts.
 * mdm_elementwise op: 0 out = silu(a); 1 out = b * silu'(a); 2 out = a + b; 3 out = a.
 */
int mdm_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype, void* stream);
int mdm_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cs, int dtype, void* stream);
int mdm_concat(void* a, void* b, void* out, size_t M, int C1, int C2, int dir, int dtype, void* stream);
int mdm_upsample2x(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int mdm_downsum2x(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream);
int mdm_elementwise(const void* a, const void* b, void* out, size_t n, int op, int dtype, void* stream);
int mdm_cast(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, void* stream);
int mdm_sincos_emb(const float* times, const float* freqs, void* out, int B, int half, int dtype, void* stream);
int mdm_masked_mean(const void* x, const float* mask, void* y, int B, int S, int D, int dtype, void* stream);
int mdm_masked_mean_bwd(const void* dy, const float* mask, void* dx, int B, int S, int D, int accumulate, int dtype,
                        void* stream);

/* ---- optimizer tail over flat fp32 arenas ---------------------------------------------------
 * replaces clip_grad_norm_ + AdamW.step + ModelEma.update + zero_grad (trainer.py:52-58,79-92;
 * clis/train_parallel.py:122-127; models/model_ema.py:25-34) -- thousands of per-tensor kernels in the
 * reference -- by two streaming passes.  All arenas hold the parameters in the same order.
 *   mdm_sumsq:          out[0] = sum g^2 (device scalar, no host sync); ws = fp32[1024]; step_counter (device int or
 *                       NULL) is incremented when the sum is finite -- the optimizer's step number then lives on the device
 *   mdm_adamw_ema_step: gs = gnorm_sq ? min(1, clip / (sqrt(*gnorm_sq) + 1e-6)) : 1; AdamW on (p, gs*g, m, v) with
 *                       bias correction for 1-based `step` (or *step_dev when non-NULL); ema = ema*d + p*(1-d) if
 *                       ema != NULL; g = 0 if zero_grad.  A non-finite *gnorm_sq (a NaN / inf loss or gradient) skips the
 *                       update and only clears g: what trainer.py:38-41 does by returning before backward, without the
 *                       host having to look at the loss in the middle of the step.
 */
int mdm_sumsq(const float* g, float* out, float* ws, size_t n, int* step_counter, void* stream);
int mdm_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, const float* gnorm_sq, size_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                       float clip, float ema_decay, int zero_grad, void* stream);

/* ---- per-pixel arithmetic around the denoiser (NCHW fp32 images, as the reference keeps them) ---------------
 * All [B, chw] images need chw % 4 == 0.  gamma / gamma_last are per-sample device vectors [B].
 * `rng_state` is a device pointer to {uint64 seed, uint64 offset}: Philox4x32-10 + Box-Muller, element i = lane
 * i & 3 of counter block offset + i / 4 of stream `rng_stream` -- reproducible on the host (oracle/philox_ref.py).
 * mdm_rng_advance bumps the offset on the device (graph-capturable); mdm_randn fills n (% 4 == 0) normals.
 *
 * mdm_sampler_step: one reverse-diffusion update, replaces Sampler.get_prediction_xt_last +
 *   clip_sample + the guidance combine of forward_model (samplers.py:281-345, 445-456, 500-508):
 *     p = pred_uncond ? pred_uncond + guidance * (pred - pred_uncond) : pred
 *     x0 = (pred_type == 2 (V)) ? x_t sqrt(g) - p sqrt(1-g) : (x_t - p sqrt(1-g)) / sqrt(g)
 *     clip 0: none; 1: clamp(x0 s, -1, 1) / s; 2: clamp(x0 s, -thr[b], thr[b]) / thr[b] / s (dynamic thresholding, thr
 *       = the clamped quantile of |x0 s|, computed by the caller); 3: write the UNCLIPPED x0 s to x0_out and stop
 *     mode 0 (ddim_eta None): x_last = x0 beta sqrt(gl) / (1-g) + x_t sqrt(alpha) (1-gl) / (1-g)
 *     mode 1: eps = (x_t - x0 sqrt(g)) / sqrt(1-g); x_last = x0 sqrt(gl) + eps sqrt(1 - gl - eta^2 beta~) (eta = 0: no noise)
 *     need_noise: x_last += sqrt(beta~) * noise_gate[0] * n, n = noise[] if given, else drawn from rng_state
 * mdm_noise_images (N3, samplers.py:244-246): x_t = sqrt(g) images inv_scale + sqrt(1-g) eps; eps == NULL draws it
 *   from rng_state and stores it to eps_out.
 * mdm_diffusion_loss_fwd / _bwd (N3, diffusion.py:144-168 + samplers.py:266-279, 347-390): loss[b] = mean over chw of
 *   (prediction mapped to the loss-target space - target)^2 for prediction / target types 0,1 (eps) or 2 (v);
 *   dpred = gloss[b] * d loss[b] / d pred.  ws from mdm_diffusion_loss_plan.
 * mdm_avgpool: F.avg_pool2d(x, r) of the image pyramid (diffusion.py:346-348).
 * mdm_sample_std_fwd / _bwd: y = x / std(x over chw, unbiased) per sample (models/unet.py:871-872, the nested level
 *   with skip_normalization = false); stats [N][2] = (mean, 1 / std); ws = fp32 [N][64][2].
 * mdm_input_stage (N4, clis/train_parallel.py:194-195): uint8 NHWC [B,H,W,3] -> fp32 NCHW, (u - 127) / 128.
 */
int mdm_rng_advance(unsigned long long* rng_state, unsigned long long blocks, void* stream);
int mdm_randn(float* out, size_t n, const unsigned long long* rng_state, int rng_stream, void* stream);
int mdm_sampler_step(const float* x_t, const float* pred, const float* pred_uncond, float guidance, const float* gamma,
                     const float* gamma_last, const float* noise, const float* noise_gate, const float* thr,
                     const unsigned long long* rng_state, int rng_stream, float* x0_out, float* x_last_out, int B,
                     size_t chw, int pred_type, int mode, float ddim_eta, int need_noise, int clip, float image_scale,
                     void* stream);
int mdm_noise_images(const float* images, const float* eps, const float* gamma, float inv_scale, float* x_t,
                     float* eps_out, const unsigned long long* rng_state, int rng_stream, int B, size_t chw,
                     void* stream);
int mdm_diffusion_loss_plan(int B, size_t chw, size_t* ws_bytes);
int mdm_diffusion_loss_fwd(const float* x_t, const float* pred, const float* images, const float* eps,
                           const float* gamma, float inv_scale, float* loss, float* ws, int B, size_t chw,
                           int pred_type, int target_type, void* stream);
int mdm_diffusion_loss_bwd(const float* x_t, const float* pred, const float* images, const float* eps,
                           const float* gamma, const float* gloss, float inv_scale, float* dpred, int B, size_t chw,
                           int pred_type, int target_type, void* stream);
int mdm_avgpool(const float* x, float* y, int N, int C, int H, int W, int r, void* stream);
int mdm_sample_std_fwd(const float* x, float* y, float* stats, float* ws, int N, size_t chw, void* stream);
int mdm_sample_std_bwd(const float* dy, const float* x, const float* stats, float* dx, float* ws, int N, size_t chw,
                       void* stream);
int mdm_input_stage(const void* u8_nhwc, float* out_nchw, int B, int H, int W, void* stream);
/* mdm_dropout (ABI 4; nn.Dropout of a ResNet block, models/unet.py:208,234): y[i] = keep(i) ? x[i] / (1 - p) : 0 over n
 * elements (n % 8 == 0), keep(i) = Philox4x32-10 word of element i under (seed, offset) >= p * 2^32.  The mask is a pure
 * function of (seed, offset, i): the backward pass calls the same entry on dy with the same two integers.  One call
 * consumes n / 4 counter blocks (the caller advances its offset by that much). */
int mdm_dropout(const void* x, void* y, size_t n, float p, unsigned long long seed, unsigned long long offset, int dtype,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_H_ */
