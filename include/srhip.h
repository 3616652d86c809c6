/* libsrhip -- C-ABI of the MI355X (gfx950) hot path for SR-CACO-2 patch-level
 * super-resolution: 3x3 residual convolutions, pixel shuffle, SwinIR window
 * attention + MLP, L1/L2 losses, PSNR-family metrics, optimizers.
 *
 * The reference (sbelharbi/sr-caco-2) has no FFI: its hot path is stock aten
 * calls made from Python.  Each entry point below
 * therefore names the reference call site(s) it replaces (file:line relative
 * to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *  - Every function returns 0 on success and a negative code on failure;
 *    srhip_last_error() returns a thread-local message.  Nothing calls exit().
 *  - All pointers are DEVICE pointers owned by the caller (PyTorch's caching
 *    allocator in practice).  Kernels never allocate or free; scratch space is
 *    passed in (sizes from the *_ws / *_plan queries).
 *  - Every call takes the hipStream_t to launch on (as void*), is asynchronous
 *    and does not synchronise.  Calls are re-entrant across streams.
 *  - Activations are fp32, channels-last: an image is [B][H][W][C] = a token
 *    matrix [B*H*W][C].  1-channel network inputs/outputs are plain [B][H][W].
 *  - "ld*" arguments are row pitches in floats and must be multiples of 4.
 */
#ifndef SRHIP_H
#define SRHIP_H
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the prototypes between this push and the pop at the end of the file are its
 * only dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

const char* srhip_last_error(void);
int srhip_abi_version(void);
/* 1 in a build made with `make EXPERIMENTS=1` (tuning / ablation switches of the C side are read from the environment, phase
   stamps compiled in), 0 in the shipped library, which takes its defaults and reads no environment. */
int srhip_experiments_enabled(void);
/* Arithmetic of the split-MFMA contractions (srhip_gemm_nt_bx3, srhip_conv3x3_nhwc_bx3), per calling THREAD (thread-local;
 * an entry point reads it when it enqueues its kernels, so it is a property of the call, not of the stream or the process):
 *   0 (default)  f32-accurate: six bf16 products of the three-way split operands
 *   1            reduced precision for INFERENCE: one bf16 product of the leading planes, f32
 *                accumulation -- the role of the reference's --amp autocast at evaluation time
 *                (model_plain.py:322-327, eval_all.sh); outputs differ from mode 0 at the 1e-3
 *                relative level (gate: PSNR within 0.01 dB).  Training always runs mode 0.
 * Re-entrant across threads and streams; a thread that wants both modes brackets the calls (net.amp does). */
int srhip_set_matmul_mode(int mode);
int srhip_get_matmul_mode(void);

/* ---- dense contractions on the exact-f32 MFMA ------------------------------ */

/* C[M,N] = epi( pro(A)[M,K] . W[N,K]^T + bias ).
 * Replaces nn.Linear forward (network_swinir.py:40-43,148,177) and, called with
 * a transposed weight copy, its data gradient.
 *   a_mode 0: A as is | 1: (A-mean)*rstd with ln_stats[M][2] (LayerNorm folded
 *          into W, see srhip_fold_layernorm; network_swinir.py:293,335)
 *        | 2: gelu(A) (exact erf, network_swinir.py:41)
 *   epi 0: +bias | 1: relu | 2: R + s*(acc+bias) (residual add with per-sample
 *          DropPath scale s = alpha*rowscale[row/rows_per_scale];
 *          network_swinir.py:334-335) | 3: s*acc*gelu'(R) | 4: acc*(R>0)
 *   aux (epi 3 only, may be NULL): second output gelu(R), the activation the fc2
 *          weight gradient needs -- erf is evaluated once for both */
int srhip_gemm_nt(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                  long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                  const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha,
                  float* aux, long ldaux, void* stream);

/* zcount independent products C_z[M,N] = A_z[M,K] . W_z[N,K]^T in ONE launch of the same kernel; the bases of problem z
 * are A + (z / zdiv) * a_z0 + (z % zdiv) * a_z1 (floats; likewise W, C): the (sample, head) slices of row-major
 * [B*T][heads*dh] matrices.  Replaces the per-(sample, head) `q @ k^T` / `attn @ v` of SelfAttention / CrossAttention
 * (network_act.py:151-227), einsum 'bhid,bhjd->bhij' / 'bhij,bhjd->bhid'. */
int srhip_gemm_nt_batched(const float* A, long lda, long a_z0, long a_z1, const float* W, long ldw, long w_z0, long w_z1, float* C,
                          long ldc, long c_z0, long c_z1, int M, int N, int K, int zcount, int zdiv, void* stream);

/* 3x3 / stride 1 / pad 1 convolution, NHWC, implicit GEMM.  Wp is the tap-major
 * pack [9][Cout][Cin] from srhip_pack_conv_weight (forward) or its flipped /
 * transposed twin (data gradient).  Epilogues as srhip_gemm_nt (0,1,2,4) plus
 *   6: LeakyReLU with negative slope alpha (network_swinir.py:857)  7: its backward,
 *      acc * (R > 0 ? 1 : alpha) with R = the activation's output.
 * Replaces default_conv / nn.Conv2d(.,.,3,1,1): network_nlsn.py:38-41,89-93,
 * network_swinir.py:544,850,700. */
int srhip_conv3x3_nhwc(const float* X, long ldx, const float* Wp, const float* bias, float* Y, long ldy,
                       int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                       const float* rowscale, float alpha, void* stream);

/* ---- the same contractions on the bf16 MFMA with 3-way split operands -------
 * x = x_h + x_m + x_l (bf16 each, 3 x 8 = 24 significant bits); a*b is summed
 * from the six cross products >= 2^-24 relative, in f32: f32-accurate results
 * at 2.67x the matrix-core rate of the f32 MFMA.  The weight operand is split
 * ONCE per step: out = planes [3][Kp/16][rows][16] bf16 (16-k sub-chunk major: the
 * rows a block stages per K chunk are contiguous), Kp = srhip_bf16x3_kp(K) (K
 * rounded up to 32, zero filled); rows = N for a Linear weight, 9*Cout for the
 * tap-major conv pack.  Same prologues / epilogues / reference lines as
 * srhip_gemm_nt and srhip_conv3x3_nhwc. */
/* stats_out (srhip_gemm_nt_bx3 only, may be NULL; needs N <= 192): {mean, rstd}[M] of the
 * output rows, i.e. srhip_layernorm_fwd of C for the next LayerNorm-prologue GEMM. */
int srhip_bf16x3_kp(int K);
int srhip_split_bf16x3(const float* W, long ldw, int rows, int K, void* out, void* stream);
/* Per-step weight preparation of a whole network in ONE launch (replaces the
 * per-weight srhip_fold_layernorm / srhip_transpose / srhip_pack_conv_weight /
 * srhip_bias_expand / srhip_split_bf16x3 sequence, same results).  The caller
 * fills the job table on the host (blk0 = running sum of srhip_prep_blocks(),
 * ascending), copies it to the device once and re-runs the launch after every
 * optimizer step.
 *   kind 0  bf16x3 planes: out[3][Kp(n2)/16][n1*n0][16] of
 *           v(tap<n1, r<n0, k<n2) = a[off + tap*s0 + r*s1 + k*s2] * g,
 *           g = 1 (mode 0) | b[k] (mode 1) | b[r] (mode 2)
 *           -- a Linear weight, its transpose, LayerNorm gamma folded either
 *           way, a conv weight as tap-major pack or flipped/transposed twin
 *   kind 1  folded bias: out[n<n0] = b[n] + sum_k a[n][k<n1] * c[k]
 *   kind 2  relative-position bias images (srhip_bias_expand): a = table,
 *           out = biasT, out2 = biasN, n0 = heads */
typedef struct {
  const float* a; const float* b; const float* c; void* out; void* out2;
  int kind, blk0;
  int n0, n1, n2;
  int s0, s1, s2, off;
  int mode;
} srhip_prep_entry;
int srhip_prep_blocks(const srhip_prep_entry* e);
int srhip_prep_table(const srhip_prep_entry* table_dev, int n, int total_blocks, void* stream);
int srhip_gemm_nt_bx3(const float* A, long lda, const void* Wb, const float* bias, float* C,
                      long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                      const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha,
                      float* aux, long ldaux, float* stats_out, void* stream);
/* Data-gradient GEMM with the LayerNorm backward fused into its epilogue:
 *   dxh = A . W^T                       (W = transposed, gamma-folded weight planes)
 *   out = res + rstd * (dxh - mean_c(dxh) - xhat * mean_c(dxh * xhat)),  xhat = (x - mean) * rstd
 * with stats[M][2] = {mean, rstd} of x (srhip_layernorm_fwd), res = gradient arriving
 * over the residual connection (may be NULL).  N <= 192 (the row must fit one block).
 * Replaces srhip_gemm_nt_bx3 + srhip_layernorm_bwd for the backward of
 * network_swinir.py:293,335. */
int srhip_gemm_nt_bx3_lnbwd(const float* A, long lda, const void* Wb, float* out, long ldo, int M, int N, int K,
                            const float* x, long ldx, const float* stats, const float* res, long ldres,
                            void* stream);
/* The same two contractions with the weight operand as TWO fp16 planes and a power-of-two scale per weight row
 * (srhip_prep_table job kind 3: planes [2][Kp/16][N][16] fp16 of W[n][:] * 2^s(n), then N floats 2^-s(n)); the
 * activation rows get their own power-of-two scale inside the kernel (a priori behind the LayerNorm prologue, else from
 * a pass over the row), the product is h.h + h.l + l.h on the fp16 MFMA with f32 accumulation and the block exponents
 * are undone exactly: three products instead of six, results indistinguishable from an f32 matmul (every row relative to
 * itself; tools/split_accuracy.py, tests/test_gpu_fallback_kernels.py).  N must run on 192-column tiles (a multiple of
 * 180, or > 128 and not a multiple of 128).  Same arguments, prologues and epilogues as the _bx3 entry points. */
int srhip_gemm_nt_f16x2(const float* A, long lda, const void* Wh, const float* bias, float* C,
                        long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                        const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha,
                        float* aux, long ldaux, float* stats_out, void* stream);
int srhip_gemm_nt_f16x2_lnbwd(const float* A, long lda, const void* Wh, float* out, long ldo, int M, int N, int K,
                              const float* x, long ldx, const float* stats, const float* res, long ldres,
                              void* stream);
/* The 3x3 conv with its weight as two fp16 planes and a power-of-two scale per OUTPUT channel (srhip_prep_table job
 * kind 4); the activation gets ONE power-of-two scale per 8 x 16 (or 4 x 16) halo tile, kept as a running scale over the
 * channel chunks; three products.  Cout <= 4096 (64-column tiles / slices) or a multiple of 180 (192-column tiles), Cin <= 4096.  What the weight preparation emits by default
 * for these shapes (SRHIP_F16X2_CONV=0: bf16x3); arguments as srhip_conv3x3_nhwc_bx3. */
int srhip_conv3x3_nhwc_f16x2(const float* X, long ldx, const void* Wh, const float* bias, float* Y, long ldy,
                             int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                             const float* rowscale, float alpha, void* stream);
/* The split-operand conv (wfmt 0: srhip_conv3x3_nhwc_bx3's weight, 1: srhip_conv3x3_nhwc_f16x2's) with what the plain CNNs of
 * the evaluation sweep put around it folded into its prologue / epilogue (one pass over the feature map less per fold):
 *   in_bn_coef [4][Cin] = running mean, 1/sqrt(var+eps), gamma/sqrt(var+eps), beta: evaluation-mode BatchNorm2d + ReLU on
 *          the INPUT (zero padding applied to the activation), MemNet's BN-ReLU-conv (network_memnet.py:27-34); the
 *          expression of srhip_bn_apply; wfmt 1, Cout a multiple of 64, <= 4096
 *   epi 8: relu(R + s*(acc+bias))                    DRRN's residual unit (network_drrn.py:58-62)
 *   epi 9: prelu(acc+bias), slope = *slope (device)  DBPN's ConvBlock / DeconvBlock activation (network_dbpn.py:16-60)
 *   epi 10: prelu(acc+bias) + alpha * R              ... and the projection units' l0 - x / h1 + h0 (:93-99,128-134)
 *   epi 11: gelu(acc+bias), exact erf                DFCAN's conv + nn.GELU() (network_dfcan.py:44-47,98-99) */
int srhip_conv3x3_nhwc_split_ex(int wfmt, const float* X, long ldx, const void* Wp, const float* bias, float* Y, long ldy,
                                int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                                const float* rowscale, float alpha, const float* in_bn_coef, const float* slope,
                                void* stream);
/* One EDSR ResBlock (64 -> 64 -> 64 channels) per launch (resblock.hip) -- ResBlock.forward,
 * dlib/models/network_nlsn.py:72-93 (conv -> ReLU -> conv, .mul(res_scale), += x) and its data gradient:
 *   forward :  a  = relu(conv3x3(x; W1) + b1)  [B][H][W][lda]   (saved: the weight gradients and the backward read it)
 *              out = x + res_scale * (conv3x3(a; W2) + b2)
 *   backward:  da = res_scale * conv3x3(g; W2^T) * (a > 0)      (the first conv's weight gradient reads it)
 *              dx = g + conv3x3(da; W1^T)
 * The second conv runs from LDS on the first one's tile (+ a one-pixel ring, recomputed per tile): one launch, one halo
 * fetch, no read-back of a / da.  W*h: fp16x2 conv operands of srhip_prep_table kind 4 for 64 -> 64 channels (the
 * backward takes the data-gradient forms: flipped taps, transposed channels -- what srhip_conv3x3_nhwc_f16x2 takes for
 * the same job).  Arithmetic of srhip_conv3x3_nhwc_f16x2: two fp16 planes, three products, f32 accumulate; the mid tile
 * under one power-of-two exponent of its own.  out / a must not alias x (dx / da: g). */
int srhip_resblock64_fwd_f16x2(const float* x, long ldx, const void* W1h, const float* b1, const void* W2h, const float* b2,
                               float res_scale, float* a, long lda, float* out, long ldout, int B, int H, int W, void* stream);
int srhip_resblock64_bwd_f16x2(const float* g, long ldg, const void* W2Th, const void* W1Th, const float* a, long lda,
                               float res_scale, float* da, long ldda, float* dx, long lddx, int B, int H, int W, void* stream);

/* ---- fp16-STORAGE inference of the plain conv family (--amp at evaluation time; eval_all.sh's sweep, select_network.py:52-210):
 * activations NHWC fp16 in HBM, the weight = the leading fp16 plane of the srhip_conv3x3_nhwc_f16x2 operand (job kind 4; ps2:
 * mode 12), ONE fp16 MFMA product, f32 accumulate, fp16 out.  Cin a multiple of 32, Cout of 64 (ps2: of 256), pitches multiples
 * of 8 halves.  epi 0 +bias | 1 relu | 2 R + alpha*(acc+bias) | 6 leaky relu(alpha) | 8 relu(R + alpha*(acc+bias)) (R fp16, laid
 * out as Y).  in_bn_coef [4][Cin] (may be NULL): evaluation-mode BatchNorm + ReLU on the input, as srhip_conv3x3_nhwc_split_ex.
 * center_only: the weight is a 1x1 conv held in the centre tap of a 3x3 (the other taps are not multiplied).
 * ps2: the result goes through PixelShuffle(2) into Y [B][2H][2W][Cout/4] (network_nlsn.py:100-118).  Replaces
 * nn.Conv2d + ReLU / ResBlock of network_vdsr.py:24-60, network_drrn.py:22-62, network_nlsn.py:72-128 under torch.autocast-like
 * reduced precision (the reference has no such mode: the PSNR gate of tests/test_gpu_amp.py is the contract). */
int srhip_conv3x3_nhwc_h16(const void* X, long ldx, const void* Wh, const float* bias, void* Y, long ldy, int B, int H, int W,
                           int Cin, int Cout, int epi, const void* R, long ldr, float alpha, int ps2, const float* in_bn_coef,
                           int center_only, void* stream);
/* the 1 -> Cout conv at the head: f32 image [B][H][W] in, fp16 features out (w [Cout][1][3][3] f32; act 0 none | 1 ReLU | 2
 * LeakyReLU(alpha)) */
int srhip_conv3x3_cin1_h16(const float* x, const float* w, const float* bias, void* y, long ldy, int B, int H, int W, int Co,
                           int act, float alpha, void* stream);
/* the Cin -> 1 conv at the tail: fp16 features in (optionally through BatchNorm-ReLU, in_bn_coef [4][Cin]), f32 image
 * [B][H][W] out, + bias + the f32 image `add` (may be NULL) */
int srhip_conv3x3_cout1_h16(const void* x, long ldx, const float* w, const float* bias, const float* add, const float* in_bn_coef,
                            float* y, int B, int H, int W, int Ci, void* stream);
/* SRCNN's evaluation forward under --amp in ONE kernel (network_srcnn.py:23-69; fp16 products, f32 accumulate): patches [T][32]
 * fp16 = the 5x5 patch matrix (25 taps + 7 zero columns) -- or NULL: built inside from image [B][H][W] f32 (T = B*H*W) --, W1h / W2h = the fp16x2 conv operands (job kind 4) of the layers held as
 * centre taps of 3x3 weights ([1024][32] and [128][1024]), w3 [128], b3 [1] f32 -> y [T] f32.  The 1024-channel feature map is
 * never written: a block walks it in 64-channel chunks through LDS. */
int srhip_srcnn_fwd_h16(const void* patches, const float* image, int B, int H, int W, const void* W1h, const float* b1,
                        const void* W2h, const float* b2, const float* w3, const float* b3, float* y, long T, void* stream);
/* srhip_conv3x3_ps2_bx3 / srhip_conv3x3_ps2_bwd_data_bx3 with the weight in that format (job kind 4, modes 12 / 16). */
int srhip_conv3x3_ps2_f16x2(const float* X, long ldx, const void* Wh, const float* bias, float* Yup, long ldy,
                            int B, int H, int W, int Cin, int Cout, int epi, float alpha, void* stream);
int srhip_conv3x3_ps2_bwd_data_f16x2(const float* dYup, long lddy, const void* Wht, float* dX, long ldx, int B, int H,
                                     int W, int Cout, int Cin, int epi, const float* R, long ldr, float alpha,
                                     void* stream);
/* The MLP half of a Swin block in one kernel per direction (mlp_f16.hip): the hidden activation goes from the first
 * product's accumulators through registers and LDS into the second product and is never read back from HBM.
 *   forward : h = LN(x) . W1^T + b1 (stats[M][2] = {mean, rstd} of x; W1 gamma-folded, b1 beta-folded),
 *             out = x + s * (gelu(h) . W2^T + b2), stats_out = {mean, rstd} of the out rows (may be NULL);
 *             h may be NULL (inference), otherwise it is what the backward reads.
 *   backward: dh = (s * dy . W2) * gelu'(h), gh = gelu(h)  (both [M][ldh], operands of the weight gradients),
 *             dx = dy + LayerNorm-backward(dh . W1)  (x, stats as in the forward).  gh may be NULL: the fc2 weight
 *             gradient then takes h itself with b_mode 2 (srhip_tn_problem: gelu applied in the operand prologue, the
 *             same x Phi(x) bit for bit) -- 4 * M * hidden bytes less written per block.
 * Operands as the Linear GEMMs take them: two fp16 planes + per-row power-of-two scales (srhip_prep_table kind 3,
 * no permutation), three products; rowscale = DropPath multipliers per sample.  W1h = planes of W1*gamma
 * [hidden][C], W2h = planes of W2 [C][hidden]; backward W2Th = planes of W2^T [hidden][C], W1Th = planes of
 * (W1*gamma)^T [C][hidden].  C <= 192, hidden <= 384, both multiples of 4; row pitches multiples of 4 floats.
 * Replaces Mlp.forward + residual (dlib/models/network_swinir.py:28-45,335-337) and its autograd. */
int srhip_mlp_fwd_f16x2(const float* x, long ldx, const float* stats, const void* W1h, const float* b1,
                        const void* W2h, const float* b2, float* h, long ldh, float* out, long ldo, int M, int C,
                        int hidden, const float* rowscale, int rows_per_scale, float* stats_out, void* stream);
int srhip_mlp_bwd_f16x2(const float* dy, long lddy, const void* W2Th, const void* W1Th, const float* h, long ldh,
                        float* dh, float* gh, const float* x, long ldx, const float* stats, float* dx, long lddx,
                        int M, int C, int hidden, const float* rowscale, int rows_per_scale, void* stream);
/* srhip_mlp_bwd_f16x2 with a third product chained behind it in the same kernel: out3 = s3 * (dx . W3^T), the data
 * gradient of the Linear whose output feeds this block's residual (the attention's proj: W3h = planes of Wproj^T [C][C],
 * rowscale3 = the DropPath multipliers of the attention branch, same rows_per_scale).  dx goes from the registers that
 * hold it into the third product's operand images: one launch, one read of dx less.  out3 [M][ld3], != dx. */
int srhip_mlp_bwd_chain_f16x2(const float* dy, long lddy, const void* W2Th, const void* W1Th, const float* h, long ldh,
                              float* dh, float* gh, const float* x, long ldx, const float* stats, float* dx, long lddx,
                              int M, int C, int hidden, const float* rowscale, int rows_per_scale, const void* W3h,
                              float* out3, long ld3, const float* rowscale3, void* stream);
/* ... and with a product in FRONT of it as well: the MLP's incoming gradient is itself computed in the kernel,
 *   dy = res0 + LayerNorm_backward(X0 . W0^T; x0, stats0)     (written to dy [M][lddy]: the weight gradients read it)
 * -- the data gradient of the qkv Linear of the Swin block BEHIND this one (X0 = dqkv [M][K0], W0h = planes of
 * (Wqkv*gamma)^T [C][K0], x0 / stats0 = that block's input rows and their {mean, rstd}, res0 = the gradient arriving
 * at its first residual), what srhip_gemm_nt_f16x2_lnbwd computes as its own launch.  One kernel then carries the
 * data-gradient chain from one block's attention backward to the next one's: K0 in passes of 192 under a running row
 * exponent, the rows go from registers into the operand images of the MLP backward.  W3h may be NULL (no chained
 * product).  dy must not alias dx or res0. */
int srhip_mlp_bwd_front_chain_f16x2(const float* X0, long ld0, int K0, const void* W0h, const float* x0, long ldx0,
                                    const float* stats0, const float* res0, long ldres0, float* dy, long lddy,
                                    const void* W2Th, const void* W1Th, const float* h, long ldh, float* dh, float* gh,
                                    const float* x, long ldx, const float* stats, float* dx, long lddx, int M, int C,
                                    int hidden, const float* rowscale, int rows_per_scale, const void* W3h, float* out3,
                                    long ld3, const float* rowscale3, void* stream);
/* The W-MSA half of a Swin block, forward, in one kernel (wmsa_f16.hip): one thread block per 8x8 window computes
 *   qkv = LN(x) . Wqkv^T + bqkv  (stats[T][2] = {mean, rstd} of x; Wqkv gamma-folded, bqkv beta-folded),
 *   att = softmax(q k^T / sqrt(D) + bias + shift mask) v  per head (cyclic shift 0 or 4, as srhip_window_attention_fwd_f16x2),
 *   out = x + s * (att . Wproj^T + bproj),  stats_out = {mean, rstd} of the out rows (may be NULL).
 * qkv [T][3C] and att [T][C] are written for the backward.  With 5 or 6 heads a wave owns one head from the qkv GEMM to the
 * attention output: q, k, v never leave its registers, qkv is only written and may be NULL (inference); other head counts
 * read their qkv rows back from L2 after a barrier.  x, out: token-major
 * [B*H*W][C], dense; out must not alias x.  Weight planes as srhip_mlp_fwd_f16x2 (prep kind 3); biasF from
 * srhip_bias_expand_f16x2; rowscale = DropPath multipliers per sample [B] or NULL.  C <= 192 (multiple of 4),
 * heads <= 8, head dim in {10, 16, 30, 32}, H, W multiples of 8.
 * Replaces norm1 + roll + window_partition + WindowAttention.forward + window_reverse + roll + residual
 * (dlib/models/network_swinir.py:288-334,153-176). */
int srhip_wmsa_fwd_f16x2(const float* x, const float* stats, const void* Wqkvh, const float* bqkv, const void* Wprojh,
                         const float* bproj, const float* biasF, const float* rowscale, float* qkv, float* att,
                         float* out, float* stats_out, int B, int H, int W, int C, int heads, int shift,
                         void* stream);
int srhip_conv3x3_nhwc_bx3(const float* X, long ldx, const void* Wb, const float* bias, float* Y, long ldy,
                           int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                           const float* rowscale, float alpha, void* stream);

/* 3x3 conv + PixelShuffle(2) as ONE kernel per direction -- the Upsampler stage of EDSR and of SwinIR's
 * 'pixelshuffle' tail (dlib/models/network_nlsn.py:89-93,103-108; network_swinir.py:857-870): the
 * [B][H][W][4F] conv result is never materialised.
 *   srhip_conv3x3_ps2_bx3          Yup [B][2H][2W][F] (ldy = its pixel pitch) = PixelShuffle(2)(conv(X) + bias)
 *                                  (epi 0 | 1 relu | 6 leaky relu(alpha)); Cout = 4F
 *   srhip_conv3x3_ps2_bwd_data_bx3 dX [B][H][W][Cin] = data gradient of that conv read from dYup [B][2H][2W][F]
 *                                  (epi 0 | 4: * (R > 0) | 7: * (R > 0 ? 1 : alpha), R [B][H][W][Cin] = the activation
 *                                  that fed the conv)
 *   srhip_conv3x3_ps2_wgrad_bx3    partial weight gradients (as srhip_conv3x3_wgrad_bx3, torch channel order:
 *                                  srhip_reduce_conv_wgrad applies unchanged) with dY read from dYup
 * Weight planes with the conv's output channels in sub-pixel-major order sp*F + c <- torch channel c*4 + sp,
 * sp = 2*i + j: srhip_prep_table kind 0 with mode 12 (rows: forward pack) / 16 (k: data-gradient pack).
 * F a multiple of 32. */
int srhip_conv3x3_ps2_bx3(const float* X, long ldx, const void* Wb, const float* bias, float* Yup, long ldy,
                          int B, int H, int W, int Cin, int Cout, int epi, float alpha, void* stream);
int srhip_conv3x3_ps2_bwd_data_bx3(const float* dYup, long lddy, const void* Wbt, float* dX, long ldx, int B, int H,
                                   int W, int Cout, int Cin, int epi, const float* R, long ldr, float alpha,
                                   void* stream);
int srhip_conv3x3_ps2_wgrad_bx3(const float* dYup, long lddy, const float* X, long ldx, int B, int H, int W,
                                int Cout, int Cin, float* part, float* part_colsum, int S, void* stream);

/* Weight gradients: out[i][j] = sum_m A[m][i] * pro(B)[m][j], reduce dimension
 * split in S slices written to part[S][(9)][NI][NJ] (+ column sums of A in
 * part_colsum[S][NI] for the bias gradient); srhip_reduce_* sums the slices.
 * srhip_tn_plan returns S and the size of `part` in floats. */
int srhip_tn_plan(int M, int NI, int NJ, int conv, int* S, long* part_floats);
int srhip_gemm_tn(const float* A, long lda, const float* B, long ldb, int M, int NI, int NJ,
                  const float* a_rowscale, int a_rowscale_rows, int b_mode, const float* ln_stats,
                  float* part, float* part_colsum, int S, void* stream);
/* Several Linear weight-gradient problems over the same M rows in one launch, each with its own partial buffers:
 * up to 4 (exact-f32 kernels: the four Linears of a Swin block) or up to 24 (_bx3: the 4 x depth Linears of a whole
 * RSTB layer -- more tiles per launch need fewer reduce slices S to fill the chip). */
typedef struct srhip_tn_problem {
  const float* A; long lda;          /* dY [M][NI] */
  const float* B; long ldb;          /* X  [M][NJ] */
  int NI, NJ;
  const float* a_rowscale; int a_rowscale_rows;
  int b_mode; const float* ln_stats;
  float* part;                       /* [S][NI][NJ] */
  float* part_colsum;                /* [S][NI] or NULL */
} srhip_tn_problem;
int srhip_tn_tiles(int NI, int NJ);
int srhip_tn_group_plan(int M, int ntiles, int* S);
int srhip_gemm_tn_grouped(const srhip_tn_problem* probs, int nprob, int M, int S, void* stream);
int srhip_conv3x3_wgrad(const float* dY, long lddy, const float* X, long ldx, int B, int H, int W,
                        int Cout, int Cin, float* part, float* part_colsum, int S, void* stream);

/* The three weight-gradient contractions above on the bf16 MFMA with 3-way split
 * operands (see srhip_gemm_nt_bx3): same arguments, alignment rules, slicing and
 * reducers; plan S with the _bx3 planners (one 8-wave block per CU).  The 3x3 conv forms with Cout and Cin multiples of
 * 64 (three taps per block) split their operands into TWO fp16 planes under a running power-of-two scale per operand
 * column and issue three products -- same f32-grade sums; SRHIP_TN_F16X2=0 in the environment: three bf16 planes, six.
 * Round 6: 3x3 conv problems of at least 64 channels on either side, on images whose width is a multiple of 64, run in the
 * strip form (all nine taps per block, power-of-two scales fixed per block, a second pass for the blocks whose guess did not
 * hold); part_floats of the conv plan is the partial sums [S][9][NI][NJ] PLUS that kernel's per-block words behind them --
 * allocate what the plan says, the reducers read the first S*9*NI*NJ floats. */
int srhip_tn_plan_bx3(int M, int NI, int NJ, int conv, int* S, long* part_floats);
int srhip_tn_group_plan_bx3(int M, int ntiles, int* S);
int srhip_gemm_tn_bx3(const float* A, long lda, const float* B, long ldb, int M, int NI, int NJ,
                      const float* a_rowscale, int a_rowscale_rows, int b_mode, const float* ln_stats,
                      float* part, float* part_colsum, int S, void* stream);
int srhip_gemm_tn_grouped_bx3(const srhip_tn_problem* probs, int nprob, int M, int S, void* stream);
int srhip_conv3x3_wgrad_bx3(const float* dY, long lddy, const float* X, long ldx, int B, int H, int W,
                            int Cout, int Cin, float* part, float* part_colsum, int S, void* stream);
int srhip_reduce_linear_wgrad(const float* part, const float* colsum, int S, float* dW, float* db,
                              int N, int K, void* stream);
/* Linear fed by a folded LayerNorm: also emits dgamma / dbeta of that norm.  Deterministic: the row blocks'
 * shares go to ln_ws (srhip_ln_affine_ws(N, K) floats, caller-owned) by plain stores and a second small launch
 * adds them in a fixed order -- no atomics, nothing to zero. */
long srhip_ln_affine_ws(int N, int K);
int srhip_reduce_ln_linear_wgrad(const float* part, const float* colsum, int S, const float* W,
                                 const float* gamma, const float* beta, float* dW, float* db,
                                 float* dgamma, float* dbeta, int N, int K, float* ln_ws,
                                 void* stream);
/* dW in torch layout [Cout][Cin][3][3]. */
/* The slice reducers of up to 24 Linear problems (srhip_gemm_tn_grouped / _bx3) in one launch.
 * gamma == NULL: plain Linear (srhip_reduce_linear_wgrad); else the LayerNorm-folded form
 * (srhip_reduce_ln_linear_wgrad; ln_ws = its workspace of srhip_ln_affine_ws(N, K) floats, one per problem). */
typedef struct {
  const float* part; const float* colsum;
  const float* W; const float* gamma; const float* beta;
  float* dW; float* db; float* dgamma; float* dbeta;
  int N, K;
  float* ln_ws;
} srhip_reduce_problem;
int srhip_reduce_wgrad_grouped(const srhip_reduce_problem* probs, int nprob, int S, void* stream);
int srhip_reduce_conv_wgrad(const float* part, const float* colsum, int S, float* dW, float* db,
                            int Co, int Ci, void* stream);

/* ---- weight preparation (once per optimizer step) --------------------------- */
/* Wf = W*gamma (per column), bf = b + W.beta: LayerNorm affine folded into the
 * Linear that consumes it. */
int srhip_fold_layernorm(const float* W, const float* b, const float* gamma, const float* beta,
                         float* Wf, float* bf, int N, int K, void* stream);
int srhip_transpose(const float* in, float* out, int R, int C, void* stream);
/* torch [Cout][Cin][3][3] -> wp [9][Cout][Cin] and/or wpt [9][Cin][Cout] (taps flipped). */
int srhip_pack_conv_weight(const float* w, float* wp, float* wpt, int Co, int Ci, void* stream);

/* ---- LayerNorm (nn.LayerNorm(C), eps 1e-5; network_swinir.py:240,248,606,846) - */
/* stats[M][2] = mean, rstd (optional); y = LN(x)*gamma+beta (optional). */
int srhip_layernorm_fwd(const float* x, float* stats, float* y, const float* gamma, const float* beta,
                        long M, int C, void* stream);
/* out = res + dx.  gamma == NULL: dy is the gradient w.r.t. the normalised
 * value; else dy is w.r.t. y and dgamma/dbeta are produced (deterministically: per-block column sums in
 * workspace -- srhip_layernorm_bwd_ws(M, C) floats, caller-owned -- added in block order by a second launch). */
long srhip_layernorm_bwd_ws(long M, int C);
int srhip_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* res,
                        const float* gamma, float* out, float* dgamma, float* dbeta, float* workspace, long M,
                        int C, void* stream);

/* ---- cv2.resize(INTER_CUBIC) on 1-channel images (resize.hip): the 'l_to_h_img' tensors of the dataset --------- */
/* dlib/datasets/dataset_dpsr.py:659-683 (_resize_low_to_scale), consumed by the SRCNN-style nets (model_plain.py:184-
 * 195).  src / dst: B images [H][W] -> [Ho][Wo], uint8 (is_u8: OpenCV's fixed-point path) or float32.  A restatement of
 * OpenCV's published algorithm (cv2 is absent here): PARITY UNPINNED against cv2; bit-exact / 1e-6 against
 * oracle/cv2_cubic.py.  srhip_u8_to_unit: uint2single (utils_image.py:322-323); srhip_clip01: np.clip(x, 0, 1). */
int srhip_resize_cubic(const void* src, void* dst, int is_u8, int B, int H, int W, int Ho, int Wo, void* stream);
int srhip_u8_to_unit(const unsigned char* src, float* dst, long n, void* stream);
int srhip_clip01(float* x, long n, void* stream);

/* ---- helpers of the generic conv-net engine (tape_ops.hip; DBPN / SRFBN / ProSR: SURVEY f1) ---------------- */
/* nn.PReLU(num_parameters = 1) (dlib/models/network_dbpn.py:85-86, network_srfbn.py:38-46): y = x > 0 ? x : a x with
 * the slope read from device memory; backward dx = g (x > 0 ? 1 : a) (dx may alias g) and dalpha (+)= sum g min(x, 0)
 * (fp64 block partials in workspace -- 4096 doubles -- added in a fixed order: deterministic). n % 4 == 0. */
int srhip_prelu_fwd(const float* x, const float* alpha, float* y, long n, void* stream);
int srhip_prelu_bwd(const float* g, const float* x, const float* alpha, float* dx, float* dalpha, double* workspace,
                    long n, int accumulate, void* stream);
/* y[r][0:cols] = a x[r][0:cols] + b y[r][0:cols] on row-major views (channel slices of NHWC tensors: torch.cat as a
 * view, network_dbpn.py:540-566); cols, ldy, ldx multiples of 4. */
int srhip_axpby2d(float* y, long ldy, const float* x, long ldx, long rows, int cols, float a, float b, void* stream);
/* nn.ReflectionPad2d(1) on NHWC (network_prosr.py:44-86): in [B][H][W][C] -> out [B][H+2][W+2][C]; adjoint != 0: its
 * gradient, in [B][H+2][W+2][C] -> out [B][H][W][C] (a gather, deterministic).  srhip_crop1: the inverse crop
 * [B][H+2][W+2][C] -> [B][H][W][C]; adjoint: zero-bordered placement. */
int srhip_pad_reflect1(const float* in, float* out, int B, int H, int W, int C, int adjoint, void* stream);
int srhip_crop1(const float* in, float* out, int B, int H, int W, int C, int adjoint, void* stream);
/* The element-wise pieces of ENLCA (Efficient Non-Local Contrastive Attention, network_enlcn.py:207-366) around its dense
 * products, token rows [T][.], in place where noted:
 *   srhip_l2norm_rows        x[t] <- k * x[t] / max(|x[t]|_2, eps)                 (F.normalize(dim=channel) * sqrt(6), :341-342)
 *   srhip_performer_features dash[t][j] <- ratio * (exp(dash[t][j] - |data[t]|^2 / 2) + eps)   (softmax_kernel :207-240)
 *   srhip_enlca_finish       out[t][c] = x[t][c] + res_scale * num[t][c] / num[t][Cy]  (linear_attention :243-248 with the
 *                            normaliser carried as column Cy of the numerator product, + the residual :364-365) */
int srhip_l2norm_rows(float* x, long ld, long T, int C, float eps, float k, void* stream);
int srhip_performer_features(float* dash, long ldd, const float* data, long ldx, long T, int F, int C, float ratio, float eps,
                             void* stream);
int srhip_enlca_finish(const float* num, long ldn, const float* x, float* out, long T, int Cy, float res_scale, void* stream);
/* Their autograd (ENLCA trains through the reference's one step, model_plain.py:318-396; the dense products in between are
 * srhip_gemm_nt* / srhip_linear_wgrad*):
 *   srhip_l2norm_rows_train      as srhip_l2norm_rows, also factors[t] = k / max(|x[t]|, eps)
 *   srhip_l2norm_rows_bwd        dy[t] <- f (dy[t] - y[t] (y[t] . dy[t]) / k^2)  (f dy[t] where the norm was clamped), in place
 *   srhip_performer_features_bwd g <- g (f - ratio eps): the gradient with respect to dash, in place (n = elements, % 4 == 0)
 *   srhip_enlca_finish_bwd       dnum[t][c] = s dout[t][c] (c < Cy), dnum[t][Cy] = -sum_c s dout[t][c] num[t][c] / num[t][Cy],
 *                                s = res_scale / num[t][Cy]; columns past Cy zero */
int srhip_l2norm_rows_train(float* x, long ld, long T, int C, float eps, float k, float* factors, void* stream);
int srhip_l2norm_rows_bwd(float* dy, long ldd, const float* y, long ldy, const float* factors, long T, int C, float eps, float k,
                          void* stream);
int srhip_performer_features_bwd(float* g, const float* f, long n, float ratio_eps, void* stream);
int srhip_enlca_finish_bwd(const float* dout, const float* num, long ldn, float* dnum, long T, int Cy, float res_scale, void* stream);

/* ---- Non-Local Sparse Attention of NLSN, evaluation forward (nlsa.hip) --------------
 * NonLocalSparseAttention.forward, dlib/models/network_nlsn.py:131-268, token-major (channels last):
 *   srhip_nlsa_hash_buckets   min(L // chunk + (L // chunk) % 2, 128)                                     (:193-194)
 *   srhip_nlsa_order          hash codes = argmax over cat([r, -r]) of the rotated embedding (first maximum, :158-161;
 *                             rotated [N*L][ld] = x_embed . rotations, columns round-major) and the tokens of every
 *                             (sample, round) ordered by code -- ONE stable radix sort of 64-bit keys (low 20 bits =
 *                             token), where the reference's torch.sort leaves the order of equal codes open (:199-201).
 *                             keys_tmp, order: N * n_hashes * L entries; workspace: srhip_nlsa_sort_ws(items) bytes.
 *   srhip_nlsa_attention      every chunk of chunk_size ordered tokens attends to itself and its two neighbouring chunks
 *                             (keys L2-normalised, eps 5e-5; queries not), log-sum-exp scores, attention-weighted sum of
 *                             the assembly embedding; result and score are written at the token's own position of its
 *                             round (ret [N][n_hashes][L][Cy], score [N][n_hashes][L]: scratch), then
 *                             out = x + res_scale * sum_h softmax_h(score) ret_h                          (:209-266).
 *                             Ce <= 64; chunk_size such that the tile fits 160 KB of LDS (144: 135 KB). */
int srhip_nlsa_hash_buckets(int L, int chunk_size);
long srhip_nlsa_sort_ws(long n_items);
int srhip_nlsa_order(const float* rotated, long ld, unsigned long long* keys_tmp, unsigned long long* order, void* workspace,
                     long ws_bytes, int N, int L, int n_hashes, int hash_buckets, void* stream);
int srhip_nlsa_attention(const float* x_embed, const float* y_embed, const unsigned long long* order, float* ret, float* score,
                         const float* x, float* out, int N, int L, int Ce, int Cy, int n_hashes, int chunk_size,
                         float res_scale, void* stream);

/* ---- Fourier channel attention of DFCAN, evaluation forward (dfca.hip) --------------
 * RCAB.forward, dlib/models/network_dfcan.py:39-70, channels last:
 *   srhip_fft2_mag_pow_shift  out = fftshift2d((|fftn(x, dim=(H, W))| + eps)^gamma)  (:60-64,27-36) as a separable DFT with
 *                             f64 accumulation; x, out [B][H][W][C]; workspace 2*B*H*W*C floats; H, W <= 256.
 *   srhip_channel_gate        gate = sigmoid(W2 act(W1 avgpool(feat) + b1) + b2), out = x0 + x1 * gate  (:65-70; also RCAN's
 *                             CALayer network_act.py:230-247 and OmniSR's SqueezeExcitation network_omni_sr.py:133-148:
 *                             mid_act 1 = SiLU, b1 / b2 / x0 NULL); the average is a two-stage fixed-order sum
 *                             (workspace: srhip_channel_gate_ws doubles; gate: B*C floats).
 *   srhip_unary               kind 0: nn.GELU() (exact erf), kind 1: sigmoid -- the activations behind DFCAN's convs
 *                             (:44-47,98-99,108,111-113); out may alias x. */
int srhip_fft2_mag_pow_shift(const float* x, float* out, float* workspace, int B, int H, int W, int C, float gamma, float eps,
                             void* stream);
/* its backward (training, Tape.fourier_gate; the reference differentiates torch.fft through autograd, network_dfcan.py:60-64):
 * dx = Re(unnormalised IFFT2(G . FFT2(x))), G = unshifted g . gamma (|F| + eps)^(gamma - 1) / |F| (0 where |F| = 0), as four
 * separable DFT passes with f64 accumulation; x, g, dx [B][H][W][C]; workspace 2*B*H*W*C floats; H, W <= 256. */
int srhip_fft2_mag_pow_shift_bwd(const float* x, const float* g, float* dx, float* workspace, int B, int H, int W, int C,
                                 float gamma, float eps, void* stream);
long srhip_channel_gate_ws(int B, long P, int C);
int srhip_channel_gate(const float* feat, const float* w1, const float* b1, const float* w2, const float* b2, const float* x0,
                       const float* x1, float* out, float* gate, double* workspace, int B, long P, int C, int Cm,
                       int mid_act, void* stream);
int srhip_unary(const float* x, float* out, long n, int kind, void* stream);
/* its backward: dx = g * f'; kind 0 (GELU): xy = the op's input, kind 1 (sigmoid): xy = its output */
int srhip_unary_bwd(const float* xy, const float* g, float* dx, long n, int kind, void* stream);

/* ---- token-branch pieces of ACT, evaluation forward (act_ops.hip) --------------
 * dlib/models/network_act.py:468-541 on channels-last data:
 *   srhip_unfold / srhip_fold   F.unfold(x, k, stride, padding) / F.fold(tok, (H, W), k, stride) between the feature map
 *                               (pixels of ldx / ldo floats, C channels used) and the token matrix (rows of ldt floats,
 *                               columns c*k*k + ky*k + kx); fold is a gather (deterministic), uncovered pixels get 0.
 *                               unfold with k = 5, s = 1, pad = 2 is the im2col of the 5 x 5 head convs (:362-364).
 *   srhip_layernorm_rows        nn.LayerNorm with affine over rows of any width (:115-133); y may alias x.
 *   srhip_softmax_rows          x[r] <- softmax(scale * x[r]) in place (:173-178,212-217). */
int srhip_unfold(const float* x, long ldx, float* tok, long ldt, int B, int H, int W, int C, int k, int s, int pad,
                 void* stream);
int srhip_fold(const float* tok, long ldt, float* out, long ldo, int B, int H, int W, int C, int k, int s, void* stream);
int srhip_layernorm_rows(const float* x, long ldx, float* y, long ldy, const float* gamma, const float* beta, long M, int C,
                         float eps, void* stream);
/* Backward of srhip_layernorm_rows (nn.LayerNorm's autograd, network_act.py:115-133 in training): dx,
 * dgamma[C], dbeta[C]; rows of at most 2048 values; ws: srhip_layernorm_rows_bwd_ws(M, C) floats; deterministic. */
long srhip_layernorm_rows_bwd_ws(long M, int C);
int srhip_layernorm_rows_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma, float* dx, long lddx,
                             float* dgamma, float* dbeta, float* ws, long M, int C, float eps, void* stream);
/* y = res + LayerNorm(x), rows of at most 256 values (GRL's post-norm residuals, network_grl.py:1061-1076); y may alias x / res */
int srhip_layernorm_rows_res(const float* x, long ldx, const float* res, long ldr, float* y, long ldy, const float* gamma,
                             const float* beta, long M, int C, float eps, void* stream);
int srhip_softmax_rows(float* x, long ld, long R, int n, float scale, void* stream);
/* Row softmax for training paths (NLSN's chunk attention, network_nlsn.py:236-243, through autograd):
 *   srhip_softmax_rows_lse   as srhip_softmax_rows, also lse[r] = log sum_j exp(scale x[r][j])   (the bucket score :238)
 *   srhip_softmax_rows_bwd   dP[r][j] <- P[r][j] (dP[r][j] - sum_k P[r][k] dP[r][k] + dlse[r]) in place: the gradient with
 *                            respect to the logits of P = softmax and of lse (dlse may be NULL)
 *   srhip_rowdot             out[r] = sum_c a[r][c] b[r][c] */
int srhip_softmax_rows_lse(float* x, long ld, long R, int n, float scale, float* lse, void* stream);
int srhip_softmax_rows_bwd(const float* P, float* dP, long ld, long R, int n, const float* dlse, void* stream);
int srhip_rowdot(const float* a, long lda, const float* b, long ldb, float* out, long R, int n, void* stream);

/* ---- pieces of OmniSR's omni self-attention blocks, evaluation forward (omni_ops.hip) --------------
 * dlib/models/network_omni_sr.py, channels last:
 *   srhip_dwconv3x3          nn.Conv2d(C, C, 3, padding=1, groups=C) (:178,318,348); w [C][9], bias may be NULL
 *   srhip_group_attention    softmax(scale q k^T + bias) v per (group of n <= 64 consecutive token rows, head) of
 *                            qkv [groups*n][3C] (Attention.forward :258-306 on window- or grid-ordered tokens); bias
 *                            [heads][n][n] = rel_pos_bias(rel_pos_indices) or NULL
 *   srhip_channel_attention  Channel_Attention / Channel_Attention_grid.forward (:353-428) on qkv [B][H][W][3C]
 *   srhip_gelu_gate          gelu(x1) * x2 of the two halves of a [T][2C] matrix (:325-326)
 *   srhip_maxpool2d, srhip_bilinear_resize, srhip_mul_sigmoid   ESA's F.max_pool2d(7, 3), F.interpolate(bilinear,
 *                            align_corners=False) and x * sigmoid(c4) (:104-114) */
int srhip_dwconv3x3(const float* x, long ldx, const float* w, const float* bias, float* out, long ldo, int B, int H, int W, int C,
                    void* stream);
int srhip_group_attention(const float* qkv, const float* bias, float* out, long groups, int n, int C, int heads, float scale,
                          void* stream);
int srhip_channel_attention(const float* qkv, const float* temperature, float* out, int B, int H, int W, int C, int heads, int ps,
                            int grid, void* stream);
int srhip_gelu_gate(const float* x, float* out, long T, int C, void* stream);
int srhip_maxpool2d(const float* x, float* out, int B, int H, int W, int C, int k, int s, void* stream);
int srhip_bilinear_resize(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, void* stream);
int srhip_mul_sigmoid(const float* x, const float* g, float* out, long n, void* stream);
/* Pieces the backward of OmniSR's training graph is composed from (srhip/omnisr_engine.py::_forward_tape; autograd of
 * network_omni_sr.py:85-114,243-306): srhip_mul out = a * b; srhip_add_periodic x[i] += v[i % period] (the relative-position
 * bias of every window, :291-294) and its adjoint srhip_sum_periodic out[j] = sum_k x[k period + j] (fixed order);
 * srhip_maxpool2d_bwd the gradient of srhip_maxpool2d as a gather (first maximum of a window, as torch). */
int srhip_mul(const float* a, const float* b, float* out, long n, void* stream);
int srhip_add_periodic(float* x, const float* v, long n, long period, void* stream);
int srhip_sum_periodic(const float* x, float* out, long n, long period, void* stream);
int srhip_maxpool2d_bwd(const float* x, const float* g, float* dx, int B, int H, int W, int C, int k, int s, void* stream);

/* ---- pieces of GRL's mixed-attention blocks, evaluation forward (grl_ops.hip) ----------------------
 * dlib/models/network_grl.py, channels last:
 *   srhip_avgpool2d        nn.AvgPool2d(k, k) of AnchorLinear (:603-620): [B,H,W,C] -> [B,H/k,W/k,C]
 *   srhip_cpb_bias         AffineTransform's bias (:305-311): biasT[h][j][i] = 16 sigmoid(table[index[i][j]][h]) from the CPB
 *                          MLP's output table [entries][heads] and a relative-position index [N1][N2] (int64, as the
 *                          module's registered buffers); key-major so a wave of queries reads consecutive words
 *   srhip_cosine_window_attention   Attention.attn (:338-355) under AffineTransform.forward (:296-319):
 *                          softmax(exp(min(logit_scale, log 100)) cos(q, k) + bias + mask) v per (window, head), between the
 *                          qwh x qww windows of a [B,qH,qW,.] image and the kwh x kww windows of a [B,kH,kW,.] image that
 *                          share a window grid -- WindowAttention.forward (:381-412: both sides the 8x8 windows of the
 *                          rolled qkv image, shift = 4 with calculate_mask's regions :1607-1622, or 0) and the two passes of
 *                          AnchorStripeAttention.forward (:463-514: 4x4 anchor windows against 8x8 stripes, then back).
 *                          q / k / v / out point at the first channel of head 0, ld* = floats between pixels, head h owns
 *                          channels [h d, (h+1) d); windows <= 64 tokens, d <= 64.  The result lands at the query token's
 *                          own pixel: torch.roll / window_partition / window_reverse are address arithmetic. */
int srhip_avgpool2d(const float* x, float* out, int B, int H, int W, int C, int k, void* stream);
int srhip_cpb_bias(const float* table, const long long* index, float* biasT, int heads, int N1, int N2, int entries, void* stream);
int srhip_cosine_window_attention(const float* q, long ldq, int qH, int qW, int qwh, int qww, const float* k, long ldk,
                                  const float* v, long ldv, int kH, int kW, int kwh, int kww, const float* logit_scale,
                                  const float* biasT, float* out, long ldo, int B, int heads, int d, int shift, void* stream);

/* ---- window attention on the two-plane fp16 split MFMA (wattn2.hip) -------------- */
/* The same contract as srhip_window_attention_fwd (network_swinir.py:48-80,153-176,297-331) with the two
 * contractions as three fp16 products under power-of-two block exponents (q, k per token row, v per head-dim
 * column, p a fixed 2^14; f32 accumulation): f32-grade results at a fifth of the matrix-core time, no LDS.
 * biasF = the bias image in the kernel's accumulator order: img[head][I][J][lane][e] = table[rpi(query 16 I +
 * (lane & 15), key 16 J + 4 (lane >> 4) + e)][head], heads*4096 floats (srhip_bias_expand_f16x2, or the `c`
 * output of a srhip_prep_table job of kind 2; biasG -- may be NULL -- see the backward). */
int srhip_bias_expand_f16x2(const float* table, float* biasF, float* biasG, int heads, void* stream);
int srhip_window_attention_fwd_f16x2(const float* qkv, float* out, const float* biasF, int B, int H, int W, int C,
                                     int heads, int shift, void* stream);
/* Backward of the same core in ONE kernel (one wave per (window, head): query side, then key side; every operand is
 * read in the form the matrix core takes it, nothing but the bias-gradient tile goes through LDS).  biasG = the bias
 * image in the key side's accumulator order, img[head][J][I][lane][e] = table[rpi(query 16 I + 4 (lane >> 4) + e,
 * key 16 J + (lane & 15))][head] (second output of srhip_bias_expand_f16x2 / `b` of a prep job of kind 2).
 * dqkv [T][3C] is overwritten; dbiasT (may be NULL) receives the bias-gradient image in the order srhip_bias_grad
 * reads; workspace: srhip_window_attention_bwd_f16x2_ws floats (partial tiles, summed in fp64 in a fixed order;
 * may be NULL when dbiasT is). */
long srhip_window_attention_bwd_f16x2_ws(int B, int H, int W, int heads);
int srhip_window_attention_bwd_f16x2(const float* qkv, const float* dout, float* dqkv, const float* biasF,
                                     const float* biasG, float* dbiasT, float* workspace, int B, int H, int W, int C,
                                     int heads, int shift, void* stream);
/* With dbiasT NULL and a workspace the backward leaves its partial tiles there; this sums the partials of nblocks
 * attention blocks of one geometry (workspace + i * ws_stride -> dbiasT + i * img_stride floats) in ONE launch -- the
 * blocks of an RSTB layer share it. */
int srhip_window_attention_dbias_reduce_f16x2(const float* workspace, long ws_stride, int nblocks, float* dbiasT,
                                              long img_stride, int B, int H, int W, int heads, void* stream);

/* ---- window attention (network_swinir.py:48-80,140-179,297-331) -------------- */
/* table (225,heads) -> two bias images of heads*4096 floats each, stored in the order
 * the attention kernels read them as 32x32 MFMA accumulator tiles:
 * img[h][a][b][lane][q], row = (q&3) + 8*(q>>2) + 4*(lane>>5), col = lane&31;
 *   biasT (fwd, bwd query pass): bias[query = col+32b][key = row+32a]
 *   biasN (bwd key pass)       : bias[query = row+32a][key = col+32b]
 * The bias-gradient image (dbiasT of srhip_window_attention_bwd, input of
 * srhip_bias_grad) is biasT's tile set stored [h][a][b][q][lane] (one wave instruction
 * touches 256 contiguous bytes).  Opaque to callers: produce with this
 * function (or a kind-2 job of srhip_prep_table), hand to the attention calls. */
int srhip_bias_expand(const float* table, float* biasT, float* biasN, int heads, void* stream);
int srhip_bias_grad(const float* dbiasT, float* dtable, int heads, void* stream);
/* The same for up to 8 attention blocks in one launch: block i's image at dbiasT + i*image_stride floats,
 * its table gradient at dtables[i] (HOST array of device pointers). */
int srhip_bias_grad_batched(const float* dbiasT, long image_stride, float* const* dtables, int nblocks, int heads,
                            void* stream);
/* qkv [B*H*W][3C] in token order -> out [B*H*W][C]; 8x8 windows, shift 0 or 4;
 * roll, window partition/reverse and the shift mask are address math. */
int srhip_window_attention_fwd(const float* qkv, float* out, const float* biasT, int B, int H, int W,
                               int C, int heads, int shift, void* stream);
/* dbiasT (may be NULL) is OVERWRITTEN with the bias-gradient image: the query pass stores one
 * partial tile set per block into the workspace (no atomics), tail blocks of the key pass sum
 * them in fp64 -- deterministic.  workspace floats: srhip_window_attention_bwd_ws(). */
long srhip_window_attention_bwd_ws(int B, int H, int W, int heads);
int srhip_window_attention_bwd(const float* qkv, const float* dout, float* dqkv, const float* biasT,
                               const float* biasN, float* dbiasT, float* workspace, int B, int H, int W,
                               int C, int heads, int shift, void* stream);

/* ---- 1-channel edge convolutions -------------------------------------------- */
/* x [B][H][W] -> y NHWC [.][Co]; flip is a flag word: bit 0 = flipped taps (= data gradient of
 * a Cout=1 conv), bit 1 = ReLU on the output (VDSR's input layer, network_vdsr.py:57-60).
 * network_swinir.py:786,945; network_nlsn.py:325. */
int srhip_conv3x3_cin1_fwd(const float* x, const float* w, const float* bias, float* y, long ldy, int B,
                           int H, int W, int Co, int flip, void* stream);
long srhip_conv3x3_cin1_wgrad_ws(int Co);
int srhip_conv3x3_cin1_wgrad(const float* x, const float* dy, long lddy, float* dw, float* db,
                             float* workspace, int B, int H, int W, int Co, int flip, void* stream);
/* x NHWC [.][Ci] -> y [B][H][W]; network_nlsn.py:347-350. */
int srhip_conv3x3_cout1_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, int B,
                            int H, int W, int Ci, void* stream);

/* ---- pixel shuffle (index only; network_swinir.py:701, network_nlsn.py:108) -- */
/* in NHWC [B][h][w][Co*r*r] -> out NCHW [B][Co][h*r][w*r] or NHWC [B][h*r][w*r][Co];
 * inverse=1 runs the mapping backwards (gradient). */
int srhip_pixel_shuffle(const float* in, float* out, int B, int h, int w, int Co, int r,
                        int nhwc_out, int inverse, void* stream);
/* out NHWC [B][h*r][w*r][Co] = PixelShuffle(in) + fac * add (add laid out as out): the addend behind a transposed conv run as
 * conv3x3 + PixelShuffle (DBPN's projection units, network_dbpn.py:93-99,128-134).  Co*(r*r+1)*4 <= 48 KB, Co*r*r >= 256. */
int srhip_pixel_shuffle_add(const float* in, float* out, int B, int h, int w, int Co, int r, const float* add, float fac,
                            void* stream);

/* ---- losses (dlib/loss/main.py:45-99; MasterLoss dlib/loss/master.py:46-56) -- */
/* mode 0: lam*mean(|pred-target| * weight?) ; mode 1: lam*mean((pred-target)^2).
 * Writes (or accumulates) the value into loss_out[0] and d loss / d pred into
 * grad.  workspace: 2048 doubles. */
int srhip_loss_l1l2(const float* pred, const float* target, const float* weight, float* grad,
                    float* loss_out, double* workspace, long n, int mode, float lam, int grad_accum,
                    int loss_accum, void* stream);
/* -lam * mean_b(mean_hw(ssim_map)) with a zero-padded Gaussian window of odd
 * size ws (sigma 1.5), value and gradient (dlib/loss/ssim.py:38-61,
 * dlib/loss/main.py:154-186).  workspace floats: srhip_ssim_loss_ws(B,H,W). */
long srhip_ssim_loss_ws(int B, int H, int W);
int srhip_ssim_loss(const float* pred, const float* target, float* grad, float* loss_out,
                    float* workspace, int B, int H, int W, int ws, float lam, int grad_accum,
                    int loss_accum, void* stream);

/* ---- optional MasterLoss terms (off by default, utils_config.py:279-374) ---- */
/* Point-wise terms, value + d loss / d pred in one pass.  mode 0 L1 (optional per-pixel weight),
 * 1 L2, 2 Charbonnier lam*mean(sqrt((t-p)^2 + eps)) (dlib/loss/main.py:125-151), 3 L2Sum
 * lam*sum((p-t)^2) (dlib/loss/main.py:102-122).  workspace: 2048 doubles. */
int srhip_loss_pointwise(const float* pred, const float* target, const float* weight, float* grad, float* loss_out,
                         double* workspace, long n, int mode, float lam, float eps, int grad_accum, int loss_accum,
                         void* stream);
/* BoundedPrediction (dlib/loss/main.py:189-237) with the extended log barrier of dlib/losses/elb.py:92-122 at
 * barrier parameter t: lam * (mean elb(s*pred - (s*target + eps)) + mean elb(s*target - eps - s*pred)) / 2,
 * s = scale (color_max when restore_range, else 1).  Value + gradient.  workspace: 2048 doubles. */
int srhip_loss_bounded(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                       long n, float lam, float eps, float t, float scale, int grad_accum, int loss_accum,
                       void* stream);
/* WeightsSparsityLoss (dlib/loss/main.py:938-959) on a flat parameter range: loss_out (+)= lam * sum|w|,
 * grad[i] += lam * sign(w[i]) (grad may be null).  workspace: 2048 doubles. */
int srhip_l1_sparsity(const float* w, float* grad, float* loss_out, double* workspace, long n, float lam,
                      int loss_accum, void* stream);
/* LocalMoments (dlib/loss/main.py:240-325; PatchMoments dlib/loss/local_terms.py:14-66, the reference's fixed
 * 3x3 window, reflect padding, unbiased variance): lam * mean_{b,y,x}(KL(target patch || pred patch) * [target
 * patch variance == 0]) with both variances + 1.  workspace doubles: srhip_loss_stencil_ws(B,H,W). */
int srhip_loss_local_moments(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                             int B, int H, int W, float lam, int grad_accum, int loss_accum, void* stream);
/* HistogramMatch (dlib/loss/main.py:690-782) over SoftHistogram(bins, 0, 1, sigma) (dlib/loss/global_terms.py:17-72),
 * p = (h + 1) / sum per image; n = values per image.  norm 1 | 2: lam * mean_{b,k} nrm(p_pred - p_target);
 * 3 (KL): lam * nn.KLDivLoss(batchmean)(log p_pred, p_target); 4 (BHATTACHARYYA): lam * elb(-sum_k sqrt(p_pred p_target))
 * with the extended log barrier of dlib/losses/elb.py:92-122 at parameter elb_t (mean over images).
 * workspace floats: srhip_loss_hist_ws(B, bins). */
long srhip_loss_hist_ws(int B, int bins);
int srhip_loss_hist(const float* pred, const float* target, float* grad, float* loss_out, float* workspace, int B,
                    long n, int bins, float sigma, int norm, float lam, int grad_accum, int loss_accum, float elb_t,
                    void* stream);
/* KDEMatch (dlib/loss/main.py:785-898) over GaussianKDE(kde_bw, bins, max_color 1, 1 channel)
 * (dlib/loss/global_terms.py:75-152), p[k] = mean_px N(x; linspace(0,1,bins)[k], kde_bw) + 1e-4.  norm 1 | 2:
 * lam * mean_{b,k} nrm(p_pred - p_target) / bins; 4 (BHATTACHARYYA): lam * elb(-sum_k sqrt(p_pred p_target)) at
 * parameter elb_t.  workspace floats: srhip_loss_hist_ws(B, bins). */
int srhip_loss_kde(const float* pred, const float* target, float* grad, float* loss_out, float* workspace, int B,
                   long n, int bins, float kde_bw, int norm, float lam, int grad_accum, int loss_accum, float elb_t,
                   void* stream);
/* Local-variation terms on 1-channel images [B][H][W] (dlib/loss/main.py:328-674 with the operators of
 * dlib/loss/local_variations.py:18-141, replicate padding): op 0 image gradient (2 stencils), 1 Laplacian
 * (1), 2 local variation over a ksz x ksz window (ksz^2 - 1 stencils; ksz 3, 5 or 7).  norm 1 | 2 = the
 * reference's NORM1 / NORM2.  channel_norm 0: lam * mean_{b,k,y,x} nrm(op_k(pred) - op_k(target))
 * (ImageGradientLoss / LaplacianFilterLoss / LocalVariationLoss); 1: lam * mean_{b,y,x} nrm(|op(pred)|_2 -
 * |op(target)|_2) (the Norm* classes).  Value into loss_out[0], gradient into grad (deterministic gather, no
 * atomics).  workspace doubles: srhip_loss_stencil_ws(B,H,W). */
long srhip_loss_stencil_ws(int B, int H, int W);
int srhip_loss_stencil(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                       int B, int H, int W, int op, int ksz, int norm, int channel_norm, float lam, int grad_accum,
                       int loss_accum, void* stream);

/* ---- input pipeline, device-side tail -------------------------------------- */
/* One training patch: crop P x P at (y0, x0) out of a resident uint8 tile [H][W], augment_img
 * mode 0..7 (dlib/utils/utils_image.py:469-487), uint8 -> float32 = np.float32(v / 255.)
 * (utils_image.py:322-323), layout [B][1][P][P] (single2tensor3 + batch collate) -- the
 * per-sample steps of dlib/datasets/dataset_dpsr.py:866-894,914-915.  jobs is a HOST array;
 * bit-exact. */
typedef struct {
  const unsigned char* img;   /* device pointer */
  int H, W, y0, x0, mode;
} srhip_patch_job;
int srhip_patch_gather(const srhip_patch_job* jobs, int B, int P, float* out, void* stream);
/* ROI-weighted patch origins, the 'roi' sampler of the training crops (PatchSampler._roi,
 * dataset_dpsr.py:330-369): origin (r, c) of the (H-P) x (W-P) candidates has probability
 * proportional to exp(5*roi) + 1, roi = img[r + P/2][c + P/2] >= threshold.  One uniform in [0,1)
 * (fp64) per patch selects the origin by the inverse CDF in row-major order; origins[b] = {row, col}
 * feeds srhip_patch_gather (y0, x0).  jobs[b].{img,H,W} are used; workspace: srhip_roi_sample_ws(B,
 * max_rows) ints with max_rows >= H - P of every tile. */
/* Patch matrix of a 1-channel image [B][H][W] for a k x k, stride 1, pad k/2 convolution:
 * out[t][dy*k+dx] = x[b][y+dy-k/2][x+dx-k/2] (zero outside the image), columns k*k .. ldo-1 zero.
 * With srhip_gemm_nt_bx3 / srhip_gemm_tn_bx3 it is the forward / weight gradient of SRCNN's
 * nn.Conv2d(1, 1024, 5, 1, 2) (network_srcnn.py:33). */
int srhip_im2col_c1(const float* x, float* out, long ldo, int B, int H, int W, int ksize, void* stream);
long srhip_roi_sample_ws(int B, int max_rows);
int srhip_roi_sample(const srhip_patch_job* jobs, int B, int P, int threshold, const double* uniforms,
                     int* workspace, int max_rows, int* origins, void* stream);

/* ---- metrics (dlib/utils/utils_image.py:369-372,843-1007,618-653,1010-1198;
 *      dlib/utils/utils_trainer.py:961-1032) ----------------------------------- */
/* (x.clamp(0,1)*255).round().clamp(0,255), round-half-even (utils_image.py:369-372). */
int srhip_tensor2uint82float(const float* in, float* out, long n, void* stream);
/* One pass: tensor2uint82float on both images (skipped if inputs_are_u8),
 * border crop, then for "no ROI" (slot 0) and each ROI threshold t (roi = H>=t):
 * out[b][slot][4] = PSNR, PSNR_Y, MSE, NRMSE in fp64.
 * workspace doubles: srhip_metrics_ws(B, nth). */
long srhip_metrics_ws(int B, int nth);
int srhip_metrics_psnr_family(const float* E, const float* Hh, int B, int H, int W, int border,
                              const int* thresholds_dev, int nth, int inputs_are_u8, double* workspace,
                              double* out, void* stream);
/* SSIM metric (11x11 sigma-1.5 valid window, inputs u8-ised then /255):
 * out[b][slot] fp32 means (ROI cropped by 5 as the reference does).
 * workspace doubles: B*(nth+1)*2. */
int srhip_metrics_ssim(const float* E, const float* Hh, int B, int H, int W, int border,
                       const int* thresholds_dev, int nth, int inputs_are_u8, double* workspace,
                       float* out, void* stream);

/* Weight (and bias) gradients of up to 40 3x3 convolutions of ONE shape -- the body of an
 * EDSR-style stack (network_nlsn.py:72-93,325-345 backward) -- in one contraction launch plus one
 * reducer launch: item k: dW_k[Cout][Cin][3][3] = sum_px dY_k (x) shifted X_k, db_k = sum_px dY_k.
 * items is a HOST array; part / part_colsum hold n * part_floats_per_item and n * S * Cout floats
 * (srhip_conv3x3_wgrad_batched_plan; item k's partial sums start at part + k * S*9*Cout*Cin, the strip-form kernel's
 * per-block words of all items sit behind the n-th item's).  fp16x2 / bf16x3 split MFMA (f32-accurate). */
typedef struct { const float* dY; const float* X; float* dW; float* db; } srhip_conv_wgrad_item;
int srhip_conv3x3_wgrad_batched_plan(int n, int B, int H, int W, int Cout, int Cin, int* S,
                                     long* part_floats_per_item);
int srhip_conv3x3_wgrad_batched_bx3(const srhip_conv_wgrad_item* items, int n, long lddy, long ldx, int B, int H,
                                    int W, int Cout, int Cin, float* part, float* part_colsum, int S,
                                    void* stream);

/* ---- optimizers (dlib/utils/utils_instance.py:216-247) on flat buffers -------
 * g is multiplied by gscale first (1/world_size after a sum all-reduce).  If
 * skip_flag != NULL and *skip_flag != 0 the update is skipped on the device
 * (non-finite loss, model_plain.py:344-346) -- no host sync. */
int srhip_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float b1,
                    float b2, float eps, float wd, float gscale, const int* skip_flag, void* stream);
int srhip_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                   int nesterov, int first, float gscale, const int* skip_flag, void* stream);
/* The same updates with the step number kept ON THE DEVICE: srhip_optim_tick does
 * counter += (skip_flag == 0); the *_dc kernels (launched after it on the same stream) take
 * Adam's bias corrections 1 - beta^counter and SGD's "first step" (counter == 1) from it.  A step
 * skipped by the non-finite flag then leaves the optimizer state exactly as the reference does
 * when it skips backward + optimizer.step() for that batch (model_plain.py:344-346), with no
 * host sync.  lr_dev (may be NULL): the learning rate read from device memory instead of the
 * argument -- a launch captured in a hipGraph then replays with the current rate of the per-iteration
 * schedule (MyStepLR, utils_trainer.py:370). */
int srhip_optim_tick(const int* skip_flag, int* counter, void* stream);
int srhip_adam_step_dc(float* p, const float* g, float* m, float* v, long n, const int* counter, float lr,
                       float b1, float b2, float eps, float wd, float gscale, const int* skip_flag,
                       const float* lr_dev, void* stream);
int srhip_sgd_step_dc(float* p, const float* g, float* buf, long n, const int* counter, float lr,
                      float momentum, float wd, int nesterov, float gscale, const int* skip_flag,
                      const float* lr_dev, void* stream);
/* Gradient clipping by the global L2 norm over the flat gradient -- torch.nn.utils.clip_grad_norm_(parameters, max_norm,
 * norm_type=2) of the step (model_plain.py:350-361, G_optimizer_clipgrad > 0).  g holds the SUM over ranks after the
 * all-reduce; the norm is taken of g * gscale (gscale = 1 / world_size: what DDP's averaged gradients give).  norm_coef[0] =
 * that norm, norm_coef[1] = min(1, max_norm / (norm + 1e-6)) (NaN if the norm is: torch's clamp); g *= norm_coef[1].  Two-stage
 * fixed-order reduction in double: deterministic, capturable in a hipGraph, no host sync.  workspace:
 * srhip_grad_norm_clip_ws() bytes, 8-byte aligned. */
long srhip_grad_norm_clip_ws(void);
int srhip_grad_norm_clip(float* g, long n, float gscale, float max_norm, float* norm_coef, void* workspace,
                         long workspace_bytes, void* stream);
/* Exponential moving average of the weights, ModelBase.update_E (model_base.py:213-219; E_decay > 0, model_plain.py:393-394):
 * e = e * decay + p * (1 - decay) over the flat parameter buffer.  Skipped on the device when *skip_flag != 0 (the
 * reference returns from the step before update_E on a non-finite loss, model_plain.py:344-346). */
int srhip_ema_update(float* e, const float* p, long n, float decay, const int* skip_flag, void* stream);
/* ---- data-parallel gradient exchange (comm.hip): what DistributedDataParallel's reducer does for the reference
 * (dlib/models/model_base.py:135-142), for a caller that is not PyTorch.  One process per GPU; one communicator per process
 * (RCCL over xGMI; resolved with dlopen at the first call -- the copy PyTorch bundles if the process has one, else ROCm's
 * librccl.so.1; never loaded by a process that does not call these).
 *   srhip_allreduce_unique_id   one rank (by convention 0) fills 128 bytes; the caller carries them to every rank (file /
 *                               socket / MPI -- its own rendezvous)
 *   srhip_allreduce_init        every rank, on its CURRENT device (hipSetDevice first): *comm = an opaque handle
 *   srhip_allreduce_bucket_async  in-place SUM over the ranks of buf[0 .. n) (a contiguous range of the flat gradient, in
 *                               backward-completion order), enqueued on comm_stream BEHIND everything enqueued on
 *                               compute_stream so far (one event); returns at once.  The optimizer divides by the world
 *                               size (gscale of srhip_adam_step / srhip_sgd_step).
 *   srhip_allreduce_flag_async  the same with MAX over one int: the step's non-finite flag (srhip_nonfinite_flag), so that
 *                               every rank skips the same update
 *   srhip_allreduce_wait        compute_stream waits for every exchange enqueued on comm_stream so far (in front of the
 *                               optimizer launch); nothing blocks the host
 *   srhip_allreduce_destroy
 * All calls of one communicator come from one thread, in the same order on every rank (RCCL's rule). */
int srhip_allreduce_unique_id(void* id128);
int srhip_allreduce_init(const void* id128, int rank, int world_size, void** comm);
int srhip_allreduce_bucket_async(void* comm, float* buf, long n, void* compute_stream, void* comm_stream);
int srhip_allreduce_flag_async(void* comm, int* flag, void* compute_stream, void* comm_stream);
int srhip_allreduce_wait(void* comm, void* comm_stream, void* compute_stream);
int srhip_allreduce_destroy(void* comm);
/* flag[0] |= any(!isfinite(x)): one device flag instead of the reference's
 * per-tensor host syncs (dlib/utils/tools.py:28-63, model_plain.py:344). */
int srhip_nonfinite_flag(const float* x, long n, int* flag, void* stream);
int srhip_axpby(float* y, const float* x, long n, float a, float b, void* stream);
/* g[i] = a[i] > 0 ? g[i] : 0 -- backward of a ReLU whose OUTPUT a was kept (nn.ReLU, network_vdsr.py:28). */
int srhip_relu_mask(float* g, const float* a, long n, void* stream);
/* g[i] = a[i] > 0 ? g[i] : alpha * g[i] -- backward of a LeakyReLU(alpha > 0) whose OUTPUT a was kept
 * (nn.LeakyReLU(0.2), network_mslapsr.py:58,83,92). */
int srhip_leaky_relu_mask(float* g, const float* a, long n, float alpha, void* stream);
/* Nearest-neighbour x2 of an NHWC image, lo [B][h][w][C] -> hi [B][2h][2w][C] (adjoint = 0), and its adjoint: lo = sum of
 * the 2 x 2 copies of hi (adjoint = 1).  F.interpolate(scale_factor=2, mode='nearest') of SwinIR's 'nearest+conv'
 * upsampler, dlib/models/network_swinir.py:948-961.  C a multiple of 4. */
int srhip_nearest_up2_nhwc(float* lo, float* hi, int B, int h, int w, int C, int adjoint, void* stream);
/* x[i] = x[i] > 0 ? x[i] : alpha * x[i] in place (nn.LeakyReLU(0.2) behind the 1-channel edge conv, network_mslapsr.py:80-83). */
int srhip_leaky_relu(float* x, long n, float alpha, void* stream);
/* out[0] = sum(x) (fp64 accumulation); workspace: 2048 doubles. */
int srhip_sum(const float* x, long n, float* out, double* workspace, void* stream);

/* ---- BatchNorm2d over channels-last activations [T][C], T = B*H*W (nn.BatchNorm2d of MemNet's BN -> ReLU -> conv
 * units: dlib/models/network_memnet.py:27-34,59-64,100-104,112-116).  C = 1 or a multiple of 64.
 *   coef [4][C] = mean, rstd = 1/sqrt(var + eps), k = gamma * rstd, beta.  Training: srhip_bn_stats fills it from the
 *   batch (biased variance) and updates running_mean / running_var in place (momentum, unbiased variance; pass NULL,
 *   NULL to leave them alone).  Evaluation: the caller fills coef from the running statistics.
 *   workspace: srhip_bn_workspace_bytes(T, C) bytes, caller-owned. */
int srhip_bn_workspace_bytes(long T, int C, long* bytes);
int srhip_bn_stats(const float* X, long T, int C, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, float* coef, void* workspace, long workspace_bytes,
                   void* stream);
/* Y = (X - mean) * k + beta, then ReLU if relu != 0 (nn.ReLU(True) behind every BatchNorm of the net). */
int srhip_bn_apply(const float* X, const float* coef, float* Y, long T, int C, int relu, void* stream);
/* Backward of Y = relu?(BN(X)) in training mode.  dz = dY * (A > 0) when A (the ReLU's output) is given, dY otherwise;
 * dgamma = sum dz * xhat, dbeta = sum dz, ADDED to what is there if accumulate != 0 (a shared unit is applied several
 * times per forward, network_memnet.py:69-72; either may be NULL); dX = k * (dz - mean(dz) - xhat * mean(dz * xhat)) (+ R: the skip
 * connection of the residual unit, network_memnet.py:36-40).  dX NULL: parameter gradients only. */
int srhip_bn_bwd(const float* dY, const float* A, const float* X, const float* coef, long T, int C, float* dX,
                 const float* R, float* dgamma, float* dbeta, int accumulate, void* workspace, long workspace_bytes,
                 void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SRHIP_H */
