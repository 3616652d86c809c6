// Internal launch descriptors shared between the kernel files and the C-ABI
// layer (api.hip).  Not part of the public interface (include/srhip.h is).
#pragma once
#include <hip/hip_runtime.h>

struct NtArgs {
  // A operand.  GEMM: row-major [M][lda].  CONV: NHWC image [batch][H][Wd][lda].
  const float* A; long lda;
  // W operand.  GEMM: [N][ldw] (K contiguous).  CONV: [9][N][ldw], tap stride wtap.
  const float* W; long ldw; long wtap;
  float* C; long ldc;
  int M, N, K;
  int n_tile;                 // set by the dispatcher
  int xcd_order;              // conv: XCD-aware tile order (set by the dispatcher)
  const float* bias;          // [N] or null
  int a_mode;                 // 0 plain | 1 (x-mean)*rstd from ln_stats | 2 gelu(x)
  const float* ln_stats;      // [M][2] = mean, rstd
  int epi;                    // 0 +bias | 1 relu | 2 R + s*(acc+bias) | 3 s*acc*gelu'(R) | 4 acc*(R>0) | 6 leaky relu(alpha) | 7 acc*(R>0 ? 1 : alpha)
                              // | 8 relu(R + s*(acc+bias)) | 9 prelu(acc+bias; *slope) | 10 prelu(acc+bias; *slope) + alpha*R
                              // | 11 gelu(acc+bias) (exact erf)
  const float* slope;         // epi 9 / 10: the PReLU's one learnable slope, on the device
  const float* pro_coef;      // conv (k_nhcw2): BatchNorm (evaluation) + ReLU on the INPUT, coef [4][K] = mean, rstd, gamma*rstd, beta
  const float* R; long ldr;
  float* aux; long ldaux;     // epi 3: optional second output gelu(R)
  // epi 5 (LayerNorm backward, bx3 GEMM only): R = x, R2 = residual gradient, ep_stats = {mean, rstd}[M]
  const float* R2; long ldr2; const float* ep_stats;
  int wide_epi;               // bx3 GEMM: transpose the tile through LDS, 16-byte epilogue accesses
  float* stats_out;           // bx3 GEMM, one N block: {mean, rstd} of every output row (next LayerNorm)
  const float* rowscale; int rows_per_scale; float alpha;
  // conv geometry
  int batch, H, Wd, tiles_x, tiles_y;
  int dbg;                    // ablation bits (env SRHIP_NT_DBG; 0 in production)
  int stagger;                // odd blocks sleep this many x 8128 cycles at start
  // bf16x3 path (gemm_ntb.hip): W pre-split into planes [3][(9)][N][Kp] bf16
  const unsigned short* Wb; int Kp;
  int amp;                    // 1: one bf16 product of the leading planes (reduced-precision inference)
  // conv + PixelShuffle(2) in one kernel (bx3 conv only).  1: the store goes to the shuffled image
  // C [batch][2H][2Wd][N/4] (ldc = its pixel pitch), kernel column n = sp*(N/4) + c is channel c of
  // sub-pixel sp = 2*i + j (weight rows in that order: prep perm 3).  2: the A operand is read from the
  // shuffled image [batch][2H][2Wd][K/4] in the same order, k = sp*(K/4) + c (the data gradient of 1)
  int ps;
  int k_rot;                  // k_ntw: rotate the K walk per block (gemm_ntw.hip)
  int wfmt;                   // weight operand format: 0 three bf16 planes | 1 two fp16 planes + per-row 2^-s (prep kind 3; k_nth)
  // batch of independent GEMMs in one launch (exact-f32 GEMM only; blockIdx.z = problem z): operand bases move by
  // (z / zdiv) * z?[0] + (z % zdiv) * z?[1] floats -- e.g. (sample, head) slices of a [B*T][heads*dh] matrix.  zcount <= 1: off
  int zcount, zdiv;
  long zA[2], zW[2], zC[2];
};

// one EDSR ResBlock (64 channels) per launch, forward or data gradient (resblock.hip)
struct ResBlockArgs {
  const float* X; long ldx;            // stage-1 input, NHWC [batch][H][Wd][ldx]: x (forward) / the incoming gradient g (backward)
  const void* W1; const void* W2;      // fp16x2 conv operands (prep kind 4) of the first / second conv of the launch
  const float* b1; const float* b2;    // forward: the convs' biases; backward: unused
  const float* Mask; long ldmask;      // backward: the saved activation a (ReLU mask of stage 1)
  float* Mid; long ldmid;              // stage-1 result: a (forward) / da (backward)
  float* Out; long ldout;              // x + rs (conv(a) + b2)  /  g + conv(da)
  float rs;
  int batch, H, Wd, tiles_x, tiles_y, k_rot;
};
int sr_resblock64(ResBlockArgs& p, int bwd, hipStream_t st);

// fp16-storage conv of the evaluation path (conv_h16.hip)
struct ConvH16Args {
  const _Float16* X; long ldx;       // NHWC fp16 [B][H][Wd][ldx]
  const unsigned short* Wb;          // leading plane of the fp16x2 conv operand (prep kind 4), the inverse scales behind both planes
  long plane_bytes; int Kp;          // set by the dispatcher
  const float* bias;                 // [N] (torch order) or null
  _Float16* Y; long ldy;             // NHWC fp16 [B][H][Wd][ldy] (ps: [B][2H][2Wd][ldy], N / 4 channels)
  const _Float16* R; long ldr;       // residual operand of epilogues 2 / 8
  int epi;                           // 0 +bias | 1 relu | 2 R + alpha*(acc+bias) | 6 leaky relu(alpha) | 8 relu(R + alpha*(acc+bias))
  float alpha;
  const float* in_bn;                // [4][K] = mean, rstd, gamma*rstd, beta: evaluation-mode BatchNorm + ReLU on the input
  int center_only;                   // 1: only the centre tap is non-zero (a 1x1 conv held as a 3x3 weight)
  int B, H, Wd, K, N, ps;
  int tiles_x, tiles_y;              // set by the dispatcher
};
int sr_conv3x3_h16(ConvH16Args& p, hipStream_t st);
int sr_conv_cin1_h16(const float* x, const float* w, const float* bias, void* y, long ldy, int B, int H, int W, int Co, int act,
                     float alpha, hipStream_t st);
int sr_srcnn_h16(const void* a0, const float* img, int B, int H, int W, const void* W1h, const float* b1, const void* W2h,
                 const float* b2, const float* w3, const float* b3, float* y, long T, hipStream_t st);
int sr_conv_cout1_h16(const void* x, long ldx, const float* w, const float* bias, const float* add, const float* in_bn, float* y,
                      int B, int H, int W, int Ci, hipStream_t st);

struct TnArgs {
  // out[i][j] = sum_m pa(A)[m][i] * pb(B)[m'][j]   (m' = m, or the tap-shifted pixel)
  const float* A; long lda;   // [M][lda]   (dY)
  const float* B; long ldb;   // [M][ldb]   (X)   / NHWC image for conv
  int M, NI, NJ;              // reduce length, out rows (dY cols), out cols (X cols)
  int a_rowscale_rows; const float* a_rowscale;   // optional per-sample scale on A rows
  int b_mode;                 // 0 plain | 1 (x-mean)*rstd | 2 gelu(x)
  const float* ln_stats;
  float* part;                // [S][taps][NI][NJ] partial sums
  float* part_colsum;         // [S][NI] column sums of A (bias grads) or null
  int S;                      // number of M slices
  int conv;                   // 0 | 1: B rows are tap-shifted pixels of a [batch][H][Wd] image
  int batch, H, Wd;
  int i_tile, j_tile, rows_per_slice;   // set by the dispatcher
  int ps;                     // conv: A is the shuffled gradient image [batch][2H][2Wd][NI/4] of a conv + PixelShuffle(2)
                              // (lda = its pixel pitch); out rows are written in torch channel order c*4 + sp
  float* aux;                 // set by the dispatcher: per-block words of the strip form (behind ALL partial sums of the call)
};

// the MLP half of a Swin block as one kernel per direction, on the two-plane fp16 operands of the Linear GEMMs (mlp_f16.hip)
struct MlpF16Args {
  const float* X; long ldx;          // GEMM-1 activation rows [M][ldx]: forward x (LayerNorm prologue), backward dy
  const float* ln_stats;             // forward: {mean, rstd}[M] of the X rows
  const unsigned short* W1; int N1, K1, Kp1;   // GEMM-1 weight (prep kind 3): rows = hidden units, K = channels
  const unsigned short* W2; int N2, K2, Kp2;   // GEMM-2 weight (prep kind 3): rows = channels, K = hidden units
  const float* b1; const float* b2;  // forward biases (b1 beta-folded)
  float* H; long ldh;                // forward: pre-activation output (may be null) | backward: its input
  float* dH; float* GH;              // backward outputs [M][ldh]
  float* out; long ldo;
  const float* R; long ldr;          // forward: residual (x) | backward: x of the LayerNorm
  const float* R2; long ldr2;        // backward: residual gradient (dy)
  const float* ep_stats;             // backward: {mean, rstd}[M] of x
  const float* rowscale; int rows_per_scale;   // DropPath multipliers per sample (null = 1)
  float* stats_out;                  // forward: {mean, rstd} of the out rows (may be null)
  int M, C, hid;
  // backward, optional third product: out3 = s3 * (dx . W3^T), W3 = planes [C][C] (prep kind 3)
  const unsigned short* W3; float* out3; long ld3; const float* rowscale3;
  // backward, optional front product: the dy rows are computed here, dy = res0 + LayerNorm_backward(X0 . W0^T; x0, stats0)
  // (W0 = planes [C][K0], prep kind 3), written to out0 and used in place of X / R2
  const unsigned short* W0; int K0, Kp0; const float* X0; long ld0;
  const float* x0; long ldx0; const float* stats0; const float* res0; long ldres0; float* out0; long ldo0;
  int k_rot;                         // 1: every block starts its six-stage K walks at another stage (set by the dispatcher)
  long long* dbg;                    // experiment builds: phase timestamps
  int stagger;                       // experiment builds: start delay of odd blocks (10-ns units)
  int stagger_mode; int* cu_count;   // experiment builds: 1 = delay the block that arrives second on its CU (per-CU arrival counters)
};
int sr_mlp_f16(MlpF16Args& p, int bwd, hipStream_t st);
// the W-MSA half of a Swin block, forward, as one launch (wmsa_f16.hip); rows are token-major [B*H*W][..], dense
struct WmsaF16Args {
  const float* X;                    // block input rows [T][C] (also the residual)
  const float* ln_stats;             // {mean, rstd}[T] of the X rows
  const unsigned short* Wqkv; const float* bqkv;    // planes of Wqkv*gamma [3C][C] (prep kind 3), beta-folded bias
  const unsigned short* Wproj; const float* bproj;  // planes of Wproj [C][C], bias (may be null)
  const float* biasF;                // relative-position bias images [heads][4096] (srhip_bias_expand_f16x2)
  const float* rowscale;             // DropPath multipliers per sample [B] (null = 1)
  float* qkv; float* att; float* out; float* stats_out;   // [T][3C], [T][C], [T][C], [T][2] (may be null)
  int B, H, W, C, heads, shift, Kp;
  float scale;
  int k_rot;                         // 1: every block starts its K walks at another stage (set by the dispatcher)
  long long* dbg;                    // experiment builds: phase timestamps
  int stagger;                       // experiment builds: start delay of a block's second window (10-ns units)
};
int sr_wmsa_f16(WmsaF16Args& p, hipStream_t st);

int sr_matmul_mode();        // 0: f32-accurate bf16x3 | 1: single bf16 product (srhip_set_matmul_mode)
int sr_gemm_nt(NtArgs& p, hipStream_t st);
int sr_conv3x3_nt(NtArgs& p, hipStream_t st);
// one job of the per-step weight preparation (device table; mirrors srhip_prep_entry)
struct PrepEntry {
  const float* a; const float* b; const float* c; void* out; void* out2;
  int kind, blk0;
  int n0, n1, n2;
  int s0, s1, s2, off;
  int mode;
};
int sr_prep_blocks(const PrepEntry& e);
int sr_prep_table(const PrepEntry* tab_dev, int n, int total_blocks, hipStream_t st);
__host__ __device__ static inline int sr_kp(int K) { return (K + 31) / 32 * 32; }
int sr_split3(const float* W, long ldw, int rows, int K, unsigned short* out, hipStream_t st);
int sr_gemm_ntb(NtArgs& p, hipStream_t st);
int sr_conv3x3_ntb(NtArgs& p, hipStream_t st);
int sr_gemm_ntb_lnbwd(NtArgs& p, hipStream_t st);
int sr_gemm_ntp(NtArgs& p, hipStream_t st);
int sr_gemm_ntw(NtArgs& p, hipStream_t st);
int sr_conv3x3_ntcw(NtArgs& p, hipStream_t st);
int sr_conv3x3_ntcw2(NtArgs& p, int rows_per_wave, hipStream_t st);
int sr_conv3x3_nhcw(NtArgs& p, hipStream_t st);
int sr_conv3x3_nhcw2(NtArgs& p, int rows_per_wave, hipStream_t st);   // the same on two fp16 planes (weight format 1)  // conv, 128- / 64-pixel x 64-column tiles, W fragments from global memory   // conv, 64-pixel x 192-column tiles, W fragments from global memory   // 192-column tiles, W fragments from global memory (gemm_ntw.hip)
int sr_gemm_tn(TnArgs& p, hipStream_t st);
int sr_gemm_tnb(TnArgs& p, hipStream_t st);
int sr_gemm_tnb_grouped(TnArgs* probs, int n, hipStream_t st);
int sr_tn_plan(int M, int NI, int NJ, int conv, int* S, long* part_floats);
int sr_gemm_tn_grouped(TnArgs* probs, int n, hipStream_t st);
int sr_tn_group_plan(int M, int ntiles, int* S);
int sr_tn_plan_t(int M, int NI, int NJ, int conv, int target, int* S, long* part_floats);
int sr_tn_plan_bx3(int M, int NI, int NJ, int conv, int* S, long* part_floats);
int sr_tn_group_plan_t(int M, int ntiles, int dflt_target, int* S);
int sr_tn_tiles(int NI, int NJ);
int sr_conv_wgrad_batched_plan(int n, int M, int NI, int NJ, int* S, long* part_floats_per_item);
int sr_conv_wgrad_batched_tnb(const TnArgs& base, const float* const* A, const float* const* B, int n,
                              long part_stride, long colsum_stride, hipStream_t st);
