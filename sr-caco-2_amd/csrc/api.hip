// C-ABI layer of libsrhip: error plumbing and the entry points of the dense
// contractions (declared in include/srhip.h).
#include <stdarg.h>
#include "common.h"
#include "kernels.h"
#include "../../include/srhip.h"

static thread_local char g_err[512] = "";

int sr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {

const char* srhip_last_error(void) { return g_err; }
int srhip_abi_version(void) { return 13; }
int srhip_experiments_enabled(void) {
#ifdef SRHIP_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// per THREAD: a call reads it when it enqueues its kernels (the launch arguments carry the choice), so two host threads
// driving two streams cannot flip each other's mode, and within a thread calls are sequential
static thread_local int g_matmul_mode = 0;
int srhip_set_matmul_mode(int mode) {
  if (mode != 0 && mode != 1) return sr_fail(-22, "set_matmul_mode: mode %d (0 = f32-accurate bf16x3, 1 = single bf16 product)", mode);
  g_matmul_mode = mode;
  return 0;
}
int srhip_get_matmul_mode(void) { return g_matmul_mode; }

int srhip_gemm_nt(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                  long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                  const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha, float* aux,
                  long ldaux, void* stream) {
  SR_REQUIRE(a_mode >= 0 && a_mode <= 2, "gemm_nt: a_mode %d", a_mode);
  SR_REQUIRE(epi >= 0 && epi <= 4, "gemm_nt: epi %d", epi);
  SR_REQUIRE(a_mode != 1 || ln_stats, "gemm_nt: layernorm prologue without stats");
  SR_REQUIRE(epi < 3 || R, "gemm_nt: epilogue %d needs R", epi);
  SR_REQUIRE(!rowscale || rows_per_scale > 0, "gemm_nt: rows_per_scale must be > 0");
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.wtap = 0; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.a_mode = a_mode; p.ln_stats = ln_stats;
  p.epi = epi; p.R = R; p.ldr = ldr; p.rowscale = rowscale; p.rows_per_scale = rows_per_scale;
  p.alpha = alpha; p.aux = aux; p.ldaux = ldaux;
  SR_REQUIRE(!aux || epi == 3, "gemm_nt: aux output is produced by epilogue 3 only");
  return sr_gemm_nt(p, (hipStream_t)stream);
}

int srhip_gemm_nt_batched(const float* A, long lda, long a_z0, long a_z1, const float* W, long ldw, long w_z0, long w_z1, float* C,
                          long ldc, long c_z0, long c_z1, int M, int N, int K, int zcount, int zdiv, void* stream) {
  SR_REQUIRE(A && W && C && zcount > 0 && zdiv > 0 && zcount <= 65535, "gemm_nt_batched: 1 <= zcount <= 65535, zdiv > 0 (zcount=%d)", zcount);
  SR_REQUIRE(a_z0 % 4 == 0 && a_z1 % 4 == 0 && w_z0 % 4 == 0 && w_z1 % 4 == 0, "gemm_nt_batched: operand strides must be multiples of 4 floats");
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.rows_per_scale = 1;
  p.zcount = zcount; p.zdiv = zdiv;
  p.zA[0] = a_z0; p.zA[1] = a_z1; p.zW[0] = w_z0; p.zW[1] = w_z1; p.zC[0] = c_z0; p.zC[1] = c_z1;
  return sr_gemm_nt(p, (hipStream_t)stream);
}

int srhip_conv3x3_nhwc(const float* X, long ldx, const float* Wp, const float* bias, float* Y, long ldy,
                       int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                       const float* rowscale, float alpha, void* stream) {
  SR_REQUIRE(epi >= 0 && epi <= 7 && epi != 3 && epi != 5, "conv3x3: epi %d", epi);
  SR_REQUIRE((epi != 4 && epi != 7) || R, "conv3x3: mask epilogue %d needs R", epi);
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = X; p.lda = ldx; p.W = Wp; p.ldw = Cin; p.wtap = (long)Cout * Cin; p.C = Y; p.ldc = ldy;
  p.N = Cout; p.K = Cin; p.bias = bias; p.epi = epi; p.R = R; p.ldr = ldr; p.rowscale = rowscale;
  p.rows_per_scale = H * W; p.alpha = alpha; p.batch = B; p.H = H; p.Wd = W;
  return sr_conv3x3_nt(p, (hipStream_t)stream);
}

int srhip_bf16x3_kp(int K) { return sr_kp(K); }

int srhip_split_bf16x3(const float* W, long ldw, int rows, int K, void* out, void* stream) {
  return sr_split3(W, ldw, rows, K, (unsigned short*)out, (hipStream_t)stream);
}

int srhip_prep_blocks(const srhip_prep_entry* e) {
  static_assert(sizeof(srhip_prep_entry) == sizeof(PrepEntry), "prep table layout");
  SR_REQUIRE(e && e->kind >= 0 && e->kind <= 4, "prep_blocks: kind");
  return sr_prep_blocks(*(const PrepEntry*)e);
}

int srhip_prep_table(const srhip_prep_entry* table_dev, int n, int total_blocks, void* stream) {
  return sr_prep_table((const PrepEntry*)table_dev, n, total_blocks, (hipStream_t)stream);
}

static int gemm_nt_split(int wfmt, const float* A, long lda, const void* Wb, const float* bias, float* C,
                      long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                      const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha, float* aux,
                      long ldaux, float* stats_out, void* stream) {
  SR_REQUIRE(a_mode >= 0 && a_mode <= 2, "gemm_nt_bx3: a_mode %d", a_mode);
  SR_REQUIRE(epi >= 0 && epi <= 4, "gemm_nt_bx3: epi %d", epi);
  SR_REQUIRE(a_mode != 1 || ln_stats, "gemm_nt_bx3: layernorm prologue without stats");
  SR_REQUIRE(epi < 3 || R, "gemm_nt_bx3: epilogue %d needs R", epi);
  SR_REQUIRE(!rowscale || rows_per_scale > 0, "gemm_nt_bx3: rows_per_scale must be > 0");
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.Wb = (const unsigned short*)Wb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.a_mode = a_mode; p.ln_stats = ln_stats;
  p.epi = epi; p.R = R; p.ldr = ldr; p.rowscale = rowscale; p.rows_per_scale = rows_per_scale;
  p.alpha = alpha; p.aux = aux; p.ldaux = ldaux; p.stats_out = stats_out;
  SR_REQUIRE(!aux || epi == 3, "gemm_nt_bx3: aux output is produced by epilogue 3 only");
  p.wfmt = wfmt;
  SR_REQUIRE(!wfmt || N % 180 == 0 || (N > 128 && N % 128 != 0), "gemm_nt_f16x2: N = %d does not run on 192-column tiles", N);
  return sr_gemm_ntb(p, (hipStream_t)stream);
}
int srhip_gemm_nt_bx3(const float* A, long lda, const void* Wb, const float* bias, float* C,
                      long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                      const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha, float* aux,
                      long ldaux, float* stats_out, void* stream) {
  return gemm_nt_split(0, A, lda, Wb, bias, C, ldc, M, N, K, a_mode, ln_stats, epi, R, ldr, rowscale, rows_per_scale, alpha,
                       aux, ldaux, stats_out, stream);
}
int srhip_gemm_nt_f16x2(const float* A, long lda, const void* Wh, const float* bias, float* C,
                        long ldc, int M, int N, int K, int a_mode, const float* ln_stats, int epi,
                        const float* R, long ldr, const float* rowscale, int rows_per_scale, float alpha, float* aux,
                        long ldaux, float* stats_out, void* stream) {
  return gemm_nt_split(1, A, lda, Wh, bias, C, ldc, M, N, K, a_mode, ln_stats, epi, R, ldr, rowscale, rows_per_scale, alpha,
                       aux, ldaux, stats_out, stream);
}

static int gemm_nt_split_lnbwd(int wfmt, const float* A, long lda, const void* Wb, float* out, long ldo, int M, int N, int K,
                            const float* x, long ldx, const float* stats, const float* res, long ldres,
                            void* stream) {
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.Wb = (const unsigned short*)Wb; p.C = out; p.ldc = ldo;
  p.M = M; p.N = N; p.K = K; p.R = x; p.ldr = ldx; p.R2 = res; p.ldr2 = ldres; p.ep_stats = stats;
  p.alpha = 1.f;
  p.wfmt = wfmt;
  SR_REQUIRE(!wfmt || N % 180 == 0 || N > 128, "gemm_nt_f16x2_lnbwd: N = %d does not run on 192-column tiles", N);
  return sr_gemm_ntb_lnbwd(p, (hipStream_t)stream);
}
int srhip_gemm_nt_bx3_lnbwd(const float* A, long lda, const void* Wb, float* out, long ldo, int M, int N, int K,
                            const float* x, long ldx, const float* stats, const float* res, long ldres,
                            void* stream) {
  return gemm_nt_split_lnbwd(0, A, lda, Wb, out, ldo, M, N, K, x, ldx, stats, res, ldres, stream);
}
int srhip_gemm_nt_f16x2_lnbwd(const float* A, long lda, const void* Wh, float* out, long ldo, int M, int N, int K,
                              const float* x, long ldx, const float* stats, const float* res, long ldres,
                              void* stream) {
  return gemm_nt_split_lnbwd(1, A, lda, Wh, out, ldo, M, N, K, x, ldx, stats, res, ldres, stream);
}

int srhip_mlp_fwd_f16x2(const float* x, long ldx, const float* stats, const void* W1h, const float* b1,
                        const void* W2h, const float* b2, float* h, long ldh, float* out, long ldo, int M, int C,
                        int hidden, const float* rowscale, int rows_per_scale, float* stats_out, void* stream) {
  SR_REQUIRE(x && stats && W1h && b1 && W2h && out, "mlp_fwd_f16x2: null operand");
  SR_REQUIRE(!rowscale || rows_per_scale > 0, "mlp_fwd_f16x2: rows_per_scale must be > 0");
  MlpF16Args p;
  memset(&p, 0, sizeof(p));
  p.X = x; p.ldx = ldx; p.ln_stats = stats;
  p.W1 = (const unsigned short*)W1h; p.N1 = hidden; p.K1 = C; p.Kp1 = sr_kp(C);
  p.W2 = (const unsigned short*)W2h; p.N2 = C; p.K2 = hidden; p.Kp2 = sr_kp(hidden);
  p.b1 = b1; p.b2 = b2; p.H = h; p.ldh = h ? ldh : 4; p.out = out; p.ldo = ldo; p.R = x; p.ldr = ldx;
  p.rowscale = rowscale; p.rows_per_scale = rows_per_scale; p.stats_out = stats_out;
  p.M = M; p.C = C; p.hid = hidden;
  return sr_mlp_f16(p, 0, (hipStream_t)stream);
}

int srhip_wmsa_fwd_f16x2(const float* x, const float* stats, const void* Wqkvh, const float* bqkv, const void* Wprojh,
                         const float* bproj, const float* biasF, const float* rowscale, float* qkv, float* att,
                         float* out, float* stats_out, int B, int H, int W, int C, int heads, int shift,
                         void* stream) {
  SR_REQUIRE(x && stats && Wqkvh && bqkv && Wprojh && biasF && att && out, "wmsa_fwd_f16x2: null operand");
  SR_REQUIRE(out != x, "wmsa_fwd_f16x2: out must not alias x (windows read the residual while others write)");
  WmsaF16Args p;
  memset(&p, 0, sizeof(p));
  p.X = x; p.ln_stats = stats;
  p.Wqkv = (const unsigned short*)Wqkvh; p.bqkv = bqkv; p.Wproj = (const unsigned short*)Wprojh; p.bproj = bproj;
  p.biasF = biasF; p.rowscale = rowscale; p.qkv = qkv; p.att = att; p.out = out; p.stats_out = stats_out;
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift;
  return sr_wmsa_f16(p, (hipStream_t)stream);
}

int srhip_mlp_bwd_f16x2(const float* dy, long lddy, const void* W2Th, const void* W1Th, const float* h, long ldh,
                        float* dh, float* gh, const float* x, long ldx, const float* stats, float* dx, long lddx,
                        int M, int C, int hidden, const float* rowscale, int rows_per_scale, void* stream) {
  SR_REQUIRE(dy && W2Th && W1Th && h && dh && x && stats && dx, "mlp_bwd_f16x2: null operand");
  SR_REQUIRE(!rowscale || rows_per_scale > 0, "mlp_bwd_f16x2: rows_per_scale must be > 0");
  SR_REQUIRE(sr_matmul_mode() == 0, "mlp_bwd_f16x2: f32-accurate matmul mode only");
  MlpF16Args p;
  memset(&p, 0, sizeof(p));
  p.X = dy; p.ldx = lddy;
  p.W1 = (const unsigned short*)W2Th; p.N1 = hidden; p.K1 = C; p.Kp1 = sr_kp(C);
  p.W2 = (const unsigned short*)W1Th; p.N2 = C; p.K2 = hidden; p.Kp2 = sr_kp(hidden);
  p.H = (float*)h; p.ldh = ldh; p.dH = dh; p.GH = gh; p.out = dx; p.ldo = lddx;
  p.R = x; p.ldr = ldx; p.R2 = dy; p.ldr2 = lddy; p.ep_stats = stats;
  p.rowscale = rowscale; p.rows_per_scale = rows_per_scale;
  p.M = M; p.C = C; p.hid = hidden;
  return sr_mlp_f16(p, 1, (hipStream_t)stream);
}

int srhip_mlp_bwd_chain_f16x2(const float* dy, long lddy, const void* W2Th, const void* W1Th, const float* h, long ldh,
                              float* dh, float* gh, const float* x, long ldx, const float* stats, float* dx, long lddx,
                              int M, int C, int hidden, const float* rowscale, int rows_per_scale, const void* W3h,
                              float* out3, long ld3, const float* rowscale3, void* stream) {
  SR_REQUIRE(dy && W2Th && W1Th && h && dh && x && stats && dx && W3h && out3, "mlp_bwd_chain_f16x2: null operand");
  SR_REQUIRE((!rowscale && !rowscale3) || rows_per_scale > 0, "mlp_bwd_chain_f16x2: rows_per_scale must be > 0");
  SR_REQUIRE(ld3 % 4 == 0 && out3 != dx, "mlp_bwd_chain_f16x2: out3 pitch must be a multiple of 4 floats, out3 != dx");
  SR_REQUIRE(sr_matmul_mode() == 0, "mlp_bwd_chain_f16x2: f32-accurate matmul mode only");
  MlpF16Args p;
  memset(&p, 0, sizeof(p));
  p.X = dy; p.ldx = lddy;
  p.W1 = (const unsigned short*)W2Th; p.N1 = hidden; p.K1 = C; p.Kp1 = sr_kp(C);
  p.W2 = (const unsigned short*)W1Th; p.N2 = C; p.K2 = hidden; p.Kp2 = sr_kp(hidden);
  p.H = (float*)h; p.ldh = ldh; p.dH = dh; p.GH = gh; p.out = dx; p.ldo = lddx;
  p.R = x; p.ldr = ldx; p.R2 = dy; p.ldr2 = lddy; p.ep_stats = stats;
  p.rowscale = rowscale; p.rows_per_scale = rows_per_scale;
  p.M = M; p.C = C; p.hid = hidden;
  p.W3 = (const unsigned short*)W3h; p.out3 = out3; p.ld3 = ld3; p.rowscale3 = rowscale3;
  return sr_mlp_f16(p, 1, (hipStream_t)stream);
}

int srhip_mlp_bwd_front_chain_f16x2(const float* X0, long ld0, int K0, const void* W0h, const float* x0, long ldx0,
                                    const float* stats0, const float* res0, long ldres0, float* dy, long lddy,
                                    const void* W2Th, const void* W1Th, const float* h, long ldh, float* dh, float* gh,
                                    const float* x, long ldx, const float* stats, float* dx, long lddx, int M, int C,
                                    int hidden, const float* rowscale, int rows_per_scale, const void* W3h, float* out3,
                                    long ld3, const float* rowscale3, void* stream) {
  SR_REQUIRE(X0 && W0h && x0 && stats0 && res0 && dy && W2Th && W1Th && h && dh && x && stats && dx,
             "mlp_bwd_front_chain_f16x2: null operand");
  SR_REQUIRE(K0 > 0 && K0 % 4 == 0 && ld0 % 4 == 0 && ldx0 % 4 == 0 && ldres0 % 4 == 0,
             "mlp_bwd_front_chain_f16x2: K0 and the front pitches must be multiples of 4 floats");
  SR_REQUIRE(!W3h || (out3 && ld3 % 4 == 0 && out3 != dx), "mlp_bwd_front_chain_f16x2: chained product needs out3 (!= dx)");
  SR_REQUIRE((!rowscale && !rowscale3) || rows_per_scale > 0, "mlp_bwd_front_chain_f16x2: rows_per_scale must be > 0");
  SR_REQUIRE(dy != dx && dy != res0, "mlp_bwd_front_chain_f16x2: dy must not alias dx or res0");
  SR_REQUIRE(sr_matmul_mode() == 0, "mlp_bwd_front_chain_f16x2: f32-accurate matmul mode only");
  MlpF16Args p;
  memset(&p, 0, sizeof(p));
  p.X = dy; p.ldx = lddy;
  p.W1 = (const unsigned short*)W2Th; p.N1 = hidden; p.K1 = C; p.Kp1 = sr_kp(C);
  p.W2 = (const unsigned short*)W1Th; p.N2 = C; p.K2 = hidden; p.Kp2 = sr_kp(hidden);
  p.H = (float*)h; p.ldh = ldh; p.dH = dh; p.GH = gh; p.out = dx; p.ldo = lddx;
  p.R = x; p.ldr = ldx; p.R2 = dy; p.ldr2 = lddy; p.ep_stats = stats;
  p.rowscale = rowscale; p.rows_per_scale = rows_per_scale;
  p.M = M; p.C = C; p.hid = hidden;
  p.W3 = (const unsigned short*)W3h; p.out3 = out3; p.ld3 = ld3; p.rowscale3 = rowscale3;
  p.W0 = (const unsigned short*)W0h; p.K0 = K0; p.Kp0 = sr_kp(K0); p.X0 = X0; p.ld0 = ld0;
  p.x0 = x0; p.ldx0 = ldx0; p.stats0 = stats0; p.res0 = res0; p.ldres0 = ldres0; p.out0 = dy; p.ldo0 = lddy;
  return sr_mlp_f16(p, 1, (hipStream_t)stream);
}

static int conv3x3_split(int wfmt, const float* X, long ldx, const void* Wb, const float* bias, float* Y, long ldy,
                           int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                           const float* rowscale, float alpha, void* stream, const float* in_bn_coef = nullptr,
                           const float* slope = nullptr) {
  SR_REQUIRE(epi >= 0 && epi <= 11 && epi != 3 && epi != 5, "conv3x3_bx3: epi %d", epi);
  SR_REQUIRE((epi != 4 && epi != 7 && epi != 8 && epi != 10) || R, "conv3x3_bx3: epilogue %d needs R", epi);
  SR_REQUIRE((epi != 9 && epi != 10) || (slope && !rowscale), "conv3x3_bx3: epilogue %d takes the PReLU slope (and no row scale)", epi);
  SR_REQUIRE(epi != 11 || (wfmt == 1 && Cout <= 4096 && Cout % 64 == 0 && Cin <= 4096),
             "conv3x3: the GELU epilogue runs on the 64-column fp16x2 kernel (Cout = %d a multiple of 64)", Cout);
  SR_REQUIRE(!in_bn_coef || (wfmt == 1 && Cout <= 4096 && Cout % 64 == 0 && Cin % 4 == 0),
             "conv3x3: the BatchNorm-ReLU input prologue runs on the 64-column fp16x2 kernel (Cout = %d a multiple of 64, <= 4096)", Cout);
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.pro_coef = in_bn_coef; p.slope = slope;
  p.A = X; p.lda = ldx; p.Wb = (const unsigned short*)Wb; p.C = Y; p.ldc = ldy;
  p.N = Cout; p.K = Cin; p.bias = bias; p.epi = epi; p.R = R; p.ldr = ldr; p.rowscale = rowscale;
  p.rows_per_scale = H * W; p.alpha = alpha; p.batch = B; p.H = H; p.Wd = W;
  p.wfmt = wfmt;
  return sr_conv3x3_ntb(p, (hipStream_t)stream);
}
int srhip_conv3x3_nhwc_bx3(const float* X, long ldx, const void* Wb, const float* bias, float* Y, long ldy,
                           int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                           const float* rowscale, float alpha, void* stream) {
  return conv3x3_split(0, X, ldx, Wb, bias, Y, ldy, B, H, W, Cin, Cout, epi, R, ldr, rowscale, alpha, stream);
}
int srhip_conv3x3_nhwc_f16x2(const float* X, long ldx, const void* Wh, const float* bias, float* Y, long ldy,
                             int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                             const float* rowscale, float alpha, void* stream) {
  SR_REQUIRE((Cout <= 4096 || Cout % 180 == 0) && Cin <= 4096, "conv3x3_f16x2: Cout <= 4096 or a multiple of 180, Cin <= 4096 (Cout=%d Cin=%d)", Cout, Cin);
  return conv3x3_split(1, X, ldx, Wh, bias, Y, ldy, B, H, W, Cin, Cout, epi, R, ldr, rowscale, alpha, stream);
}

int srhip_conv3x3_nhwc_split_ex(int wfmt, const float* X, long ldx, const void* Wp, const float* bias, float* Y, long ldy,
                                int B, int H, int W, int Cin, int Cout, int epi, const float* R, long ldr,
                                const float* rowscale, float alpha, const float* in_bn_coef, const float* slope,
                                void* stream) {
  SR_REQUIRE(wfmt == 0 || wfmt == 1, "conv3x3_split_ex: weight format %d (0 three bf16 planes, 1 two fp16 planes)", wfmt);
  SR_REQUIRE(wfmt == 0 || ((Cout <= 4096 || Cout % 180 == 0) && Cin <= 4096), "conv3x3_f16x2: Cout <= 4096 or a multiple of 180, Cin <= 4096 (Cout=%d Cin=%d)", Cout, Cin);
  return conv3x3_split(wfmt, X, ldx, Wp, bias, Y, ldy, B, H, W, Cin, Cout, epi, R, ldr, rowscale, alpha, stream, in_bn_coef, slope);
}

int srhip_resblock64_fwd_f16x2(const float* x, long ldx, const void* W1h, const float* b1, const void* W2h, const float* b2,
                               float res_scale, float* a, long lda, float* out, long ldout, int B, int H, int W, void* stream) {
  SR_REQUIRE(x && W1h && b1 && W2h && b2 && a && out, "resblock64_fwd: null operand");
  SR_REQUIRE(ldx % 4 == 0 && lda % 4 == 0 && ldout % 4 == 0 && ldx >= 64 && lda >= 64 && ldout >= 64,
             "resblock64_fwd: pixel pitches must be multiples of 4 floats, >= 64");
  SR_REQUIRE(out != x && a != x, "resblock64_fwd: out / a must not alias x (neighbouring tiles read its halo)");
  SR_REQUIRE((long)B * H * W * ldx < (1L << 30), "resblock64_fwd: input beyond 4 GB");
  SR_REQUIRE(sr_matmul_mode() == 0, "resblock64_fwd: f32-accurate matmul mode only");
  ResBlockArgs p;
  memset(&p, 0, sizeof(p));
  p.X = x; p.ldx = ldx; p.W1 = W1h; p.W2 = W2h; p.b1 = b1; p.b2 = b2; p.Mid = a; p.ldmid = lda; p.Out = out; p.ldout = ldout;
  p.rs = res_scale; p.batch = B; p.H = H; p.Wd = W;
  return sr_resblock64(p, 0, (hipStream_t)stream);
}
int srhip_resblock64_bwd_f16x2(const float* g, long ldg, const void* W2Th, const void* W1Th, const float* a, long lda,
                               float res_scale, float* da, long ldda, float* dx, long lddx, int B, int H, int W, void* stream) {
  SR_REQUIRE(g && W2Th && W1Th && a && da && dx, "resblock64_bwd: null operand");
  SR_REQUIRE(ldg % 4 == 0 && lda % 4 == 0 && ldda % 4 == 0 && lddx % 4 == 0 && ldg >= 64 && lda >= 64 && ldda >= 64 && lddx >= 64,
             "resblock64_bwd: pixel pitches must be multiples of 4 floats, >= 64");
  SR_REQUIRE(dx != g && da != g && da != a, "resblock64_bwd: dx / da must not alias g, da must not alias a");
  SR_REQUIRE((long)B * H * W * ldg < (1L << 30), "resblock64_bwd: input beyond 4 GB");
  SR_REQUIRE(sr_matmul_mode() == 0, "resblock64_bwd: f32-accurate matmul mode only");
  ResBlockArgs p;
  memset(&p, 0, sizeof(p));
  p.X = g; p.ldx = ldg; p.W1 = W2Th; p.W2 = W1Th; p.Mask = a; p.ldmask = lda; p.Mid = da; p.ldmid = ldda; p.Out = dx; p.ldout = lddx;
  p.rs = res_scale; p.batch = B; p.H = H; p.Wd = W;
  return sr_resblock64(p, 1, (hipStream_t)stream);
}

int srhip_conv3x3_nhwc_h16(const void* X, long ldx, const void* Wh, const float* bias, void* Y, long ldy, int B, int H, int W,
                           int Cin, int Cout, int epi, const void* R, long ldr, float alpha, int ps2, const float* in_bn_coef,
                           int center_only, void* stream) {
  ConvH16Args p;
  memset(&p, 0, sizeof(p));
  p.in_bn = in_bn_coef; p.center_only = center_only;
  p.X = (const _Float16*)X; p.ldx = ldx; p.Wb = (const unsigned short*)Wh; p.bias = bias; p.Y = (_Float16*)Y; p.ldy = ldy;
  p.R = (const _Float16*)R; p.ldr = ldr; p.epi = epi; p.alpha = alpha; p.B = B; p.H = H; p.Wd = W; p.K = Cin; p.N = Cout;
  p.ps = ps2 ? 1 : 0;
  return sr_conv3x3_h16(p, (hipStream_t)stream);
}
int srhip_conv3x3_cin1_h16(const float* x, const float* w, const float* bias, void* y, long ldy, int B, int H, int W, int Co,
                           int act, float alpha, void* stream) {
  SR_REQUIRE(act >= 0 && act <= 2, "conv3x3_cin1_h16: act %d (0 none, 1 relu, 2 leaky relu)", act);
  return sr_conv_cin1_h16(x, w, bias, y, ldy, B, H, W, Co, act, alpha, (hipStream_t)stream);
}
int srhip_conv3x3_cout1_h16(const void* x, long ldx, const float* w, const float* bias, const float* add, const float* in_bn_coef,
                            float* y, int B, int H, int W, int Ci, void* stream) {
  return sr_conv_cout1_h16(x, ldx, w, bias, add, in_bn_coef, y, B, H, W, Ci, (hipStream_t)stream);
}

int srhip_srcnn_fwd_h16(const void* patches, const float* image, int B, int H, int W, const void* W1h, const float* b1,
                        const void* W2h, const float* b2, const float* w3, const float* b3, float* y, long T, void* stream) {
  return sr_srcnn_h16(patches, image, B, H, W, W1h, b1, W2h, b2, w3, b3, y, T, (hipStream_t)stream);
}

int srhip_tn_plan(int M, int NI, int NJ, int conv, int* S, long* part_floats) {
  return sr_tn_plan(M, NI, NJ, conv, S, part_floats);
}

static int gemm_tn_any(bool bx, const float* A, long lda, const float* B, long ldb, int M, int NI, int NJ,
                       const float* a_rowscale, int a_rowscale_rows, int b_mode, const float* ln_stats,
                       float* part, float* part_colsum, int S, void* stream) {
  SR_REQUIRE(b_mode >= 0 && b_mode <= 2, "gemm_tn: b_mode %d", b_mode);
  SR_REQUIRE(b_mode != 1 || ln_stats, "gemm_tn: layernorm prologue without stats");
  SR_REQUIRE(!a_rowscale || a_rowscale_rows > 0, "gemm_tn: a_rowscale_rows must be > 0");
  TnArgs p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.NI = NI; p.NJ = NJ;
  p.a_rowscale = a_rowscale; p.a_rowscale_rows = a_rowscale_rows; p.b_mode = b_mode;
  p.ln_stats = ln_stats; p.part = part; p.part_colsum = part_colsum; p.S = S; p.conv = 0;
  return bx ? sr_gemm_tnb(p, (hipStream_t)stream) : sr_gemm_tn(p, (hipStream_t)stream);
}
int srhip_gemm_tn(const float* A, long lda, const float* B, long ldb, int M, int NI, int NJ,
                  const float* a_rowscale, int a_rowscale_rows, int b_mode, const float* ln_stats,
                  float* part, float* part_colsum, int S, void* stream) {
  return gemm_tn_any(false, A, lda, B, ldb, M, NI, NJ, a_rowscale, a_rowscale_rows, b_mode, ln_stats, part,
                     part_colsum, S, stream);
}
int srhip_gemm_tn_bx3(const float* A, long lda, const float* B, long ldb, int M, int NI, int NJ,
                      const float* a_rowscale, int a_rowscale_rows, int b_mode, const float* ln_stats,
                      float* part, float* part_colsum, int S, void* stream) {
  return gemm_tn_any(true, A, lda, B, ldb, M, NI, NJ, a_rowscale, a_rowscale_rows, b_mode, ln_stats, part,
                     part_colsum, S, stream);
}

int srhip_tn_tiles(int NI, int NJ) { return sr_tn_tiles(NI, NJ); }
// bx3 kernels hold one 8-wave block per CU: plan for one round of 256 blocks
int srhip_tn_plan_bx3(int M, int NI, int NJ, int conv, int* S, long* part_floats) {
  return sr_tn_plan_bx3(M, NI, NJ, conv, S, part_floats);
}
int srhip_tn_group_plan_bx3(int M, int ntiles, int* S) { return sr_tn_group_plan_t(M, ntiles, 256, S); }
int srhip_tn_group_plan(int M, int ntiles, int* S) { return sr_tn_group_plan(M, ntiles, S); }

static int gemm_tn_grouped_any(bool bx, const srhip_tn_problem* probs, int nprob, int M, int S, void* stream) {
  SR_REQUIRE(nprob >= 1 && nprob <= (bx ? 24 : 4), "gemm_tn_grouped: 1..%d problems", bx ? 24 : 4);
  TnArgs a[24];
  memset(a, 0, sizeof(a));
  for (int k = 0; k < nprob; ++k) {
    const srhip_tn_problem& q = probs[k];
    SR_REQUIRE(q.b_mode >= 0 && q.b_mode <= 2, "gemm_tn_grouped: b_mode %d", q.b_mode);
    SR_REQUIRE(q.b_mode != 1 || q.ln_stats, "gemm_tn_grouped: layernorm prologue without stats");
    TnArgs& p = a[k];
    p.A = q.A; p.lda = q.lda; p.B = q.B; p.ldb = q.ldb; p.M = M; p.NI = q.NI; p.NJ = q.NJ;
    p.a_rowscale = q.a_rowscale; p.a_rowscale_rows = q.a_rowscale_rows; p.b_mode = q.b_mode;
    p.ln_stats = q.ln_stats; p.part = q.part; p.part_colsum = q.part_colsum; p.S = S;
  }
  return bx ? sr_gemm_tnb_grouped(a, nprob, (hipStream_t)stream) : sr_gemm_tn_grouped(a, nprob, (hipStream_t)stream);
}
int srhip_gemm_tn_grouped(const srhip_tn_problem* probs, int nprob, int M, int S, void* stream) {
  return gemm_tn_grouped_any(false, probs, nprob, M, S, stream);
}
int srhip_gemm_tn_grouped_bx3(const srhip_tn_problem* probs, int nprob, int M, int S, void* stream) {
  return gemm_tn_grouped_any(true, probs, nprob, M, S, stream);
}

// conv + PixelShuffle(2) as one kernel per direction (NtArgs.ps / TnArgs.ps)
static int conv3x3_ps2_split(int wfmt, const float* X, long ldx, const void* Wb, const float* bias, float* Yup, long ldy,
                          int B, int H, int W, int Cin, int Cout, int epi, float alpha, void* stream) {
  SR_REQUIRE(epi == 0 || epi == 1 || epi == 6, "conv3x3_ps2_bx3: epi %d (0 bias | 1 relu | 6 leaky relu)", epi);
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = X; p.lda = ldx; p.Wb = (const unsigned short*)Wb; p.C = Yup; p.ldc = ldy;
  p.N = Cout; p.K = Cin; p.bias = bias; p.epi = epi; p.rows_per_scale = H * W; p.alpha = alpha;
  p.batch = B; p.H = H; p.Wd = W; p.ps = 1; p.wfmt = wfmt;
  return sr_conv3x3_ntb(p, (hipStream_t)stream);
}
static int conv3x3_ps2_bwd_data_split(int wfmt, const float* dYup, long lddy, const void* Wbt, float* dX, long ldx, int B, int H,
                                   int W, int Cout, int Cin, int epi, const float* R, long ldr, float alpha,
                                   void* stream) {
  SR_REQUIRE(epi == 0 || ((epi == 4 || epi == 7) && R), "conv3x3_ps2_bwd_data_bx3: epi %d (0 | 4, 7 with R)", epi);
  NtArgs p;
  memset(&p, 0, sizeof(p));
  p.A = dYup; p.lda = lddy; p.Wb = (const unsigned short*)Wbt; p.C = dX; p.ldc = ldx;
  p.N = Cin; p.K = Cout; p.rows_per_scale = H * W; p.alpha = epi == 7 ? alpha : 1.f; p.batch = B; p.H = H; p.Wd = W; p.ps = 2;
  p.epi = epi; p.R = R; p.ldr = ldr; p.wfmt = wfmt;
  return sr_conv3x3_ntb(p, (hipStream_t)stream);
}
int srhip_conv3x3_ps2_bx3(const float* X, long ldx, const void* Wb, const float* bias, float* Yup, long ldy,
                          int B, int H, int W, int Cin, int Cout, int epi, float alpha, void* stream) {
  return conv3x3_ps2_split(0, X, ldx, Wb, bias, Yup, ldy, B, H, W, Cin, Cout, epi, alpha, stream);
}
int srhip_conv3x3_ps2_f16x2(const float* X, long ldx, const void* Wh, const float* bias, float* Yup, long ldy,
                            int B, int H, int W, int Cin, int Cout, int epi, float alpha, void* stream) {
  return conv3x3_ps2_split(1, X, ldx, Wh, bias, Yup, ldy, B, H, W, Cin, Cout, epi, alpha, stream);
}
int srhip_conv3x3_ps2_bwd_data_bx3(const float* dYup, long lddy, const void* Wbt, float* dX, long ldx, int B, int H,
                                   int W, int Cout, int Cin, int epi, const float* R, long ldr, float alpha,
                                   void* stream) {
  return conv3x3_ps2_bwd_data_split(0, dYup, lddy, Wbt, dX, ldx, B, H, W, Cout, Cin, epi, R, ldr, alpha, stream);
}
int srhip_conv3x3_ps2_bwd_data_f16x2(const float* dYup, long lddy, const void* Wht, float* dX, long ldx, int B, int H,
                                     int W, int Cout, int Cin, int epi, const float* R, long ldr, float alpha,
                                     void* stream) {
  return conv3x3_ps2_bwd_data_split(1, dYup, lddy, Wht, dX, ldx, B, H, W, Cout, Cin, epi, R, ldr, alpha, stream);
}
int srhip_conv3x3_ps2_wgrad_bx3(const float* dYup, long lddy, const float* X, long ldx, int B, int H, int W,
                                int Cout, int Cin, float* part, float* part_colsum, int S, void* stream) {
  TnArgs p;
  memset(&p, 0, sizeof(p));
  p.A = dYup; p.lda = lddy; p.B = X; p.ldb = ldx; p.M = B * H * W; p.NI = Cout; p.NJ = Cin;
  p.part = part; p.part_colsum = part_colsum; p.S = S; p.conv = 1; p.batch = B; p.H = H; p.Wd = W; p.ps = 1;
  return sr_gemm_tnb(p, (hipStream_t)stream);
}

static int conv_wgrad_any(bool bx, const float* dY, long lddy, const float* X, long ldx, int B, int H, int W,
                          int Cout, int Cin, float* part, float* part_colsum, int S, void* stream) {
  TnArgs p;
  memset(&p, 0, sizeof(p));
  p.A = dY; p.lda = lddy; p.B = X; p.ldb = ldx; p.M = B * H * W; p.NI = Cout; p.NJ = Cin;
  p.part = part; p.part_colsum = part_colsum; p.S = S; p.conv = 1; p.batch = B; p.H = H; p.Wd = W;
  return bx ? sr_gemm_tnb(p, (hipStream_t)stream) : sr_gemm_tn(p, (hipStream_t)stream);
}
int srhip_conv3x3_wgrad(const float* dY, long lddy, const float* X, long ldx, int B, int H, int W,
                        int Cout, int Cin, float* part, float* part_colsum, int S, void* stream) {
  return conv_wgrad_any(false, dY, lddy, X, ldx, B, H, W, Cout, Cin, part, part_colsum, S, stream);
}
int srhip_conv3x3_wgrad_bx3(const float* dY, long lddy, const float* X, long ldx, int B, int H, int W,
                            int Cout, int Cin, float* part, float* part_colsum, int S, void* stream) {
  return conv_wgrad_any(true, dY, lddy, X, ldx, B, H, W, Cout, Cin, part, part_colsum, S, stream);
}

}  // extern "C"

int sr_matmul_mode() { return g_matmul_mode; }
