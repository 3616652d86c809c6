// Token-branch pieces of ACT, evaluation forward (reference dlib/models/network_act.py:468-541): F.unfold / F.fold between the
// channels-last feature map and the token matrix (:475,487-492,499-503,512-514,525), LayerNorm over wide token rows (576,
// 1152: nn.LayerNorm :115-133), the row softmax of the attention (:176,215).  The dense products around them are the
// library's GEMMs.
#include "common.h"

namespace {

// tokens[b][ty * nTx + tx][c * k * k + ky * k + kx] = x[b][ty * s + ky - pad][tx * s + kx - pad][c]  (zero outside)
__global__ void __launch_bounds__(256) k_unfold(const float* __restrict__ x, long ldx, float* __restrict__ tok, long ldt, int B,
                                                int H, int W, int C, int k, int s, int pad, int nTy, int nTx) {
  const int kk = k * k;
  const long n = (long)B * nTy * nTx * C * kk;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int d = (int)(i % ((long)C * kk));
    const long t = i / ((long)C * kk);                      // b * nT + token
    const int c = d / kk, kidx = d - c * kk, ky = kidx / k, kx = kidx - ky * k;
    const int tx = (int)(t % nTx), ty = (int)((t / nTx) % nTy), b = (int)(t / ((long)nTx * nTy));
    const int y = ty * s + ky - pad, xx = tx * s + kx - pad;
    tok[t * ldt + d] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[(((long)b * H + y) * W + xx) * ldx + c] : 0.f;
  }
}
// F.fold (pad 0): out[b][y][x][c] = sum over the tokens covering (y, x); a gather: deterministic
__global__ void __launch_bounds__(256) k_fold(const float* __restrict__ tok, long ldt, float* __restrict__ out, long ldo, int B,
                                              int H, int W, int C, int k, int s, int nTy, int nTx) {
  const int kk = k * k;
  const long n = (long)B * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xx = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    float a = 0.f;
    const int ty1 = min(y / s, nTy - 1), tx1 = min(xx / s, nTx - 1);
    for (int ty = ty1; ty >= 0 && ty * s + k > y; --ty)
      for (int tx = tx1; tx >= 0 && tx * s + k > xx; --tx)
        a += tok[(((long)b * nTy + ty) * nTx + tx) * ldt + c * kk + (y - ty * s) * k + (xx - tx * s)];
    out[p * ldo + c] = a;
  }
}
// nn.LayerNorm over rows of any width (two passes over the row, eps inside the root, biased variance): one wave per row
__global__ void __launch_bounds__(256) k_layernorm_rows(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                        const float* __restrict__ g, const float* __restrict__ bta, long M,
                                                        int C, float eps) {
  const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* r = x + m * ldx;
  float s1 = 0.f;
  for (int c = lane; c < C; c += 64) s1 += r[c];
  const float mean = wave_sum(s1) / (float)C;
  float s2 = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = r[c] - mean; s2 += d * d; }
  const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
  float* o = y + m * ldy;
  for (int c = lane; c < C; c += 64) o[c] = (r[c] - mean) * rstd * g[c] + bta[c];
}
// backward of nn.LayerNorm over rows of any width (the token branch of ACT in training): with xhat = (x - mean) rstd and
// u = dy gamma, dx = rstd (u - mean(u) - xhat mean(u xhat)); a wave owns rows m, m + 4 nblk, ...; its sums of dy xhat / dy per
// column live in its own LDS slice (one lane per column: no atomics, a fixed order), the block's four slices are added into
// part[block][2][C] and k_ln_rows_bwd_reduce adds the blocks in order
__global__ void __launch_bounds__(256) k_layernorm_rows_bwd(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                            long ldx, const float* __restrict__ g, float* __restrict__ dx,
                                                            long lddx, float* __restrict__ part, long M, int C, float eps) {
  extern __shared__ float acc[];                   // [4 waves][2][C]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* const ag = acc + (long)wave * 2 * C;
  float* const ab = ag + C;
  for (int c = lane; c < C; c += 64) { ag[c] = 0.f; ab[c] = 0.f; }
  for (long m = blockIdx.x * 4L + wave; m < M; m += gridDim.x * 4L) {
    const float* r = x + m * ldx;
    const float* d = dy + m * lddy;
    float s1 = 0.f;
    for (int c = lane; c < C; c += 64) s1 += r[c];
    const float mean = wave_sum(s1) / (float)C;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float t = r[c] - mean; s2 += t * t; }
    const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
    float su = 0.f, sx = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (r[c] - mean) * rstd, u = d[c] * g[c];
      su += u;
      sx += u * xh;
      ag[c] += d[c] * xh;
      ab[c] += d[c];
    }
    const float mu = wave_sum(su) / (float)C, mx = wave_sum(sx) / (float)C;
    float* o = dx + m * lddx;
    for (int c = lane; c < C; c += 64) {
      const float xh = (r[c] - mean) * rstd;
      o[c] = rstd * (d[c] * g[c] - mu - xh * mx);
    }
  }
  __syncthreads();
  float* const pb = part + (long)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    pb[c] = (acc[c] + acc[2 * C + c]) + (acc[4 * C + c] + acc[6 * C + c]);
}
// 64 columns per block; the four waves each add every fourth block's partial, then wave 0 adds the four sums in order
__global__ void __launch_bounds__(256) k_ln_rows_bwd_reduce(const float* __restrict__ part, int nblk, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float acc[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float a = 0.f;
  if (c < 2 * C)
    for (int b = q; b < nblk; b += 4) a += part[(long)b * 2 * C + c];
  acc[q][threadIdx.x & 63] = a;
  __syncthreads();
  if (q == 0 && c < 2 * C) {
    a = (acc[0][threadIdx.x] + acc[1][threadIdx.x]) + (acc[2][threadIdx.x] + acc[3][threadIdx.x]);
    if (c < C) dgamma[c] = a;
    else dbeta[c - C] = a;
  }
}
// the same for rows of at most 256 values, held in registers (one read of the row), with an optional residual: y = res + LN(x)
__global__ void __launch_bounds__(256) k_layernorm_rows_reg(const float* __restrict__ x, long ldx, const float* __restrict__ res,
                                                            long ldr, float* __restrict__ y, long ldy, const float* __restrict__ g,
                                                            const float* __restrict__ bta, long M, int C, float eps) {
  const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* r = x + m * ldx;
  float v[4], s1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < C ? r[c] : 0.f;
    s1 += v[k];
  }
  const float mean = wave_sum(s1) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d = lane + 64 * k < C ? v[k] - mean : 0.f;
    s2 += d * d;
  }
  const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
  float* o = y + m * ldy;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      const float t = (v[k] - mean) * rstd * g[c] + bta[c];
      o[c] = res ? res[m * ldr + c] + t : t;
    }
  }
}
// x[r][0:n] <- softmax(scale * x[r][0:n]): one wave per row
__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ x, long ld, long R, int n, float scale) {
  const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float* p = x + r * ld;
  float mx = -3.0e38f;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, p[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(p[j] * scale - mx); p[j] = e; sum += e; }
  const float inv = 1.0f / wave_sum(sum);
  for (int j = lane; j < n; j += 64) p[j] *= inv;
}

// the same, also leaving lse[r] = log sum exp(scale * x[r]) (a training forward keeps it: NLSN's bucket score)
__global__ void __launch_bounds__(256) k_softmax_rows_lse(float* __restrict__ x, long ld, long R, int n, float scale,
                                                          float* __restrict__ lse) {
  const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float* p = x + r * ld;
  float mx = -3.0e38f;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, p[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(p[j] * scale - mx); p[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < n; j += 64) p[j] *= inv;
  if (lane == 0) lse[r] = mx + logf(sum);
}
// backward of P = softmax(s) (and of lse = log sum exp s, whose gradient is P itself), in place on dP:
//   ds[r][j] = P[r][j] (dP[r][j] - sum_k P[r][k] dP[r][k] + dlse[r])        (dlse may be NULL)
__global__ void __launch_bounds__(256) k_softmax_rows_bwd(const float* __restrict__ P, float* __restrict__ dP, long ld, long R,
                                                          int n, const float* __restrict__ dlse) {
  const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* p = P + r * ld;
  float* d = dP + r * ld;
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) dot += p[j] * d[j];
  dot = wave_sum(dot);
  const float add = (dlse ? dlse[r] : 0.f) - dot;
  for (int j = lane; j < n; j += 64) d[j] = p[j] * (d[j] + add);
}
// out[r] = sum_c a[r][c] b[r][c]
__global__ void __launch_bounds__(256) k_rowdot(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                float* __restrict__ out, long R, int n) {
  const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += a[r * lda + j] * b[r * ldb + j];
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}

inline int ew_blocks(long n) { const long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }

}  // namespace

extern "C" {

/* F.unfold(x, k, stride = s, padding = pad) on channels-last data: x [B][H][W] pixels of ldx floats (C channels used) ->
 * tok [B * nTy * nTx] rows of ldt floats, columns c * k * k + ky * k + kx (torch's channel-major order); nT. = (dim + 2 pad
 * - k) / s + 1.  With k = 5, s = 1, pad = 2 it is the im2col of ACT's 5 x 5 head convs (network_act.py:362-364). */
int srhip_unfold(const float* x, long ldx, float* tok, long ldt, int B, int H, int W, int C, int k, int s, int pad,
                 void* stream) {
  SR_REQUIRE(x && tok && B > 0 && C > 0 && k > 0 && s > 0 && pad >= 0 && H + 2 * pad >= k && W + 2 * pad >= k && ldx >= C,
             "unfold: bad arguments");
  const int nTy = (H + 2 * pad - k) / s + 1, nTx = (W + 2 * pad - k) / s + 1;
  SR_REQUIRE(ldt >= (long)C * k * k, "unfold: token pitch %ld < %d", ldt, C * k * k);
  const long n = (long)B * nTy * nTx * C * k * k;
  hipLaunchKernelGGL(k_unfold, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, ldx, tok, ldt, B, H, W, C, k, s, pad,
                     nTy, nTx);
  SR_LAUNCH_CHECK("unfold");
  return 0;
}

/* F.fold(tok, (H, W), k, stride = s): the overlap-add inverse of srhip_unfold (pad 0); pixels no token covers get 0. */
int srhip_fold(const float* tok, long ldt, float* out, long ldo, int B, int H, int W, int C, int k, int s, void* stream) {
  SR_REQUIRE(tok && out && B > 0 && C > 0 && k > 0 && s > 0 && H >= k && W >= k && ldo >= C && ldt >= (long)C * k * k,
             "fold: bad arguments");
  const int nTy = (H - k) / s + 1, nTx = (W - k) / s + 1;
  const long n = (long)B * H * W * C;
  hipLaunchKernelGGL(k_fold, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, tok, ldt, out, ldo, B, H, W, C, k, s, nTy, nTx);
  SR_LAUNCH_CHECK("fold");
  return 0;
}

/* y[m] = LayerNorm(x[m]) * gamma + beta over rows of C floats (any C; eps 1e-5 in nn.LayerNorm).  y may alias x. */
int srhip_layernorm_rows(const float* x, long ldx, float* y, long ldy, const float* gamma, const float* beta, long M, int C,
                         float eps, void* stream) {
  SR_REQUIRE(x && y && gamma && beta && M > 0 && C > 0 && ldx >= C && ldy >= C, "layernorm_rows: bad arguments");
  if (C <= 256)
    hipLaunchKernelGGL(k_layernorm_rows_reg, dim3(sr_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0, y, ldy, gamma,
                       beta, M, C, eps);
  else
    hipLaunchKernelGGL(k_layernorm_rows, dim3(sr_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gamma, beta, M, C,
                       eps);
  SR_LAUNCH_CHECK("layernorm_rows");
  return 0;
}

static int ln_rows_bwd_blocks(long M) { return (int)(sr_cdiv(M, 4) < 512 ? sr_cdiv(M, 4) : 512); }
/* floats of workspace srhip_layernorm_rows_bwd takes */
long srhip_layernorm_rows_bwd_ws(long M, int C) { return M > 0 && C > 0 ? (long)ln_rows_bwd_blocks(M) * 2 * C : 0; }
/* Backward of srhip_layernorm_rows: dx, dgamma[C], dbeta[C]; rows of at most 2048 values; deterministic. */
int srhip_layernorm_rows_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma, float* dx, long lddx,
                             float* dgamma, float* dbeta, float* ws, long M, int C, float eps, void* stream) {
  SR_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && ws && M > 0 && C > 0 && C <= 2048 && lddy >= C && ldx >= C && lddx >= C,
             "layernorm_rows_bwd: bad arguments (rows of at most 2048 values; C=%d)", C);
  const int nblk = ln_rows_bwd_blocks(M);
  hipLaunchKernelGGL(k_layernorm_rows_bwd, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), (hipStream_t)stream, dy, lddy, x,
                     ldx, gamma, dx, lddx, ws, M, C, eps);
  SR_LAUNCH_CHECK("layernorm_rows_bwd");
  hipLaunchKernelGGL(k_ln_rows_bwd_reduce, dim3(sr_cdiv(2 * C, 64)), dim3(256), 0, (hipStream_t)stream, ws, nblk, C, dgamma, dbeta);
  SR_LAUNCH_CHECK("layernorm_rows_bwd_reduce");
  return 0;
}

/* y = res + LayerNorm(x) over rows of at most 256 values: the post-norm residuals of GRL's blocks (network_grl.py:1061-1076);
 * y may be x or res */
int srhip_layernorm_rows_res(const float* x, long ldx, const float* res, long ldr, float* y, long ldy, const float* gamma,
                             const float* beta, long M, int C, float eps, void* stream) {
  SR_REQUIRE(x && res && y && gamma && beta && M > 0 && C > 0 && C <= 256 && ldx >= C && ldy >= C && ldr >= C,
             "layernorm_rows_res: rows of at most 256 values (C=%d)", C);
  hipLaunchKernelGGL(k_layernorm_rows_reg, dim3(sr_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, res, ldr, y, ldy, gamma,
                     beta, M, C, eps);
  SR_LAUNCH_CHECK("layernorm_rows_res");
  return 0;
}

/* x[r] <- softmax(scale * x[r]) over n columns, in place. */
int srhip_softmax_rows(float* x, long ld, long R, int n, float scale, void* stream) {
  SR_REQUIRE(x && R > 0 && n > 0 && ld >= n, "softmax_rows: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows, dim3(sr_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, R, n, scale);
  SR_LAUNCH_CHECK("softmax_rows");
  return 0;
}

int srhip_softmax_rows_lse(float* x, long ld, long R, int n, float scale, float* lse, void* stream) {
  SR_REQUIRE(x && lse && R > 0 && n > 0 && ld >= n, "softmax_rows_lse: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows_lse, dim3(sr_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, R, n, scale, lse);
  SR_LAUNCH_CHECK("softmax_rows_lse");
  return 0;
}

int srhip_softmax_rows_bwd(const float* P, float* dP, long ld, long R, int n, const float* dlse, void* stream) {
  SR_REQUIRE(P && dP && R > 0 && n > 0 && ld >= n, "softmax_rows_bwd: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows_bwd, dim3(sr_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, P, dP, ld, R, n, dlse);
  SR_LAUNCH_CHECK("softmax_rows_bwd");
  return 0;
}

int srhip_rowdot(const float* a, long lda, const float* b, long ldb, float* out, long R, int n, void* stream) {
  SR_REQUIRE(a && b && out && R > 0 && n > 0 && lda >= n && ldb >= n, "rowdot: bad arguments");
  hipLaunchKernelGGL(k_rowdot, dim3(sr_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, R, n);
  SR_LAUNCH_CHECK("rowdot");
  return 0;
}

}  // extern "C"
