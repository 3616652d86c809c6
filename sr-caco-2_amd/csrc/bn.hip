// BatchNorm2d over channels-last activations [T][C] (T = B*H*W pixels): the normalisation of MemNet's
// BN -> ReLU -> conv units (reference dlib/models/network_memnet.py:27-34,59-64,100-104,112-116).  HBM-bound:
// training forward = one statistics pass + one apply pass (ReLU fused), backward = one reduction pass (ReLU mask
// fused) + one apply pass (skip connection fused).
//
// One thread mapping for all four kernels: a lane owns ONE channel column (its coefficients sit in registers), the
// four waves of a block walk the rows of the block's row range -- a wave instruction touches 256 contiguous bytes.
// C = 1 (the image itself) is walked as rows of 64 pixels, the 64 pseudo-columns are folded by the finaliser.
// Sums are carried in fp64 (sum x, sum x^2 without cancellation trouble); partials are combined in a fixed order:
// deterministic results.
//
//   coef [4][C] = batch (or running) mean, rstd = 1/sqrt(var + eps), k = gamma * rstd, beta
#include "common.h"
#include "../../include/srhip.h"

namespace {

constexpr int BN_MAXBLK = 2048;

struct BnPlan {
  int Wd;        // row width walked: C, or 64 for C = 1
  long rows;     // ceil(n / Wd)
  int ncg;       // column groups of 64
  int nblk;      // row ranges
  long rpb;      // rows per block
};

inline BnPlan bn_plan(long T, int C) {
  BnPlan p;
  p.Wd = C == 1 ? 64 : C;
  const long n = T * C;
  p.rows = (n + p.Wd - 1) / p.Wd;
  p.ncg = p.Wd / 64;
  const long maxblk = BN_MAXBLK / p.ncg > 0 ? BN_MAXBLK / p.ncg : 1;
  p.rpb = (p.rows + maxblk - 1) / maxblk;
  if (p.rpb < 16) p.rpb = 16;
  p.nblk = (int)((p.rows + p.rpb - 1) / p.rpb);
  return p;
}
inline long bn_ws_bytes(const BnPlan& p, int C) { return ((long)p.nblk * 2 * p.Wd) * 8 + 2L * C * 4; }

// BWD = false: part[blk][0][col] = sum x, [1] = sum x^2
// BWD = true : part[blk][0][col] = sum dz, [1] = sum dz * xhat, dz = dY (* (A > 0)), xhat = (x - mean) * rstd
template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_reduce(const float* __restrict__ X, const float* __restrict__ DY,
                                                   const float* __restrict__ A, const float* __restrict__ coef,
                                                   long n, int C, int Wd, long rows, long rpb,
                                                   double* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + lane;
  const int ch = C == 1 ? 0 : col;
  const long r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  float mean = 0.f, rstd = 1.f;
  if (BWD) { mean = coef[ch]; rstd = coef[C + ch]; }
  double s1 = 0.0, s2 = 0.0;
  for (long r = r0 + wave; r < r1; r += 16) {
    float x[4], d[4], a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long idx = (r + 4 * u) * Wd + col;
      const bool ok = r + 4 * u < r1 && idx < n;
      x[u] = ok ? X[idx] : (BWD ? mean : 0.f);
      if (BWD) {
        d[u] = ok ? DY[idx] : 0.f;
        a[u] = (ok && A) ? A[idx] : 1.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (BWD) {
        const float dz = a[u] > 0.f ? d[u] : 0.f;
        const float xh = (x[u] - mean) * rstd;
        s1 += (double)dz;
        s2 += (double)dz * (double)xh;
      } else {
        s1 += (double)x[u];
        s2 += (double)x[u] * (double)x[u];
      }
    }
  }
  __shared__ double sh[4][2][64];
  sh[wave][0][lane] = s1;
  sh[wave][1][lane] = s2;
  __syncthreads();
  if (wave < 2) {
    const double v = sh[0][wave][lane] + sh[1][wave][lane] + sh[2][wave][lane] + sh[3][wave][lane];
    part[((long)blockIdx.x * 2 + wave) * Wd + col] = v;
  }
}

// training statistics -> coef, running-statistics update (nn.BatchNorm2d: momentum update with the UNBIASED variance)
__global__ void __launch_bounds__(256) k_bn_finalize(const double* __restrict__ part, int nblk, int Wd, int C, double cnt,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                     float eps, float* __restrict__ coef) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= C) return;
  double s1 = 0.0, s2 = 0.0;
  const int c0 = C == 1 ? 0 : ch, c1 = C == 1 ? 64 : ch + 1;
  for (int b = 0; b < nblk; ++b)
    for (int c = c0; c < c1; ++c) {
      s1 += part[((long)b * 2) * Wd + c];
      s2 += part[((long)b * 2 + 1) * Wd + c];
    }
  const double mean = s1 / cnt;
  double var = s2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  coef[ch] = (float)mean;
  coef[C + ch] = rstd;
  coef[2 * C + ch] = gamma[ch] * rstd;
  coef[3 * C + ch] = beta[ch];
  if (rmean) {
    const float unb = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
    rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
    rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * unb;
  }
}

// backward sums -> dgamma (+)= sum dz*xhat, dbeta (+)= sum dz, kv[0][ch] = sum dz / cnt, kv[1][ch] = sum dz*xhat / cnt
__global__ void __launch_bounds__(256) k_bn_bwd_finalize(const double* __restrict__ part, int nblk, int Wd, int C, double cnt,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         int accumulate, float* __restrict__ kv) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= C) return;
  double s1 = 0.0, s2 = 0.0;
  const int c0 = C == 1 ? 0 : ch, c1 = C == 1 ? 64 : ch + 1;
  for (int b = 0; b < nblk; ++b)
    for (int c = c0; c < c1; ++c) {
      s1 += part[((long)b * 2) * Wd + c];
      s2 += part[((long)b * 2 + 1) * Wd + c];
    }
  if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)s1;
  if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)s2;
  kv[ch] = (float)(s1 / cnt);
  kv[C + ch] = (float)(s2 / cnt);
}

// BWD = false: Y = act((X - mean) * k + beta)
// BWD = true : Y = k * (dz - kv0 - xhat * kv1) (+ R),  dz = X2 (* (A > 0)), X = the normalised tensor
template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_apply(const float* __restrict__ X, const float* __restrict__ DY,
                                                  const float* __restrict__ A, const float* __restrict__ R,
                                                  const float* __restrict__ coef, const float* __restrict__ kv,
                                                  float* __restrict__ Y, long n, int C, int Wd, long rows, long rpb,
                                                  int relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + lane;
  const int ch = C == 1 ? 0 : col;
  const long r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  const float mean = coef[ch], rstd = coef[C + ch], k = coef[2 * C + ch], beta = coef[3 * C + ch];
  float k0 = 0.f, k1 = 0.f;
  if (BWD) { k0 = kv[ch]; k1 = kv[C + ch]; }
  for (long r = r0 + wave; r < r1; r += 16) {
    float x[4], d[4], a[4], rr[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long idx = (r + 4 * u) * Wd + col;
      ok[u] = r + 4 * u < r1 && idx < n;
      x[u] = ok[u] ? X[idx] : 0.f;
      if (BWD) {
        d[u] = ok[u] ? DY[idx] : 0.f;
        a[u] = (ok[u] && A) ? A[idx] : 1.f;
        rr[u] = (ok[u] && R) ? R[idx] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long idx = (r + 4 * u) * Wd + col;
      float y;
      if (BWD) {
        const float dz = a[u] > 0.f ? d[u] : 0.f;
        const float xh = (x[u] - mean) * rstd;
        y = k * (dz - k0 - xh * k1) + rr[u];
      } else {
        y = (x[u] - mean) * k + beta;
        if (relu) y = fmaxf(y, 0.f);
      }
      if (ok[u]) Y[idx] = y;
    }
  }
}

}  // namespace

extern "C" {

int srhip_bn_workspace_bytes(long T, int C, long* bytes) {
  SR_REQUIRE(T > 0 && (C == 1 || (C > 0 && C % 64 == 0)) && bytes, "bn_workspace_bytes: T %ld C %d (C = 1 or a multiple of 64)", T, C);
  *bytes = bn_ws_bytes(bn_plan(T, C), C);
  return 0;
}

int srhip_bn_stats(const float* X, long T, int C, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, float* coef, void* ws, long ws_bytes, void* stream) {
  SR_REQUIRE(X && gamma && beta && coef && ws, "bn_stats: null operand");
  SR_REQUIRE(T > 0 && (C == 1 || (C > 0 && C % 64 == 0)), "bn_stats: T %ld C %d (C = 1 or a multiple of 64)", T, C);
  SR_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running_mean / running_var go together");
  const BnPlan p = bn_plan(T, C);
  SR_REQUIRE(ws_bytes >= bn_ws_bytes(p, C), "bn_stats: workspace %ld < %ld bytes", ws_bytes, bn_ws_bytes(p, C));
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  hipLaunchKernelGGL((k_bn_reduce<false>), dim3(p.nblk, p.ncg), dim3(256), 0, st, X, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, T * C, C, p.Wd, p.rows, p.rpb, part);
  SR_LAUNCH_CHECK("bn_reduce");
  hipLaunchKernelGGL(k_bn_finalize, dim3(sr_cdiv(C, 256)), dim3(256), 0, st, part, p.nblk, p.Wd, C, (double)T, gamma, beta,
                     running_mean, running_var, momentum, eps, coef);
  SR_LAUNCH_CHECK("bn_finalize");
  return 0;
}

int srhip_bn_apply(const float* X, const float* coef, float* Y, long T, int C, int relu, void* stream) {
  SR_REQUIRE(X && coef && Y, "bn_apply: null operand");
  SR_REQUIRE(T > 0 && (C == 1 || (C > 0 && C % 64 == 0)), "bn_apply: T %ld C %d (C = 1 or a multiple of 64)", T, C);
  const BnPlan p = bn_plan(T, C);
  hipLaunchKernelGGL((k_bn_apply<false>), dim3(p.nblk, p.ncg), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, coef, (const float*)nullptr, Y, T * C, C, p.Wd, p.rows,
                     p.rpb, relu);
  SR_LAUNCH_CHECK("bn_apply");
  return 0;
}

int srhip_bn_bwd(const float* dY, const float* A, const float* X, const float* coef, long T, int C, float* dX,
                 const float* R, float* dgamma, float* dbeta, int accumulate, void* ws, long ws_bytes, void* stream) {
  SR_REQUIRE(dY && X && coef && ws, "bn_bwd: null operand");
  SR_REQUIRE(T > 0 && (C == 1 || (C > 0 && C % 64 == 0)), "bn_bwd: T %ld C %d (C = 1 or a multiple of 64)", T, C);
  const BnPlan p = bn_plan(T, C);
  SR_REQUIRE(ws_bytes >= bn_ws_bytes(p, C), "bn_bwd: workspace %ld < %ld bytes", ws_bytes, bn_ws_bytes(p, C));
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  float* kv = (float*)(part + (long)p.nblk * 2 * p.Wd);
  hipLaunchKernelGGL((k_bn_reduce<true>), dim3(p.nblk, p.ncg), dim3(256), 0, st, X, dY, A, coef, T * C, C, p.Wd, p.rows,
                     p.rpb, part);
  SR_LAUNCH_CHECK("bn_bwd_reduce");
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(sr_cdiv(C, 256)), dim3(256), 0, st, part, p.nblk, p.Wd, C, (double)T, dgamma,
                     dbeta, accumulate, kv);
  SR_LAUNCH_CHECK("bn_bwd_finalize");
  if (dX) {
    hipLaunchKernelGGL((k_bn_apply<true>), dim3(p.nblk, p.ncg), dim3(256), 0, st, X, dY, A, R, coef, (const float*)kv, dX,
                       T * C, C, p.Wd, p.rows, p.rpb, 0);
    SR_LAUNCH_CHECK("bn_bwd_apply");
  }
  return 0;
}

}  // extern "C"
