// HBM-bound helpers around the contractions: slice reduction of weight-gradient
// partials, LayerNorm folding of Linear weights, weight re-layouts, LayerNorm
// kernels, pixel-shuffle, losses, optimizers.  One wave = 64 lanes throughout.
#include "common.h"
#include "kernels.h"
#include "../../include/srhip.h"

namespace {

// ----------------------------------------------------------------------------
// weight-gradient finalisation
// ----------------------------------------------------------------------------
// out[i] = sum_s part[s][i]           (Linear: out = dW [NI][NJ])
// ... and db[j] = sum_s colsum[s][j] in the same launch: the blocks behind the
// first `main_blocks` handle the nb bias-gradient entries.
__global__ void k_reduce_slices(const float* __restrict__ part, float* __restrict__ out,
                                long n, int S, float beta, const float* __restrict__ colsum,
                                float* __restrict__ db, int nb, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {
    const int j = (blockIdx.x - main_blocks) * blockDim.x + threadIdx.x;
    if (j < nb) {
      float d = 0.f;
      for (int s = 0; s < S; ++s) d += colsum[(long)s * nb + j];
      db[j] = d;
    }
    return;
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)main_blocks * blockDim.x) {
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(s + u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < S; ++s) a += part[(long)s * n + i];
    out[i] = beta != 0.f ? out[i] * beta + a : a;
  }
}
// conv: part [S][9][Co][Ci] -> dW torch layout [Co][Ci][3][3]
// (the bias gradient, sum_s colsum[s][co], rides in the last block)
// dw[cc][tap] = sum over the S slices of part[s][tap][cc] (cc = co * Ci + ci).  Threads walk the partial sums in THEIR
// order -- consecutive lanes read consecutive floats of a slice, four slices in flight; the transposition to the torch
// weight layout happens on the write, 1 / S of the traffic.  (Walking dw's order instead made neighbouring lanes read nine
// different tap planes: 13 us for the 33 MB of a 180 x 180 conv at 28 slices.)
__device__ __forceinline__ void reduce_conv_slices(const float* __restrict__ part, float* __restrict__ dw, int Co, int Ci,
                                                   int S, int main_blocks) {
  const long cc_n = (long)Co * Ci, n = cc_n * 9;
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += (long)main_blocks * blockDim.x) {
    const int tap = (int)(j / cc_n);
    const long cc = j - tap * cc_n;
    const float* q = part + j;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {
      a0 += q[(long)(s + 0) * n]; a1 += q[(long)(s + 1) * n];
      a2 += q[(long)(s + 2) * n]; a3 += q[(long)(s + 3) * n];
    }
    for (; s < S; ++s) a0 += q[(long)s * n];
    dw[cc * 9 + tap] = (a0 + a1) + (a2 + a3);
  }
}
__global__ void k_reduce_conv_w(const float* __restrict__ part, float* __restrict__ dw,
                                int Co, int Ci, int S, const float* __restrict__ colsum,
                                float* __restrict__ db, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {
    for (int n = threadIdx.x; n < Co; n += blockDim.x) {
      float d = 0.f;
      for (int s = 0; s < S; ++s) d += colsum[(long)s * Co + n];
      db[n] = d;
    }
    return;
  }
  reduce_conv_slices(part, dw, Co, Ci, S, main_blocks);
}
// the same for n problems of one shape (blockIdx.y = problem; partial buffers part_stride /
// colsum_stride floats apart)
struct ConvReduceBatch {
  float* dw[40];
  float* db[40];
};
__global__ void k_reduce_conv_w_batched(const float* __restrict__ part, long part_stride,
                                        const float* __restrict__ colsum, long colsum_stride, int Co, int Ci,
                                        int S, ConvReduceBatch out, int main_blocks) {
  const int k = blockIdx.y;
  part += (long)k * part_stride;
  float* const dw = out.dw[k];
  float* const db = out.db[k];
  if ((int)blockIdx.x >= main_blocks) {
    if (!colsum || !db) return;
    colsum += (long)k * colsum_stride;
    for (int n = threadIdx.x; n < Co; n += blockDim.x) {
      float d = 0.f;
      for (int s = 0; s < S; ++s) d += colsum[(long)s * Co + n];
      db[n] = d;
    }
    return;
  }
  reduce_conv_slices(part, dw, Co, Ci, S, main_blocks);
}
// Linear fed by a folded LayerNorm (W_f = W*gamma, b_f = b + W.beta):
//   G = sum_s part, dbv = sum_s colsum
//   dW[n][k] = gamma[k]*G[n][k] + beta[k]*dbv[n];  db = dbv
//   dgamma[k] = sum_n W[n][k]*G[n][k];  dbeta[k] = sum_n W[n][k]*dbv[n]
// Block = 64 columns x 4 rows (one element per thread, slices unrolled by 8 so
// the loads overlap); the dgamma / dbeta contributions of a row block go to the workspace
// lnws[row block][2][K] (plain stores) and k_ln_affine_finish sums them in row-block order:
// no atomics, no memset, the same bits every run.
__global__ void __launch_bounds__(256) k_fin_ln_linear(
    const float* __restrict__ part, const float* __restrict__ colsum, int S,
    const float* __restrict__ W, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dW, float* __restrict__ db,
    float* __restrict__ lnws, int N, int K) {
  __shared__ float sd[4], sg[4][64], sb[4][64];
  const int c = threadIdx.x & 63, rr = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const int n0 = blockIdx.y * 4;
  if (c == 0) {   // one lane per row sums that row's bias-gradient slices
    const int n = n0 + rr;
    float d = 0.f;
    if (n < N)
      for (int s = 0; s < S; ++s) d += colsum[(long)s * N + n];
    sd[rr] = d;
    if (n < N && blockIdx.x == 0) db[n] = d;
  }
  __syncthreads();
  float ag = 0.f, ab = 0.f;
  const int n = n0 + rr;
  if (k < K && n < N) {
    const long sl = (long)N * K;
    const float* pp = part + (long)n * K + k;
    float G = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[(s + u) * sl];
#pragma unroll
      for (int u = 0; u < 8; ++u) G += v[u];
    }
    for (; s < S; ++s) G += pp[s * sl];
    const float d = sd[rr];
    const float w = W[(long)n * K + k];
    dW[(long)n * K + k] = gamma[k] * G + beta[k] * d;
    ag = w * G;
    ab = w * d;
  }
  sg[rr][c] = ag; sb[rr][c] = ab;
  __syncthreads();
  if (rr == 0 && k < K) {
    float* w2 = lnws + (long)blockIdx.y * 2 * K;
    w2[k] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
    w2[K + k] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
  }
}
// dgamma[k] = sum over row blocks of ws[rb][0][k], dbeta[k] = ... ws[rb][1][k], in row-block order (four
// interleaved chains per column, combined in a fixed order)
__device__ __forceinline__ void ln_affine_finish_one(const float* __restrict__ ws, int nrb, int K,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int idx) {
  if (idx >= 2 * K) return;
  const int which = idx >= K, k = idx - which * K;
  const float* q = ws + (long)which * K + k;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int rb = 0;
  for (; rb + 4 <= nrb; rb += 4) {
    a0 += q[(long)(rb + 0) * 2 * K]; a1 += q[(long)(rb + 1) * 2 * K];
    a2 += q[(long)(rb + 2) * 2 * K]; a3 += q[(long)(rb + 3) * 2 * K];
  }
  for (; rb < nrb; ++rb) a0 += q[(long)rb * 2 * K];
  (which ? dbeta : dgamma)[k] = (a0 + a1) + (a2 + a3);
}
__global__ void __launch_bounds__(256) k_ln_affine_finish(const float* __restrict__ ws, int nrb, int K,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  ln_affine_finish_one(ws, nrb, K, dgamma, dbeta, blockIdx.x * 256 + threadIdx.x);
}
// ... for many row blocks (k_ln_bwd: up to 2048): 16 row lanes per column, four chains each, combined in a fixed order
// (two 256-thread blocks walking 2048 rows serially took 144 us at T = 32768)
__global__ void __launch_bounds__(1024) k_ln_affine_finish_wide(const float* __restrict__ ws, int nrb, int K,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float sm[16][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int which = idx >= K, k = idx - which * K;
  if (idx < 2 * K) {
    const float* q = ws + (long)which * K + k;
    int rb = rl;
    for (; rb + 48 < nrb; rb += 64) {
      a0 += q[(long)(rb + 0) * 2 * K]; a1 += q[(long)(rb + 16) * 2 * K];
      a2 += q[(long)(rb + 32) * 2 * K]; a3 += q[(long)(rb + 48) * 2 * K];
    }
    for (; rb < nrb; rb += 16) a0 += q[(long)rb * 2 * K];
  }
  sm[rl][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && idx < 2 * K) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][c];
    (which ? dbeta : dgamma)[k] = t;
  }
}
// The slice reducers of up to 24 Linear weight gradients (one Swin block: 4; one RSTB layer: 4 x depth) in ONE
// launch: k_fin_ln_linear's geometry per problem; gamma == null means a plain Linear
// (dW = G, db = dbv, no LayerNorm gradients).
struct ReduceGroup {
  struct P {
    const float* part; const float* colsum; const float* W; const float* gamma; const float* beta;
    float* dW; float* db; float* dgamma; float* dbeta; float* lnws;
    int N, K, blk0, kblocks, fblk0;
  } p[24];              // = TNB_GROUP_MAX of gemm_tnb.hip
  int n, S;
};
// a block: RG_ROWS rows x 64 columns of one problem (four rows per pass); the 24-way problem select from the kernel
// arguments is paid once per 1024 elements (at 4 rows per block it cost more than the sums)
constexpr int RG_ROWS = 16;
__global__ void __launch_bounds__(256) k_reduce_group(ReduceGroup g) {
  __shared__ float sd[RG_ROWS], sg[4][64], sb[4][64];
  // select with constant indices (a runtime-indexed struct array would go to scratch)
  ReduceGroup::P P = g.p[0];
#pragma unroll
  for (int i = 1; i < 24; ++i)
    if (i < g.n && (int)blockIdx.x >= g.p[i].blk0) P = g.p[i];
  const int S = g.S, N = P.N, K = P.K;
  const int lb = blockIdx.x - P.blk0;
  const int bx = lb % P.kblocks, by = lb / P.kblocks;
  const int c = threadIdx.x & 63, rr = threadIdx.x >> 6;
  const int k = bx * 64 + c;
  if (threadIdx.x < RG_ROWS) {   // one lane per row sums that row's bias-gradient slices
    const int n = by * RG_ROWS + threadIdx.x;
    float d = 0.f;
    if (n < N)
      for (int s = 0; s < S; ++s) d += P.colsum[(long)s * N + n];
    sd[threadIdx.x] = d;
    if (n < N && bx == 0) P.db[n] = d;
  }
  __syncthreads();
  float ag = 0.f, ab = 0.f;
  const long sl = (long)N * K;
  const float gk = (P.gamma && k < K) ? P.gamma[k] : 0.f, bk = (P.gamma && k < K) ? P.beta[k] : 0.f;
#pragma unroll
  for (int pass = 0; pass < RG_ROWS / 4; ++pass) {
    const int n = by * RG_ROWS + pass * 4 + rr;
    if (k < K && n < N) {
      const float* pp = P.part + (long)n * K + k;
      float G = 0.f;
      int s = 0;
      for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(s + u) * sl];
#pragma unroll
        for (int u = 0; u < 8; ++u) G += v[u];
      }
      for (; s + 4 <= S; s += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = pp[(s + u) * sl];
#pragma unroll
        for (int u = 0; u < 4; ++u) G += v[u];
      }
      for (; s < S; ++s) G += pp[s * sl];
      if (P.gamma) {
        const float d = sd[pass * 4 + rr];
        const float w = P.W[(long)n * K + k];
        P.dW[(long)n * K + k] = gk * G + bk * d;
        ag += w * G;
        ab += w * d;
      } else {
        P.dW[(long)n * K + k] = G;
      }
    }
  }
  if (P.gamma) {          // block-uniform
    sg[rr][c] = ag; sb[rr][c] = ab;
    __syncthreads();
    if (rr == 0 && k < K) {     // this row block's share of dgamma / dbeta: plain stores, summed by k_ln_affine_finish_group
      float* w2 = P.lnws + (long)by * 2 * K;
      w2[k] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
      w2[K + k] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
    }
  }
}
// second stage for the LayerNorm problems of a group: block -> problem by fblk0 (blocks of 256 threads over 2 K entries)
__global__ void __launch_bounds__(256) k_ln_affine_finish_group(ReduceGroup g) {
  ReduceGroup::P P = g.p[0];
  bool found = false;
#pragma unroll
  for (int i = 0; i < 24; ++i)
    if (i < g.n && g.p[i].gamma && (int)blockIdx.x >= g.p[i].fblk0) { P = g.p[i]; found = true; }
  if (!found) return;
  ln_affine_finish_one(P.lnws, (P.N + RG_ROWS - 1) / RG_ROWS, P.K, P.dgamma, P.dbeta, ((int)blockIdx.x - P.fblk0) * 256 + threadIdx.x);
}
__global__ void k_reduce_colsum(const float* __restrict__ colsum, float* __restrict__ db,
                                int N, int S) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float d = 0.f;
  for (int s = 0; s < S; ++s) d += colsum[(long)s * N + n];
  db[n] = d;
}

// ----------------------------------------------------------------------------
// weight preparation (once per optimizer step)
// ----------------------------------------------------------------------------
// Wf[n][k] = W[n][k]*gamma[k];  bf[n] = b[n] + sum_k W[n][k]*beta[k]; one wave per row
__global__ void k_fold_ln(const float* __restrict__ W, const float* __restrict__ b,
                          const float* __restrict__ gamma, const float* __restrict__ beta,
                          float* __restrict__ Wf, float* __restrict__ bf, int N, int K) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float a = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(long)n * K + k];
    Wf[(long)n * K + k] = w * gamma[k];
    a += w * beta[k];
  }
  a = wave_sum(a);
  if (lane == 0) bf[n] = (b ? b[n] : 0.f) + a;
}
// out[c][r] = in[r][c]
__global__ void k_transpose(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) t[i][tx] = in[(long)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) out[(long)(c0 + i) * R + r0 + tx] = t[tx][i];
}
// torch conv weight [Co][Ci][3][3] -> fwd pack [9][Co][Ci], bwd-data pack [9][Ci][Co] (taps flipped)
__global__ void k_pack_conv_w(const float* __restrict__ w, float* __restrict__ wp,
                              float* __restrict__ wpt, int Co, int Ci) {
  const long n = (long)Co * Ci * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const int tap = i % 9;
    const long cc = i / 9;
    const int ci = cc % Ci, co = cc / Ci;
    const float v = w[i];
    if (wp) wp[((long)tap * Co + co) * Ci + ci] = v;
    if (wpt) wpt[((long)(8 - tap) * Ci + ci) * Co + co] = v;
  }
}

// ----------------------------------------------------------------------------
// LayerNorm over the last dim C (<= 256), eps 1e-5, biased variance
// (nn.LayerNorm at network_swinir.py:240,248,606,846).  One wave per row.
// ----------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // values per lane

__device__ __forceinline__ void ln_row_stats(const float* __restrict__ x, int C, int lane,
                                             float (&v)[LN_MAXV], float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? x[c] : 0.f;
    s += v[i];
  }
  mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
}

// stats only: st[m] = {mean, rstd}; optionally y = (x-mean)*rstd*g + b
__global__ void __launch_bounds__(256) k_ln_fwd(const float* __restrict__ x, float* __restrict__ st,
                                                float* __restrict__ y, const float* __restrict__ g,
                                                const float* __restrict__ b, long M, int C) {
  const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  float v[LN_MAXV], mean, rstd;
  ln_row_stats(x + m * C, C, lane, v, mean, rstd);
  if (lane == 0 && st) { st[2 * m] = mean; st[2 * m + 1] = rstd; }
  if (y) {
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < C) y[m * C + c] = (v[i] - mean) * rstd * g[c] + b[c];
    }
  }
}
// dx from the gradient w.r.t. the normalised value (dxh), with the residual
// gradient added:  out = res + rstd*(dxh - mean(dxh) - xh*mean(dxh*xh)).
// If g != null the incoming gradient is w.r.t. y = xh*g+b: dxh = dy*g and the
// per-column sums dgamma = sum dy*xh, dbeta = sum dy of a BLOCK go to ws[block][2][C] (plain stores);
// k_ln_affine_finish adds the blocks in order (deterministic: no atomics, no memset).
__global__ void __launch_bounds__(256) k_ln_bwd(const float* __restrict__ dyp, const float* __restrict__ x,
                                                const float* __restrict__ st, const float* __restrict__ res,
                                                const float* __restrict__ g, float* __restrict__ out,
                                                float* __restrict__ ws,
                                                long M, int C, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
  float ag[LN_MAXV] = {0.f, 0.f, 0.f, 0.f}, ab[LN_MAXV] = {0.f, 0.f, 0.f, 0.f};
  // four rows in flight per wave: every load of the group is issued before the first reduction (one row at a time the
  // kernel ran at the latency of 3 loads + 12 shuffles per row: 88 us for 32768 x 180, 1.1 TB/s)
  constexpr int RU = 4;
  const long m_end = min(M, (w + 1) * (long)rows_per_wave);
  for (long m0 = w * rows_per_wave; m0 < m_end; m0 += RU) {
    float dxh[RU][LN_MAXV], xh[RU][LN_MAXV], rr[RU][LN_MAXV], rstd[RU], s1[RU], s2[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long m = min(m0 + u, m_end - 1);           // rows past the end repeat the last one, never stored
      const float mean = st[2 * m];
      rstd[u] = st[2 * m + 1];
      s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        dxh[u][i] = 0.f; xh[u][i] = 0.f; rr[u][i] = 0.f;
        if (c < C) {
          const float dy = dyp[m * C + c];
          xh[u][i] = (x[m * C + c] - mean) * rstd[u];
          rr[u][i] = res ? res[m * C + c] : 0.f;
          dxh[u][i] = g ? dy * g[c] : dy;
          if (g && m0 + u < m_end) { ag[i] += dy * xh[u][i]; ab[i] += dy; }
          s1[u] += dxh[u][i];
          s2[u] += dxh[u][i] * xh[u][i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      s1[u] = wave_sum(s1[u]) / (float)C;
      s2[u] = wave_sum(s2[u]) / (float)C;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (m0 + u < m_end) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
          const int c = lane + 64 * i;
          if (c < C) out[(m0 + u) * C + c] = rr[u][i] + rstd[u] * (dxh[u][i] - s1[u] - xh[u][i] * s2[u]);
        }
      }
    }
  }
  if (g) {      // the block's four waves meet in LDS, then one plain store per column and block
    __shared__ float red[2][4][64 * LN_MAXV];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) { red[0][wv][lane + 64 * i] = ag[i]; red[1][wv][lane + 64 * i] = ab[i]; }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
          float* w2 = ws + (long)blockIdx.x * 2 * C;
          w2[c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
          w2[C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------
// pixel shuffle (index only; nn.PixelShuffle, network_swinir.py:701, network_nlsn.py:108)
//   in  NHWC [B][h][w][Cout*r*r]
//   out NCHW [B][Cout][h*r][w*r]                 (nhwc_out = 0)
//       NHWC [B][h*r][w*r][Cout]                 (nhwc_out = 1)
//   out[b, c, y*r+i, x*r+j] = in[b, y, x, c*r*r + i*r + j]
// inverse = the same mapping read the other way (bwd of the shuffle).
// ----------------------------------------------------------------------------
__global__ void k_pixel_shuffle(const float* __restrict__ in, float* __restrict__ out, int B, int h,
                                int w, int Co, int r, int nhwc_out, int inverse) {
  const long n = (long)B * h * w * Co * r * r;
  const int H = h * r, W = w * r;
  for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < n;
       o += (long)gridDim.x * blockDim.x) {
    // decode the HIGH-res (shuffled) side index o
    int c, X, Y, b;
    long t = o;
    if (nhwc_out) { c = t % Co; t /= Co; X = t % W; t /= W; Y = t % H; b = t / H; }
    else { X = t % W; t /= W; Y = t % H; t /= H; c = t % Co; b = t / Co; }
    const int y = Y / r, i = Y - y * r, x = X / r, j = X - x * r;
    const long li = (((long)b * h + y) * w + x) * ((long)Co * r * r) + (long)c * r * r + i * r + j;
    if (inverse) out[li] = in[o];
    else out[o] = in[li];
  }
}

// Any r, channels-last on both sides (the r x r sub-kernel form of DBPN's / SRFBN's transposed convs, network_dbpn.py:107-143):
// one low-res pixel per block pass.  Its Co*r*r values are one contiguous run and its r*r high-res pixels are runs of Co
// channels: the [Co][r*r] -> [r*r][Co] transposition goes through LDS (row pitch r*r + 1: both sides conflict free), so
// both global sides move whole cache lines -- the element-wise gather reached 0.8 TB/s on the 16-KB runs of r = 8.
// add (forward only): out = shuffled + fac * add, add laid out as out (a projection unit's h1 + h0 / h0 - x behind a
// transposed conv, network_dbpn.py:93-99,128-134).
__global__ void __launch_bounds__(256) k_pixel_shuffle_nhwc_t(const float* __restrict__ in, float* __restrict__ out, long npix,
                                                              int h, int w, int Co, int r, int inverse,
                                                              const float* __restrict__ add, float fac) {
  extern __shared__ float tile[];                 // [Co][r*r + 1]
  const int rr = r * r, P = rr + 1, n = Co * rr;
  const long W = (long)w * r;
  for (long pix = blockIdx.x; pix < npix; pix += gridDim.x) {
    const int x = (int)(pix % w);
    const long t = pix / w;
    const int y = (int)(t % h);
    const long b = t / h;
    const long hi0 = ((b * h + y) * r * W + (long)x * r) * Co;      // high-res pixel (y r, x r), channel 0
    const long lo0 = pix * n;
    __syncthreads();
    if (!inverse) {
      for (int i = threadIdx.x; i < n; i += 256) tile[(i / rr) * P + i % rr] = in[lo0 + i];
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / Co, c = i - s * Co;
        const long o = hi0 + ((long)(s / r) * W + s % r) * Co + c;
        out[o] = add ? tile[c * P + s] + fac * add[o] : tile[c * P + s];
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 256) {
        const int s = i / Co, c = i - s * Co;
        tile[c * P + s] = in[hi0 + ((long)(s / r) * W + s % r) * Co + c];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256) out[lo0 + i] = tile[(i / rr) * P + i % rr];
    }
  }
}

// r = 2, channels-last on both sides (the EDSR upsampler stages, network_nlsn.py:108):
// a lane moves the 4 sub-pixel values of channel c of one low-res pixel as ONE float4
// (in[pix][4c .. 4c+3]) and four coalesced dwords (out[2y+i][2x+j][c]) -- no per-element
// index decoding.  One wave per low-res pixel and 64 channels; blockIdx.y = image row.
__global__ void __launch_bounds__(256) k_pixel_shuffle_r2_nhwc(const float* __restrict__ in, float* __restrict__ out,
                                                               int h, int w, int Co, int inverse) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.y;                  // b * h + y
  const int b = row / h, y = row - b * h;
  const int W = 2 * w;
  const int cgroups = (Co + 63) / 64;
  for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < w * cgroups; item += gridDim.x * 4) {
    const int x = item / cgroups, c = (item - x * cgroups) * 64 + lane;
    if (c >= Co) continue;
    const float* lp = in + ((long)row * w + x) * (4L * Co) + 4 * c;
    float* hp = out + (((long)b * 2 * h + 2 * y) * W + 2 * x) * Co + c;
    if (!inverse) {
      const f32x4 v = *(const f32x4*)lp;
      hp[0] = v.x; hp[Co] = v.y; hp[(long)W * Co] = v.z; hp[(long)W * Co + Co] = v.w;
    }
  }
}
// nearest-neighbour x2 of an NHWC image (F.interpolate(scale_factor=2, mode='nearest'), network_swinir.py:953-960) and its
// adjoint (each low-resolution pixel receives the sum of its 2 x 2 copies).  One thread per (low-res pixel, float4 of channels).
__global__ void __launch_bounds__(256) k_nearest_up2_nhwc(float* __restrict__ lo, float* __restrict__ hi, long npix,
                                                          int h, int w, int C, int adjoint) {
  const int c4n = C >> 2;
  const long n = npix * c4n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c4n;
    const int c4 = (int)(i - pix * c4n);
    const int x = (int)(pix % w);
    const long t = pix / w;
    const int y = (int)(t % h);
    const long b = t / h;
    const long W2 = 2L * w;
    float* hp = hi + (((b * 2 * h + 2 * y) * W2 + 2 * x) * C) + 4 * c4;
    float* lp = lo + pix * C + 4 * c4;
    if (!adjoint) {
      const f32x4 v = *(const f32x4*)lp;
      *(f32x4*)hp = v; *(f32x4*)(hp + C) = v; *(f32x4*)(hp + W2 * C) = v; *(f32x4*)(hp + W2 * C + C) = v;
    } else {
      const f32x4 a = *(const f32x4*)hp, bq = *(const f32x4*)(hp + C), c = *(const f32x4*)(hp + W2 * C),
                  d = *(const f32x4*)(hp + W2 * C + C);
      f32x4 r;
      r.x = (a.x + bq.x) + (c.x + d.x); r.y = (a.y + bq.y) + (c.y + d.y);
      r.z = (a.z + bq.z) + (c.z + d.z); r.w = (a.w + bq.w) + (c.w + d.w);
      *(f32x4*)lp = r;
    }
  }
}
__global__ void __launch_bounds__(256) k_pixel_unshuffle_r2_nhwc(const float* __restrict__ hi, float* __restrict__ lo,
                                                                 int h, int w, int Co) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.y;
  const int b = row / h, y = row - b * h;
  const int W = 2 * w;
  const int cgroups = (Co + 63) / 64;
  for (int item = blockIdx.x * 4 + (threadIdx.x >> 6); item < w * cgroups; item += gridDim.x * 4) {
    const int x = item / cgroups, c = (item - x * cgroups) * 64 + lane;
    if (c >= Co) continue;
    const float* hp = hi + (((long)b * 2 * h + 2 * y) * W + 2 * x) * Co + c;
    f32x4 v;
    v.x = hp[0]; v.y = hp[Co]; v.z = hp[(long)W * Co]; v.w = hp[(long)W * Co + Co];
    *(f32x4*)(lo + ((long)row * w + x) * (4L * Co) + 4 * c) = v;
  }
}

// ----------------------------------------------------------------------------
// losses (dlib/loss/main.py:45-99): fused value + gradient
//   mode 0: L1  lam*mean(|e|*w?)    grad = lam*sign(e)*w?/n
//   mode 1: L2  lam*mean(e^2)       grad = 2*lam*e/n
// partial sums go to part[gridDim.x] (double); k_sum_partials finishes.
// ----------------------------------------------------------------------------
// g *= (a > 0): the backward of a ReLU whose output a is kept (VDSR's last ConvReLU in front of the 64 -> 1 conv)
__global__ void __launch_bounds__(256) k_relu_mask(float* __restrict__ g, const float* __restrict__ a, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (!(a[i] > 0.f)) g[i] = 0.f;
}

// x = x > 0 ? x : alpha * x in place (nn.LeakyReLU after the 1-channel edge conv)
__global__ void __launch_bounds__(256) k_leaky(float* __restrict__ x, long n, float alpha) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    x[i] = v > 0.f ? v : v * alpha;
  }
}
// g *= (a > 0 ? 1 : alpha): the backward of a LeakyReLU(alpha > 0) whose output a is kept (sign(output) = sign(input))
__global__ void __launch_bounds__(256) k_leaky_mask(float* __restrict__ g, const float* __restrict__ a, long n, float alpha) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (!(a[i] > 0.f)) g[i] *= alpha;
}

__global__ void __launch_bounds__(256) k_loss_l1l2(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                   const float* __restrict__ wgt, float* __restrict__ grad,
                                                   double* __restrict__ part, long n, int mode, float lam,
                                                   int grad_accum) {
  __shared__ double sh[4];
  double acc = 0.0;
  const float gs = lam / (float)n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float e = pred[i] - tgt[i];
    float v, g;
    if (mode == 0) {
      v = fabsf(e);
      g = (e > 0.f) ? gs : (e < 0.f ? -gs : 0.f);
      if (wgt) { v *= wgt[i]; g *= wgt[i]; }
    } else {
      v = e * e;
      g = 2.f * gs * e;
    }
    acc += (double)v;
    if (grad) grad[i] = grad_accum ? grad[i] + g : g;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(256) k_sum(const float* __restrict__ x, double* __restrict__ part, long n) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += (double)x[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[0] (+)= scale * sum(part[0..n))  as float
__global__ void __launch_bounds__(1024) k_sum_partials(const double* __restrict__ part, int n, double scale,
                               float* __restrict__ out, int accum) {
  __shared__ double sh[16];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += part[i];
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    out[0] = (accum ? out[0] : 0.f) + (float)(t * scale);
  }
}

// ----------------------------------------------------------------------------
// optimizers on the flat parameter buffer (utils_instance.py:216-247)
// ----------------------------------------------------------------------------
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                       float wd, float bc1, float bc2_sqrt, float gscale, const int* __restrict__ skip) {
  if (skip && *skip) return;   // non-finite loss: skip the update (model_plain.py:344-346)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] * b1 + (1.f - b1) * gi;
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
__global__ void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                      long n, float lr, float momentum, float wd, int nesterov, int first,
                      float gscale, const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float d = gi;
    if (momentum != 0.f) {
      const float bi = first ? gi : buf[i] * momentum + gi;
      buf[i] = bi;
      d = nesterov ? gi + momentum * bi : bi;
    }
    p[i] = pi - lr * d;
  }
}
// device-resident step counter: counter += (skip == 0).  The *_dc optimizer kernels read it
// (they run after this launch on the same stream), so a skipped step advances neither
// Adam's bias correction nor SGD's first-step momentum initialisation.
__global__ void k_optim_tick(const int* __restrict__ skip, int* __restrict__ counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip)) *counter += 1;
}
__global__ void k_adam_dc(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                          float* __restrict__ v, long n, const int* __restrict__ counter, float lr, float b1,
                          float b2, float eps, float wd, float gscale, const int* __restrict__ skip,
                          const float* __restrict__ lr_dev) {
  if (skip && *skip) return;
  if (lr_dev) lr = *lr_dev;       // learning rate from device memory: a captured launch replays with the current one
  const float step = (float)*counter;
  const float bc1 = 1.f - powf(b1, step);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, step));
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] * b1 + (1.f - b1) * gi;
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
__global__ void k_sgd_dc(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                         long n, const int* __restrict__ counter, float lr, float momentum, float wd,
                         int nesterov, float gscale, const int* __restrict__ skip,
                         const float* __restrict__ lr_dev) {
  if (skip && *skip) return;
  if (lr_dev) lr = *lr_dev;
  const int first = *counter == 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float d = gi;
    if (momentum != 0.f) {
      const float bi = first ? gi : buf[i] * momentum + gi;
      buf[i] = bi;
      d = nesterov ? gi + momentum * bi : bi;
    }
    p[i] = pi - lr * d;
  }
}
// ---- gradient clipping by the global L2 norm (torch.nn.utils.clip_grad_norm_, model_plain.py:350-361) --------------------
// Two-stage fixed-order reduction (no float atomics: a replayed or data-parallel step clips bit for bit like the eager one):
// block b sums the squares of ITS slice of the flat gradient in double, one block joins the GC_BLOCKS partials in order,
// writes norm = gscale * sqrt(sum) and coef = min(1, max_norm / (norm + 1e-6)) (a NaN norm gives a NaN coefficient, as
// torch's clamp does), and a third launch scales the gradient by the device-resident coefficient.
constexpr int GC_BLOCKS = 1024;
__global__ void __launch_bounds__(256) k_sumsq_partials(const float* __restrict__ g, long n, double* __restrict__ part) {
  __shared__ double sh[4];
  const long per = ((n + GC_BLOCKS - 1) / GC_BLOCKS + 3) / 4 * 4;
  const long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double a = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) { const double v = g[i]; a += v * v; }
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void __launch_bounds__(GC_BLOCKS) k_clip_coef(const double* __restrict__ part, float gscale, float max_norm,
                                                         float* __restrict__ out) {
  __shared__ double sh[GC_BLOCKS / 64];
  double a = wave_sum_d(part[threadIdx.x]);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < GC_BLOCKS / 64; ++i) t += sh[i];
    const float norm = (float)(sqrt(t) * (double)gscale);
    const float c = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = c < 1.f ? c : (c != c ? c : 1.f);
  }
}
__global__ void k_scale_dev(float* __restrict__ g, long n, const float* __restrict__ coef) {
  const float c = *coef;
  if (c == 1.f) return;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] *= c;
}
// exponential moving average of the weights (ModelBase.update_E, model_base.py:213-219): e = e * decay + p * (1 - decay);
// skipped with the optimizer update on a non-finite loss (optimize_parameters returns before update_E, model_plain.py:344-346)
__global__ void k_ema(float* __restrict__ e, const float* __restrict__ p, long n, float decay, float alpha,
                      const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    e[i] = __fadd_rn(__fmul_rn(e[i], decay), __fmul_rn(p[i], alpha));
}
// finite check: flag[0] |= any(!isfinite(x))
__global__ void k_nonfinite(const float* __restrict__ x, long n, int* __restrict__ flag) {
  int bad = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    bad |= !isfinite(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void k_axpby(float* __restrict__ y, const float* __restrict__ x, long n, float a, float b) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}

inline int ew_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ============================ C-ABI (include/srhip.h) ========================
extern "C" {

int srhip_reduce_linear_wgrad(const float* part, const float* colsum, int S, float* dW, float* db,
                              int N, int K, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)N * K;
  const int nb = (colsum && db) ? N : 0;
  const int mb = ew_grid(n);
  hipLaunchKernelGGL(k_reduce_slices, dim3(mb + sr_cdiv(nb, 256)), dim3(256), 0, st, part, dW, n, S, 0.f,
                     colsum, db, nb, mb);
  SR_LAUNCH_CHECK("reduce_linear_wgrad");
  return 0;
}

int srhip_reduce_ln_linear_wgrad(const float* part, const float* colsum, int S, const float* W,
                                 const float* gamma, const float* beta, float* dW, float* db,
                                 float* dgamma, float* dbeta, int N, int K, float* ln_ws,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  SR_REQUIRE(ln_ws != nullptr, "reduce_ln_linear_wgrad: workspace of srhip_ln_affine_ws(N, K) floats required");
  hipLaunchKernelGGL(k_fin_ln_linear, dim3(sr_cdiv(K, 64), sr_cdiv(N, 4)), dim3(256), 0, st, part,
                     colsum, S, W, gamma, beta, dW, db, ln_ws, N, K);
  hipLaunchKernelGGL(k_ln_affine_finish, dim3(sr_cdiv(2 * K, 256)), dim3(256), 0, st, ln_ws, sr_cdiv(N, 4), K,
                     dgamma, dbeta);
  SR_LAUNCH_CHECK("reduce_ln_linear_wgrad");
  return 0;
}

long srhip_ln_affine_ws(int N, int K) { return (long)sr_cdiv(N, 4) * 2 * K; }

int srhip_reduce_wgrad_grouped(const srhip_reduce_problem* probs, int nprob, int S, void* stream) {
  SR_REQUIRE(nprob >= 1 && nprob <= 24 && S > 0, "reduce_wgrad_grouped: 1..24 problems, S > 0");
  ReduceGroup g;
  memset(&g, 0, sizeof(g));
  g.n = nprob; g.S = S;
  int blocks = 0, fblocks = 0;
  for (int i = 0; i < nprob; ++i) {
    const srhip_reduce_problem& q = probs[i];
    SR_REQUIRE(q.part && q.colsum && q.dW && q.db && q.N > 0 && q.K > 0, "reduce_wgrad_grouped: problem %d incomplete", i);
    SR_REQUIRE(!q.gamma || (q.W && q.beta && q.dgamma && q.dbeta && q.ln_ws), "reduce_wgrad_grouped: LayerNorm problem %d incomplete", i);
    ReduceGroup::P& d = g.p[i];
    d.lnws = q.ln_ws;
    d.fblk0 = fblocks;
    if (q.gamma) fblocks += sr_cdiv(2 * q.K, 256);
    d.part = q.part; d.colsum = q.colsum; d.W = q.W; d.gamma = q.gamma; d.beta = q.beta;
    d.dW = q.dW; d.db = q.db; d.dgamma = q.dgamma; d.dbeta = q.dbeta; d.N = q.N; d.K = q.K;
    d.blk0 = blocks; d.kblocks = sr_cdiv(q.K, 64);
    blocks += d.kblocks * sr_cdiv(q.N, RG_ROWS);
  }
  hipLaunchKernelGGL(k_reduce_group, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  if (fblocks) hipLaunchKernelGGL(k_ln_affine_finish_group, dim3(fblocks), dim3(256), 0, (hipStream_t)stream, g);
  SR_LAUNCH_CHECK("reduce_wgrad_grouped");
  return 0;
}

int srhip_reduce_conv_wgrad(const float* part, const float* colsum, int S, float* dW, float* db,
                            int Co, int Ci, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int mb = ew_grid((long)Co * Ci * 9);
  const bool bias = colsum && db;
  hipLaunchKernelGGL(k_reduce_conv_w, dim3(mb + (bias ? 1 : 0)), dim3(256), 0, st, part, dW, Co, Ci, S,
                     colsum, db, mb);
  SR_LAUNCH_CHECK("reduce_conv_wgrad");
  return 0;
}

int srhip_conv3x3_wgrad_batched_plan(int n, int B, int H, int W, int Cout, int Cin, int* S,
                                     long* part_floats_per_item) {
  SR_REQUIRE(n >= 1 && n <= 40, "conv3x3_wgrad_batched_plan: 1..40 problems (got %d)", n);
  return sr_conv_wgrad_batched_plan(n, B * H * W, Cout, Cin, S, part_floats_per_item);
}

int srhip_conv3x3_wgrad_batched_bx3(const srhip_conv_wgrad_item* items, int n, long lddy, long ldx, int B, int H,
                                    int W, int Cout, int Cin, float* part, float* part_colsum, int S,
                                    void* stream) {
  SR_REQUIRE(n >= 1 && n <= 40, "conv3x3_wgrad_batched: 1..40 problems (got %d)", n);
  SR_REQUIRE(items && part && part_colsum, "conv3x3_wgrad_batched: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  TnArgs p;
  memset(&p, 0, sizeof(p));
  p.lda = lddy; p.ldb = ldx; p.M = B * H * W; p.NI = Cout; p.NJ = Cin;
  p.part = part; p.part_colsum = part_colsum; p.S = S; p.conv = 1; p.batch = B; p.H = H; p.Wd = W;
  const float* A[40];
  const float* Bp[40];
  ConvReduceBatch out;
  memset(&out, 0, sizeof(out));
  for (int k = 0; k < n; ++k) {
    SR_REQUIRE(items[k].dY && items[k].X && items[k].dW, "conv3x3_wgrad_batched: item %d has a NULL pointer", k);
    A[k] = items[k].dY; Bp[k] = items[k].X; out.dw[k] = items[k].dW; out.db[k] = items[k].db;
  }
  const long pstride = (long)S * 9 * Cout * Cin, cstride = (long)S * Cout;
  if (int rc = sr_conv_wgrad_batched_tnb(p, A, Bp, n, pstride, cstride, st)) return rc;
  const int mb = ew_grid((long)Cout * Cin * 9) > 64 ? 64 : ew_grid((long)Cout * Cin * 9);
  hipLaunchKernelGGL(k_reduce_conv_w_batched, dim3(mb + 1, n), dim3(256), 0, st, part, pstride, part_colsum,
                     cstride, Cout, Cin, S, out, mb);
  SR_LAUNCH_CHECK("reduce_conv_wgrad_batched");
  return 0;
}

int srhip_fold_layernorm(const float* W, const float* b, const float* gamma, const float* beta,
                         float* Wf, float* bf, int N, int K, void* stream) {
  hipLaunchKernelGGL(k_fold_ln, dim3(sr_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, W, b, gamma,
                     beta, Wf, bf, N, K);
  SR_LAUNCH_CHECK("fold_layernorm");
  return 0;
}

int srhip_transpose(const float* in, float* out, int R, int C, void* stream) {
  hipLaunchKernelGGL(k_transpose, dim3(sr_cdiv(C, 32), sr_cdiv(R, 32)), dim3(256), 0,
                     (hipStream_t)stream, in, out, R, C);
  SR_LAUNCH_CHECK("transpose");
  return 0;
}

int srhip_pack_conv_weight(const float* w, float* wp, float* wpt, int Co, int Ci, void* stream) {
  hipLaunchKernelGGL(k_pack_conv_w, dim3(ew_grid((long)Co * Ci * 9)), dim3(256), 0,
                     (hipStream_t)stream, w, wp, wpt, Co, Ci);
  SR_LAUNCH_CHECK("pack_conv_weight");
  return 0;
}

int srhip_layernorm_fwd(const float* x, float* stats, float* y, const float* gamma, const float* beta,
                        long M, int C, void* stream) {
  SR_REQUIRE(C <= 64 * LN_MAXV, "layernorm: C=%d > %d unsupported", C, 64 * LN_MAXV);
  SR_REQUIRE(!y || (gamma && beta), "layernorm_fwd: y requested without gamma/beta");
  if (M <= 0) return 0;
  hipLaunchKernelGGL(k_ln_fwd, dim3(sr_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, stats, y,
                     gamma, beta, M, C);
  SR_LAUNCH_CHECK("layernorm_fwd");
  return 0;
}

static inline int ln_bwd_rpw(long M) { return (int)((M + 8191) / 8192); }   // rows per wave with the affine gradients: 4 at T = 32768
long srhip_layernorm_bwd_ws(long M, int C) {
  const int rpw = ln_bwd_rpw(M);
  return (long)sr_cdiv((M + rpw - 1) / rpw, 4) * 2 * C;
}
int srhip_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* res,
                        const float* gamma, float* out, float* dgamma, float* dbeta, float* workspace, long M,
                        int C, void* stream) {
  SR_REQUIRE(C <= 64 * LN_MAXV, "layernorm: C=%d > %d unsupported", C, 64 * LN_MAXV);
  SR_REQUIRE(!gamma || (dgamma && dbeta && workspace),
             "layernorm_bwd: gamma given without dgamma / dbeta / workspace (srhip_layernorm_bwd_ws floats)");
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int rpw = gamma ? ln_bwd_rpw(M) : 1;      // fewer, longer waves so that the per-block partial rows stay few
  const long waves = (M + rpw - 1) / rpw;
  const int blocks = sr_cdiv(waves, 4);
  hipLaunchKernelGGL(k_ln_bwd, dim3(blocks), dim3(256), 0, st, dy, x, stats, res, gamma, out, workspace, M, C, rpw);
  if (gamma) {
    if (blocks >= 64)
      hipLaunchKernelGGL(k_ln_affine_finish_wide, dim3(sr_cdiv(2 * C, 64)), dim3(1024), 0, st, workspace, blocks, C, dgamma, dbeta);
    else
      hipLaunchKernelGGL(k_ln_affine_finish, dim3(sr_cdiv(2 * C, 256)), dim3(256), 0, st, workspace, blocks, C, dgamma, dbeta);
  }
  SR_LAUNCH_CHECK("layernorm_bwd");
  return 0;
}

static int pixel_shuffle_any(const float* in, float* out, int B, int h, int w, int Co, int r, int nhwc_out, int inverse,
                             const float* add, float fac, void* stream);
int srhip_pixel_shuffle(const float* in, float* out, int B, int h, int w, int Co, int r,
                        int nhwc_out, int inverse, void* stream) {
  return pixel_shuffle_any(in, out, B, h, w, Co, r, nhwc_out, inverse, nullptr, 0.f, stream);
}
int srhip_pixel_shuffle_add(const float* in, float* out, int B, int h, int w, int Co, int r, const float* add, float fac,
                            void* stream) {
  SR_REQUIRE(add, "pixel_shuffle_add: null addend");
  return pixel_shuffle_any(in, out, B, h, w, Co, r, 1, 0, add, fac, stream);
}
static int pixel_shuffle_any(const float* in, float* out, int B, int h, int w, int Co, int r, int nhwc_out, int inverse,
                             const float* add, float fac, void* stream) {
  SR_REQUIRE(r >= 1 && Co >= 1, "pixel_shuffle: bad r/Co");
  if (!add && r == 2 && nhwc_out && (long)B * h < 65536 && ((size_t)in & 15) == 0 && ((size_t)out & 15) == 0) {
    // fast path: in = low-res side [B][h][w][4*Co], out = high-res side [B][2h][2w][Co]
    // (inverse: `in` is the high-res side, `out` the low-res one)
    dim3 grid(sr_cdiv((long)w * sr_cdiv(Co, 64), 4) < 64 ? sr_cdiv((long)w * sr_cdiv(Co, 64), 4) : 64, B * h);
    if (inverse) hipLaunchKernelGGL(k_pixel_unshuffle_r2_nhwc, grid, dim3(256), 0, (hipStream_t)stream, in, out, h, w, Co);
    else hipLaunchKernelGGL(k_pixel_shuffle_r2_nhwc, grid, dim3(256), 0, (hipStream_t)stream, in, out, h, w, Co, 0);
    SR_LAUNCH_CHECK("pixel_shuffle_r2");
    return 0;
  }
  const long n = (long)B * h * w * Co * r * r;
  if (n == 0) return 0;
  if (nhwc_out && r > 1 && (long)Co * (r * r + 1) * 4 <= 48 * 1024 && Co * r * r >= 256) {
    const long npix = (long)B * h * w;
    hipLaunchKernelGGL(k_pixel_shuffle_nhwc_t, dim3((unsigned)(npix < 65536 ? npix : 65536)), dim3(256), (size_t)Co * (r * r + 1) * 4,
                       (hipStream_t)stream, in, out, npix, h, w, Co, r, inverse, add, fac);
    SR_LAUNCH_CHECK("pixel_shuffle_nhwc");
    return 0;
  }
  SR_REQUIRE(!add, "pixel_shuffle_add: runs on the channels-last transposing kernel (Co (r r + 1) * 4 <= 48 KB, Co r r >= 256; Co=%d r=%d)", Co, r);
  hipLaunchKernelGGL(k_pixel_shuffle, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, in, out,
                     B, h, w, Co, r, nhwc_out, inverse);
  SR_LAUNCH_CHECK("pixel_shuffle");
  return 0;
}

// workspace: 2048 doubles
int srhip_loss_l1l2(const float* pred, const float* target, const float* weight, float* grad,
                    float* loss_out, double* workspace, long n, int mode, float lam, int grad_accum,
                    int loss_accum, void* stream) {
  SR_REQUIRE(n > 0, "loss: empty input");
  SR_REQUIRE(mode == 0 || mode == 1, "loss: mode %d", mode);
  hipStream_t st = (hipStream_t)stream;
  const int g = ew_grid(n);
  hipLaunchKernelGGL(k_loss_l1l2, dim3(g), dim3(256), 0, st, pred, target, weight, grad, workspace, n,
                     mode, lam, grad_accum);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st, workspace, g, (double)lam / (double)n,
                     loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_l1l2");
  return 0;
}

// workspace: 2048 doubles
int srhip_sum(const float* x, long n, float* out, double* workspace, void* stream) {
  SR_REQUIRE(n > 0, "sum: empty input");
  hipStream_t st = (hipStream_t)stream;
  const int g = ew_grid(n);
  hipLaunchKernelGGL(k_sum, dim3(g), dim3(256), 0, st, x, workspace, n);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st, workspace, g, 1.0, out, 0);
  SR_LAUNCH_CHECK("sum");
  return 0;
}

int srhip_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float b1,
                    float b2, float eps, float wd, float gscale, const int* skip_flag, void* stream) {
  if (n <= 0) return 0;
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(k_adam, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     b1, b2, eps, wd, bc1, sqrtf(bc2), gscale, skip_flag);
  SR_LAUNCH_CHECK("adam_step");
  return 0;
}

int srhip_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                   int nesterov, int first, float gscale, const int* skip_flag, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_sgd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr,
                     momentum, wd, nesterov, first, gscale, skip_flag);
  SR_LAUNCH_CHECK("sgd_step");
  return 0;
}

int srhip_optim_tick(const int* skip_flag, int* counter, void* stream) {
  SR_REQUIRE(counter != nullptr, "optim_tick: counter is NULL");
  hipLaunchKernelGGL(k_optim_tick, dim3(1), dim3(64), 0, (hipStream_t)stream, skip_flag, counter);
  SR_LAUNCH_CHECK("optim_tick");
  return 0;
}

int srhip_adam_step_dc(float* p, const float* g, float* m, float* v, long n, const int* counter, float lr,
                       float b1, float b2, float eps, float wd, float gscale, const int* skip_flag,
                       const float* lr_dev, void* stream) {
  if (n <= 0) return 0;
  SR_REQUIRE(counter != nullptr, "adam_step_dc: counter is NULL");
  hipLaunchKernelGGL(k_adam_dc, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, counter,
                     lr, b1, b2, eps, wd, gscale, skip_flag, lr_dev);
  SR_LAUNCH_CHECK("adam_step_dc");
  return 0;
}

int srhip_sgd_step_dc(float* p, const float* g, float* buf, long n, const int* counter, float lr,
                      float momentum, float wd, int nesterov, float gscale, const int* skip_flag,
                      const float* lr_dev, void* stream) {
  if (n <= 0) return 0;
  SR_REQUIRE(counter != nullptr, "sgd_step_dc: counter is NULL");
  hipLaunchKernelGGL(k_sgd_dc, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, counter,
                     lr, momentum, wd, nesterov, gscale, skip_flag, lr_dev);
  SR_LAUNCH_CHECK("sgd_step_dc");
  return 0;
}

long srhip_grad_norm_clip_ws(void) { return (long)GC_BLOCKS * 8; }

int srhip_grad_norm_clip(float* g, long n, float gscale, float max_norm, float* norm_coef, void* workspace,
                         long workspace_bytes, void* stream) {
  if (n <= 0) return 0;
  SR_REQUIRE(g && norm_coef && workspace, "grad_norm_clip: null operand");
  SR_REQUIRE(workspace_bytes >= (long)GC_BLOCKS * 8 && ((uintptr_t)workspace & 7) == 0,
             "grad_norm_clip: workspace of %ld bytes, 8-byte aligned (srhip_grad_norm_clip_ws)", (long)GC_BLOCKS * 8);
  SR_REQUIRE(max_norm > 0.f, "grad_norm_clip: max_norm = %g must be positive", (double)max_norm);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sumsq_partials, dim3(GC_BLOCKS), dim3(256), 0, st, g, n, (double*)workspace);
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(GC_BLOCKS), 0, st, (const double*)workspace, gscale, max_norm, norm_coef);
  hipLaunchKernelGGL(k_scale_dev, dim3(ew_grid(n)), dim3(256), 0, st, g, n, norm_coef + 1);
  SR_LAUNCH_CHECK("grad_norm_clip");
  return 0;
}

int srhip_ema_update(float* e, const float* p, long n, float decay, const int* skip_flag, void* stream) {
  if (n <= 0) return 0;
  SR_REQUIRE(e && p, "ema_update: null operand");
  SR_REQUIRE(decay >= 0.f && decay <= 1.f, "ema_update: decay = %g outside [0, 1]", (double)decay);
  hipLaunchKernelGGL(k_ema, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, e, p, n, decay,
                     (float)(1.0 - (double)decay), skip_flag);
  SR_LAUNCH_CHECK("ema_update");
  return 0;
}

int srhip_nonfinite_flag(const float* x, long n, int* flag, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_nonfinite, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, flag);
  SR_LAUNCH_CHECK("nonfinite_flag");
  return 0;
}

int srhip_relu_mask(float* g, const float* a, long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_relu_mask, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, g, a, n);
  SR_LAUNCH_CHECK("relu_mask");
  return 0;
}

int srhip_nearest_up2_nhwc(float* lo, float* hi, int B, int h, int w, int C, int adjoint, void* stream) {
  SR_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "nearest_up2: C must be a multiple of 4 (C=%d)", C);
  SR_REQUIRE(lo && hi, "nearest_up2: null image");
  const long npix = (long)B * h * w;
  hipLaunchKernelGGL(k_nearest_up2_nhwc, dim3(ew_grid(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     lo, hi, npix, h, w, C, adjoint);
  SR_LAUNCH_CHECK("nearest_up2");
  return 0;
}

int srhip_leaky_relu(float* x, long n, float alpha, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_leaky, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, alpha);
  SR_LAUNCH_CHECK("leaky_relu");
  return 0;
}

int srhip_leaky_relu_mask(float* g, const float* a, long n, float alpha, void* stream) {
  if (n <= 0) return 0;
  SR_REQUIRE(alpha > 0.f, "leaky_relu_mask: the sign of the kept output decides the slope only for alpha > 0");
  hipLaunchKernelGGL(k_leaky_mask, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, g, a, n, alpha);
  SR_LAUNCH_CHECK("leaky_relu_mask");
  return 0;
}

int srhip_axpby(float* y, const float* x, long n, float a, float b, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, a, b);
  SR_LAUNCH_CHECK("axpby");
  return 0;
}

}  // extern "C"
