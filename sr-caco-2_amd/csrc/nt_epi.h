// Shared epilogue of the NT kernels (gemm_nt.hip: exact-f32 MFMA; gemm_ntb.hip:
// 3-way bf16 split MFMA): bias, ReLU, residual + per-sample scale (DropPath),
// GELU' and ReLU' gating, applied to the 32x32 accumulator tiles of one wave.
#pragma once
#include "common.h"
#include "kernels.h"

// Sum of a[q] over the 32 lanes of a wave half for all 16 q at once (halving
// butterfly: 8+4+2+1+1 exchanges instead of 16x5).  Returns the total of row
// q* = bit4*8 + bit3*4 + bit2*2 + bit1 of the lane index; a[] is destroyed.
__device__ __forceinline__ float half_reduce16(float (&a)[16], int lane) {
#pragma unroll
  for (int n = 8, m = 16; n >= 1; n >>= 1, m >>= 1) {
    const bool up = lane & m;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float s = up ? a[i] : a[i + n], k = up ? a[i + n] : a[i];
      a[i] = k + __shfl_xor(s, m, 64);
    }
  }
  return a[0] + __shfl_xor(a[0], 1, 64);
}
__device__ __forceinline__ int half_reduce16_row(int lane) {
  return ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

// LayerNorm statistics {mean, rstd} (eps 1e-5, biased variance: nn.LayerNorm,
// network_swinir.py:240,248) of the rows a block has just produced, for the NEXT
// LayerNorm-prologue GEMM: saves the separate pass over the activation.  Two-pass
// like the reference (mean, then squared deviations).  One N block, WM = 1.
template <int WN>
__device__ __forceinline__ void nt_row_stats(const NtArgs& p, f32x16 (&acc)[1][WN], int lane, int wm, int wn,
                                             int m0, int nvalid, float* red) {
  const int r = lane & 31;
  const float inv = 1.0f / (float)p.N;
  float a[16], mean[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < WN; ++j) s += ((wn * WN + j) * 32 + r < nvalid) ? acc[0][j][q] : 0.f;
    a[q] = s;
  }
  const float t1 = half_reduce16(a, lane);
  const int qs = half_reduce16_row(lane);
  __syncthreads();                                  // staging buffers are dead from here on
  if (!(lane & 1)) red[wn * 64 + wm * 32 + mfma_row(qs, lane)] = t1;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int lr = wm * 32 + mfma_row(q, lane);
    mean[q] = (red[lr] + red[64 + lr]) * inv;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const float d = ((wn * WN + j) * 32 + r < nvalid) ? acc[0][j][q] - mean[q] : 0.f;
      s += d * d;
    }
    a[q] = s;
  }
  const float t2 = half_reduce16(a, lane);
  if (!(lane & 1)) red[128 + wn * 64 + wm * 32 + mfma_row(qs, lane)] = t2;
  __syncthreads();
  if (wn == 0 && r == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int lr = wm * 32 + mfma_row(q, lane), g = m0 + lr;
      if (g < p.M) {
        const float var = (red[128 + lr] + red[192 + lr]) * inv;
        *(float2*)(p.stats_out + 2 * (long)g) = float2{mean[q], rsqrtf(var + 1e-5f)};
      }
    }
  }
}

template <int WM, int WN, bool CONV>
__device__ __forceinline__ void nt_epilogue(const NtArgs& p, f32x16 (&acc)[WM][WN], int lane, int wm,
                                            int wn, int n0, int nvalid, int m0, int img, int y0,
                                            int x0) {
  constexpr int BM = 64 * WM;
  const int r = lane & 31;
  // per-sample scale (DropPath): one sample per block whenever the sample's
  // row count is a multiple of the block's rows, else looked up per row.
  float blk_s = p.alpha;
  bool per_row = false;
  if (p.rowscale) {
    if (CONV) blk_s *= p.rowscale[img];
    else if (p.rows_per_scale % BM == 0) blk_s *= p.rowscale[m0 / p.rows_per_scale];
    else per_row = true;
  }
  // The epilogue mode is block-uniform: switch OUTSIDE the element loops, and
  // batch the 16 loads of a tile ahead of the math (one wait per tile, not one
  // per element).
  const bool needR = p.R != nullptr && p.epi >= 2 && p.epi != 9;
  const float slope = (p.epi == 9 || p.epi == 10) ? ldg_f(p.slope) : 0.f;
  if (p.epi == 9 || p.epi == 10) blk_s = 1.f;       // alpha is the residual's factor there
  // pass 1: ALL residual / gate loads of the wave (WM x WN tiles x 16) are issued
  // before any of them is used: the epilogue is latency bound otherwise (every block
  // of the launch reaches it at the same moment)
  constexpr bool BATCH = WM * WN <= 3;             // larger wave tiles would spill: per-tile batches
  float rvall[BATCH ? WM : 1][BATCH ? WN : 1][16];
#pragma unroll
  for (int j = 0; j < (BATCH ? WN : 0); ++j) {
    const int col = (wn * WN + j) * 32 + r;
    const bool cok = col < nvalid;
    const int gn = n0 + col;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int mt = wm * WM + i;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int lr = mfma_row(q, lane);
        int g;
        if (CONV) {
          const int y = y0 + 2 * mt + (lr >> 4), x = x0 + (lr & 15);
          g = (cok && y < p.H && x < p.Wd) ? (img * p.H + y) * p.Wd + x : -1;
        } else {
          const int gg = m0 + mt * 32 + lr;
          g = (cok && gg < p.M) ? gg : -1;
        }
        rvall[i][j][q] = (needR && g >= 0) ? p.R[(long)g * p.ldr + gn] : 0.f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = (wn * WN + j) * 32 + r;        // column inside the N block
    const bool cok = col < nvalid;
    const int gn = n0 + col;
    // conv + PixelShuffle(2): column gn = sp*(N/4) + cc is channel cc of sub-pixel sp (torch channel cc*4 + sp)
    const bool shuf = CONV && p.ps == 1;
    const int fs = p.N >> 2, sp = shuf ? gn / fs : 0, cc = shuf ? gn - sp * fs : gn;
    const float bv = (cok && p.bias) ? p.bias[shuf ? cc * 4 + sp : gn] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int mt = wm * WM + i;
      int grow[16];                                // global row (token / pixel), -1 = masked
      float (&rv)[16] = rvall[BATCH ? i : 0][BATCH ? j : 0];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int lr = mfma_row(q, lane);          // row inside the 32-row tile
        if (CONV) {
          const int y = y0 + 2 * mt + (lr >> 4), x = x0 + (lr & 15);
          if (shuf) grow[q] = (cok && y < p.H && x < p.Wd) ? ((img * 2 * p.H + 2 * y + (sp >> 1)) * 2 * p.Wd + 2 * x + (sp & 1)) : -1;
          else grow[q] = (cok && y < p.H && x < p.Wd) ? (img * p.H + y) * p.Wd + x : -1;
        } else {
          const int g = m0 + mt * 32 + lr;
          grow[q] = (cok && g < p.M) ? g : -1;
        }
      }
      if (!BATCH) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          rv[q] = (needR && grow[q] >= 0) ? p.R[(long)grow[q] * p.ldr + gn] : 0.f;
      }
      f32x16& v = acc[i][j];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] += bv;
      switch (p.epi) {
        case 1:
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = fmaxf(v[q], 0.f);
          break;
        case 2:
          if (per_row) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (grow[q] >= 0) v[q] *= p.rowscale[grow[q] / p.rows_per_scale];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = v[q] * blk_s + rv[q];
          break;
        case 3:
          if (per_row) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (grow[q] >= 0) v[q] *= p.rowscale[grow[q] / p.rows_per_scale];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            // Backward-only evaluation of Phi(x) and phi(x): one exp shared by both
            // (e1 = exp(-x^2/2), exp(-z^2) = e1 for z = x/sqrt(2)) and erf from Abramowitz &
            // Stegun 7.1.26 (|error| <= 1.5e-7, i.e. <= 7.5e-8 on Phi): gradient-grade
            // accuracy at 40 % of the vector instructions of erff + expf.  The FORWARD GELU
            // (GEMM prologue) keeps the exact erff.
            const float x = rv[q];
            const float z = fabsf(x) * 0.70710678118654752440f;
            const float e1 = __expf(-0.5f * x * x);
            const float t = __frcp_rn(1.0f + 0.3275911f * z);
            const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f +
                               t * (-1.453152027f + t * 1.061405429f))));
            const float erfz = 1.0f - poly * e1;                  // erf(|x| / sqrt 2)
            const float cdf = 0.5f * (1.0f + copysignf(erfz, x));
            const float pdf = 0.39894228040143267794f * e1;
            v[q] = v[q] * blk_s * (cdf + x * pdf);
            rv[q] = x * cdf;                     // gelu(R), stored below when aux is given
          }
          if (p.aux) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (grow[q] >= 0) p.aux[(long)grow[q] * p.ldaux + gn] = rv[q];
          }
          break;
        case 4:
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = rv[q] > 0.f ? v[q] : 0.f;
          break;
        case 6:            // LeakyReLU(alpha) (network_swinir.py:857,938)
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * p.alpha;
          break;
        case 7:            // its backward: R = the activation's OUTPUT (same sign as its input)
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = rv[q] > 0.f ? v[q] : v[q] * p.alpha;
          break;
        case 8:            // residual, then ReLU (DRRN's unit, network_drrn.py:58-62)
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = fmaxf(v[q] * blk_s + rv[q], 0.f);
          break;
        case 9:            // PReLU, one slope (network_dbpn.py ConvBlock / DeconvBlock activation)
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = v[q] > 0.f ? v[q] : slope * v[q];
          break;
        // (epilogue 11, nn.GELU(), lives in the 64-column conv kernel's own 16-byte epilogue only -- gemm_ntw.hip: erff in this
        // shared one put 448 bytes of scratch into the widest exact-f32 tiles)
        case 10:           // PReLU, then + alpha * R (the projection units' l0 - x / h1 + h0)
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = (v[q] > 0.f ? v[q] : slope * v[q]) + p.alpha * rv[q];
          break;
        default:
          break;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (grow[q] >= 0) p.C[(long)grow[q] * p.ldc + cc] = v[q];
    }
  }
}

// Epilogue 5 -- LayerNorm backward fused into the data-gradient GEMM that produces
// dxh = d loss / d xhat (the LayerNorm affine is folded into the Linear weight):
//
//   out = res + rstd * (dxh - mean_c(dxh) - xhat * mean_c(dxh * xhat)),  xhat = (x - mean) * rstd
//
// (network_swinir.py:293,335 backward).  Needs the whole row in the block: one N
// block (N <= 64*WN), WM = 1.  A row's columns live in two waves (wn = 0, 1) and,
// inside a wave, in the 32 lanes of a half: per-lane partial sums over the lane's
// tiles, a halving butterfly over the 32 lanes (16 rows x 2 sums in 16 exchanges
// each instead of 80), then the two wave halves meet in LDS.
template <int WN>
__device__ __forceinline__ void nt_epilogue_lnbwd(const NtArgs& p, f32x16 (&acc)[1][WN], int lane, int wm,
                                                  int wn, int m0, int nvalid, float* red) {
  const int r = lane & 31;
  const int rbase = m0 + wm * 32;
  float rs[16];
  float xh[WN][16];
  {
    float mu[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int g = min(rbase + mfma_row(q, lane), p.M - 1);      // clamped rows are never stored
      const float2 st = *(const float2*)(p.ep_stats + 2 * (long)g);
      mu[q] = st.x; rs[q] = st.y;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = min((wn * WN + j) * 32 + r, nvalid - 1);    // clamped columns are zeroed below
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int g = min(rbase + mfma_row(q, lane), p.M - 1);
        xh[j][q] = p.R[(long)g * p.ldr + col];
      }
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const bool cok = (wn * WN + j) * 32 + r < nvalid;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[0][j][q] = cok ? acc[0][j][q] : 0.f;                  // padded columns hold garbage
        xh[j][q] = cok ? (xh[j][q] - mu[q]) * rs[q] : 0.f;
      }
    }
  }
  float a1[16], a2[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < WN; ++j) { s1 += acc[0][j][q]; s2 += acc[0][j][q] * xh[j][q]; }
    a1[q] = s1; a2[q] = s2;
  }
  const float t1 = half_reduce16(a1, lane), t2 = half_reduce16(a2, lane);
  const int qs = half_reduce16_row(lane);
  __syncthreads();                                  // staging buffers are dead from here on
  if (!(lane & 1)) *(float2*)(red + ((wn * 64) + wm * 32 + mfma_row(qs, lane)) * 2) = float2{t1, t2};
  // the residual gradient is fetched while the two wave halves meet in LDS
  float rres[WN][16];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = min((wn * WN + j) * 32 + r, nvalid - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int g = min(rbase + mfma_row(q, lane), p.M - 1);
      rres[j][q] = p.R2 ? p.R2[(long)g * p.ldr2 + col] : 0.f;
    }
  }
  __syncthreads();
  const float inv = 1.0f / (float)p.N;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int lr = wm * 32 + mfma_row(q, lane);
    const float2 p0 = *(const float2*)(red + lr * 2), p1 = *(const float2*)(red + (64 + lr) * 2);
    const float m1 = (p0.x + p1.x) * inv, m2 = (p0.y + p1.y) * inv;
    const bool rok = rbase + mfma_row(q, lane) < p.M;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = (wn * WN + j) * 32 + r;
      if (rok && col < nvalid)
        p.C[(long)(rbase + mfma_row(q, lane)) * p.ldc + col] =
            rres[j][q] + rs[q] * (acc[0][j][q] - m1 - xh[j][q] * m2);
    }
  }
}

// Wide epilogue (GEMM, WM = 1, epilogues 0-4): the accumulator layout gives a lane ONE
// column of 16 rows per tile, i.e. 4-byte global accesses -- 48 loads + 48 stores per
// lane for a 32 x 96 wave tile, and the store tail is issue bound (every block of a
// launch reaches it together).  Here the wave transposes its tile through LDS (the
// staging buffers are dead by now) and finishes in ROW-MAJOR float4 pieces: residual /
// gate loads and stores are 16 bytes per lane, a quarter of the instructions.
// LDS row pitch 32*WN + 8 floats: the two lane halves (rows 4 apart) land 32 banks
// apart, so the transposing ds_write_b32 is conflict free.
template <int WN, bool STATS>
__device__ __forceinline__ void nt_epilogue_wide_impl(const NtArgs& p, f32x16 (&acc)[1][WN], int lane, int wave,
                                                 int wm, int wn, int n0, int nvalid, int m0, float* lds) {
  constexpr int P = 32 * WN + 8;
  constexpr int F4 = 8 * WN;                   // float4 per tile row
  constexpr int NIT = 32 * F4 / 64;            // float4 per lane
  const int r = lane & 31;
  float* tile = lds + wave * (32 * P);
  __syncthreads();                             // all waves are done with the staging buffers
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = (wn * WN + j) * 32 + r;
    const float bv = (col < nvalid && p.bias) ? p.bias[n0 + col] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[mfma_row(q, lane) * P + j * 32 + r] = acc[0][j][q] + bv;
  }
  __builtin_amdgcn_wave_barrier();
  const bool needR = p.R != nullptr && p.epi >= 2;
  const bool per_row = p.rowscale && (p.rows_per_scale % 64 != 0);
  float blk_s = p.alpha;
  if (p.rowscale && !per_row) blk_s *= p.rowscale[m0 / p.rows_per_scale];
  f32x4 rv[NIT];
  long goff[NIT];                              // element offset row * ld is per matrix: keep (row, col)
  int grow[NIT], gcol[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / F4, c4 = idx - row * F4;
    const int col = wn * (32 * WN) + c4 * 4;
    const int g = m0 + wm * 32 + row;
    grow[it] = (g < p.M && col < nvalid) ? g : -1;      // nvalid % 4 == 0 (checked by the dispatcher)
    gcol[it] = n0 + col;
    goff[it] = (long)row * P + c4 * 4;
    rv[it] = (needR && grow[it] >= 0) ? *(const f32x4*)(p.R + (long)g * p.ldr + gcol[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    f32x4 v = *(const f32x4*)(tile + goff[it]);
    float s = blk_s;
    if (per_row && grow[it] >= 0) s *= p.rowscale[grow[it] / p.rows_per_scale];
    switch (p.epi) {
      case 1:
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        break;
      case 2:         // component-wise: no v_pk_fma_f32 (common.h)
        v.x = v.x * s + rv[it].x; v.y = v.y * s + rv[it].y; v.z = v.z * s + rv[it].z; v.w = v.w * s + rv[it].w;
        break;
      case 3: {
        f32x4 gl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {          // Phi / phi as in nt_epilogue (backward-grade erf)
          const float x = rv[it][e];
          const float z = fabsf(x) * 0.70710678118654752440f;
          const float e1 = __expf(-0.5f * x * x);
          const float t = __frcp_rn(1.0f + 0.3275911f * z);
          const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f +
                             t * (-1.453152027f + t * 1.061405429f))));
          const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e1, x));
          v[e] = v[e] * s * (cdf + x * 0.39894228040143267794f * e1);
          gl[e] = x * cdf;
        }
        if (p.aux && grow[it] >= 0) *(f32x4*)(p.aux + (long)grow[it] * p.ldaux + gcol[it]) = gl;
        break;
      }
      case 4:
        v.x = rv[it].x > 0.f ? v.x : 0.f; v.y = rv[it].y > 0.f ? v.y : 0.f;
        v.z = rv[it].z > 0.f ? v.z : 0.f; v.w = rv[it].w > 0.f ? v.w : 0.f;
        break;
      case 6:
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
        break;
      case 7:
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rv[it][e] > 0.f ? v[e] : v[e] * p.alpha;
        break;
      default:
        break;
    }
    if (grow[it] >= 0) *(f32x4*)(p.C + (long)grow[it] * p.ldc + gcol[it]) = v;
    if (STATS) *(f32x4*)(tile + goff[it]) = v;            // final values back into the wave's tile
  }
  if (STATS) {
    // LayerNorm statistics of the OUTPUT rows (the next LayerNorm's mean / rstd; one N block, so a
    // row lives in the two column waves of its row group).  The tile is row-major in LDS now: lane
    // (row = l & 31, half = l >> 5) sums its 48 columns, the halves meet by one shuffle, the two
    // column waves in LDS.  Two-pass like the reference: mean first, then squared deviations.
    float* red = lds + 4 * (32 * P);                       // [2][2][64]
    __builtin_amdgcn_wave_barrier();
    const int half = lane >> 5;
    const int cbase = wn * (32 * WN) + half * (16 * WN);
    f32x4 xv[4 * WN];
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4 * WN; ++k) {
      xv[k] = *(const f32x4*)(tile + r * P + half * (16 * WN) + 4 * k);
      if (cbase + 4 * k < nvalid) s1 += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);     // nvalid % 4 == 0
    }
    s1 += __shfl_xor(s1, 32, 64);
    if (half == 0) red[wn * 64 + wm * 32 + r] = s1;
    __syncthreads();
    const float mean = (red[wm * 32 + r] + red[64 + wm * 32 + r]) * (1.0f / (float)p.N);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4 * WN; ++k)
      if (cbase + 4 * k < nvalid) {
        const float d0 = xv[k].x - mean, d1 = xv[k].y - mean, d2 = xv[k].z - mean, d3 = xv[k].w - mean;
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) red[128 + wn * 64 + wm * 32 + r] = s2;
    __syncthreads();
    const int g = m0 + wm * 32 + r;
    if (wn == 0 && half == 0 && g < p.M) {
      const float var = (red[128 + wm * 32 + r] + red[192 + wm * 32 + r]) * (1.0f / (float)p.N);
      *(float2*)(p.stats_out + 2 * (long)g) = float2{mean, rsqrtf(var + 1e-5f)};
    }
  }
}

template <int WN>
__device__ __forceinline__ void nt_epilogue_wide(const NtArgs& p, f32x16 (&acc)[1][WN], int lane, int wave,
                                                 int wm, int wn, int n0, int nvalid, int m0, float* lds) {
  if (p.stats_out) nt_epilogue_wide_impl<WN, true>(p, acc, lane, wave, wm, wn, n0, nvalid, m0, lds);
  else nt_epilogue_wide_impl<WN, false>(p, acc, lane, wave, wm, wn, n0, nvalid, m0, lds);
}
