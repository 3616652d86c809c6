// Window / shifted-window multi-head self-attention core on the f32 MFMA.
//
// Replaces WindowAttention.forward's q@k^T, +bias, +mask, softmax, attn@v and
// the roll / window_partition / window_reverse copies around it
// (dlib/models/network_swinir.py:48-80,153-176,297-331) for window 8x8.
//
// Inputs stay in token order: qkv [B*H*W][3*C] as produced by the qkv Linear;
// the cyclic shift and the window grouping are pure address math here.  One
// wave owns one (window, head): S^T = K.Q^T is computed with the QUERY on the
// MFMA lane, so a softmax row lives in 32 registers of a lane plus its partner
// lane (lane^32): no LDS, one cross-lane exchange.  P^T in accumulator layout
// is directly the A operand of the P.V product (its reduce index, the key, is
// the accumulator row).  The shifted-window mask collapses to two bits per
// lane: with shift = 4 the region of a token inside the last window row /
// column depends only on (py>=4) / (px>=4), which are the tile index and the
// lane bits of the MFMA layout.
#include "common.h"
#include "kernels.h"

namespace {

struct WaGeom {
  int head, b, wy, wx;
  bool last_row, last_col;
};

__device__ __forceinline__ WaGeom wa_decode(long gid, int heads, int nWx, int nWy, int shift) {
  WaGeom g;
  g.head = (int)(gid % heads);
  long win = gid / heads;
  g.wx = (int)(win % nWx); win /= nWx;
  g.wy = (int)(win % nWy);
  g.b = (int)(win / nWy);
  g.last_row = shift > 0 && g.wy == nWy - 1;
  g.last_col = shift > 0 && g.wx == nWx - 1;
  return g;
}
// token index of window-local position pos (0..63) under the cyclic shift
__device__ __forceinline__ int wa_token(const WaGeom& g, int pos, int H, int W, int shift) {
  int y = g.wy * 8 + (pos >> 3) + shift, x = g.wx * 8 + (pos & 7) + shift;
  if (y >= H) y -= H;
  if (x >= W) x -= W;
  return (g.b * H + y) * W + x;
}

// Coalesced staging of one [64 tokens][D] matrix (q, k, v or dO of one head)
// into this wave's LDS region: consecutive lanes take consecutive float2 of a
// token row, so an instruction touches ~5 cache lines instead of 64.
template <int D>
__device__ __forceinline__ void wa_stage(float* __restrict__ lds, const float* __restrict__ gsrc,
                                         long row_pitch, int mytok, int lane) {
  constexpr int F2 = D / 2;
#pragma unroll
  for (int i = 0; i < F2; ++i) {
    const int idx = i * 64 + lane;
    const int tok = idx / F2, c2 = idx - tok * F2;
    const int t = __shfl(mytok, tok, 64);
    const float2 v = *(const float2*)(gsrc + (long)t * row_pitch + 2 * c2);
    *(float2*)(lds + tok * D + 2 * c2) = v;
  }
}

template <int D>
__global__ void __launch_bounds__(256) k_wattn_fwd(const float* __restrict__ qkv, float* __restrict__ out,
                                                   const float* __restrict__ biasT, long total, int H,
                                                   int W, int C, int heads, int shift, float scale) {
  constexpr int HD = D / 2;
  __shared__ __attribute__((aligned(16))) float smem[4][2][64 * D];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  const long gid = blockIdx.x * 4L + wv;
  if (gid >= total) return;               // no block-level barrier below
  const WaGeom g = wa_decode(gid, heads, W / 8, H / 8, shift);
  const int C3 = 3 * C;
  const int mytok = wa_token(g, lane, H, W, shift);
  float* Ks = smem[wv][0];
  float* Qs = smem[wv][1];
  const float* hb = qkv + g.head * D;
  wa_stage<D>(Qs, hb, C3, mytok, lane);
  wa_stage<D>(Ks, hb + C, C3, mytok, lane);
  __builtin_amdgcn_wave_barrier();

  float qf[2][HD], kf[2][HD];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int t = 0; t < HD; ++t) {
      qf[blk][t] = Qs[(r + 32 * blk) * D + h * HD + t];
      kf[blk][t] = Ks[(r + 32 * blk) * D + h * HD + t];
    }
  f32x16 T[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) T[a][b][q] = 0.f;
#pragma unroll
  for (int t = 0; t < HD; ++t)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) T[kb][qb] = mfma32(kf[kb][t], qf[qb][t], T[kb][qb]);

  // V replaces K in LDS (the K fragments are in registers by now)
  __builtin_amdgcn_wave_barrier();
  wa_stage<D>(Ks, hb + 2 * C, C3, mytok, lane);
  __builtin_amdgcn_wave_barrier();
  // V operand of P.V: lane = head-dim index, one value per (key block, reg)
  float vc[2][16];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      vc[kb][q] = r < D ? Ks[(mfma_row(q, lane) + 32 * kb) * D + r] : 0.f;

  const float lane_mask = (g.last_col && h != ((lane >> 2) & 1)) ? -100.f : 0.f;
  const float* bt = biasT + (long)g.head * 4096;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float tile_mask = (g.last_row && kb != qb) ? -100.f : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int key = mfma_row(q, lane) + 32 * kb;
        float s = T[kb][qb][q] * scale + bt[key * 64 + r + 32 * qb];
        s += tile_mask; s += lane_mask;
        T[kb][qb][q] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float e = expf(T[kb][qb][q] - mx);
        T[kb][qb][q] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f32x16 O;
#pragma unroll
    for (int q = 0; q < 16; ++q) O[q] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) O = mfma32(T[kb][qb][q] * inv, vc[kb][q], O);
#pragma unroll
    for (int q = 0; q < 16; ++q) {   // shuffle with every lane active, then guard the store
      const int tok = __shfl(mytok, mfma_row(q, lane) + 32 * qb, 64);
      if (r < D) out[(long)tok * C + g.head * D + r] = O[q];
    }
  }
}

// Backward: recompute P in both orientations (query-on-lane for dQ and the
// bias gradient, key-on-lane for dK and dV) so that every product reduces
// over an accumulator-row index and no 64x64 tile is transposed.
template <int D>
__global__ void __launch_bounds__(256) k_wattn_bwd(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                   float* __restrict__ dqkv, const float* __restrict__ biasT,
                                                   const float* __restrict__ biasN, float* __restrict__ dbiasT,
                                                   long total, int H, int W, int C, int heads, int shift,
                                                   float scale) {
  constexpr int HD = D / 2;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const long gid = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (gid >= total) return;
  const WaGeom g = wa_decode(gid, heads, W / 8, H / 8, shift);
  const int C3 = 3 * C;
  const int mytok = wa_token(g, lane, H, W, shift);
  const float lane_mask = (g.last_col && h != ((lane >> 2) & 1)) ? -100.f : 0.f;

  // row-pattern fragments (row = window position r+32*blk, this lane half's 15 dims)
  float qf[2][HD], kf[2][HD], vf[2][HD], gf[2][HD];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int tok = __shfl(mytok, r + 32 * blk, 64);
    const float* base = qkv + (long)tok * C3 + g.head * D + h * HD;
    const float* gb = dout + (long)tok * C + g.head * D + h * HD;
#pragma unroll
    for (int t = 0; t < HD; ++t) {
      qf[blk][t] = base[t]; kf[blk][t] = base[C + t]; vf[blk][t] = base[2 * C + t];
      gf[blk][t] = gb[t];
    }
  }
  float mrow[2], lrow[2], drow[2];   // per query (r+32*qb): max, 1/sum, delta

  // ---------------- pass 1: query on the lane ----------------
  {
    f32x16 T[2][2], G[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) { T[a][b][q] = 0.f; G[a][b][q] = 0.f; }
#pragma unroll
    for (int t = 0; t < HD; ++t)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          T[kb][qb] = mfma32(kf[kb][t], qf[qb][t], T[kb][qb]);   // S^T
          G[kb][qb] = mfma32(vf[kb][t], gf[qb][t], G[kb][qb]);   // dP^T = V.dO^T
        }
    const float* bt = biasT + (long)g.head * 4096;
    float* dbt = dbiasT + (long)g.head * 4096;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = -3.0e38f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float tile_mask = (g.last_row && kb != qb) ? -100.f : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int key = mfma_row(q, lane) + 32 * kb;
          float s = T[kb][qb][q] * scale + bt[key * 64 + r + 32 * qb];
          s += tile_mask; s += lane_mask;
          T[kb][qb][q] = s;
          mx = fmaxf(mx, s);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float e = expf(T[kb][qb][q] - mx);
          T[kb][qb][q] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      float dl = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          T[kb][qb][q] *= inv;                     // P^T
          dl += T[kb][qb][q] * G[kb][qb][q];
        }
      dl += __shfl_xor(dl, 32, 64);
      mrow[qb] = mx; lrow[qb] = inv; drow[qb] = dl;
      // dS^T (w.r.t. the biased, scaled logits), bias gradient, dQ
      f32x16 dQ;
#pragma unroll
      for (int q = 0; q < 16; ++q) dQ[q] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int key = mfma_row(q, lane) + 32 * kb;
          const float ds = T[kb][qb][q] * (G[kb][qb][q] - dl);
          atomicAdd(dbt + key * 64 + r + 32 * qb, ds);
          const int tok = __shfl(mytok, key, 64);
          const float kc = r < D ? qkv[(long)tok * C3 + C + g.head * D + r] : 0.f;
          dQ = mfma32(ds, kc, dQ);
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int tok = __shfl(mytok, mfma_row(q, lane) + 32 * qb, 64);
        if (r < D) dqkv[(long)tok * C3 + g.head * D + r] = dQ[q] * scale;
      }
    }
  }

  // ---------------- pass 2: key on the lane ----------------
  {
    f32x16 S[2][2], G[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) { S[a][b][q] = 0.f; G[a][b][q] = 0.f; }
#pragma unroll
    for (int t = 0; t < HD; ++t)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          S[qb][kb] = mfma32(qf[qb][t], kf[kb][t], S[qb][kb]);   // S   (rows = queries)
          G[qb][kb] = mfma32(gf[qb][t], vf[kb][t], G[qb][kb]);   // dP = dO.V^T
        }
    const float* bn = biasN + (long)g.head * 4096;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 dK, dV;
#pragma unroll
      for (int q = 0; q < 16; ++q) { dK[q] = 0.f; dV[q] = 0.f; }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float tile_mask = (g.last_row && kb != qb) ? -100.f : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int qry = mfma_row(q, lane);       // query index inside block qb
          const float mx = __shfl(mrow[qb], qry, 64);
          const float inv = __shfl(lrow[qb], qry, 64);
          const float dl = __shfl(drow[qb], qry, 64);
          float s = S[qb][kb][q] * scale + bn[(qry + 32 * qb) * 64 + r + 32 * kb];
          s += tile_mask; s += lane_mask;          // lane_mask is symmetric in (query,key)
          const float pv = expf(s - mx) * inv;
          const float ds = pv * (G[qb][kb][q] - dl);
          const int tok = __shfl(mytok, qry + 32 * qb, 64);
          const float qc = r < D ? qkv[(long)tok * C3 + g.head * D + r] : 0.f;
          const float gc = r < D ? dout[(long)tok * C + g.head * D + r] : 0.f;
          dV = mfma32(pv, gc, dV);
          dK = mfma32(ds, qc, dK);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int tok = __shfl(mytok, mfma_row(q, lane) + 32 * kb, 64);
        if (r < D) {
          dqkv[(long)tok * C3 + C + g.head * D + r] = dK[q] * scale;
          dqkv[(long)tok * C3 + 2 * C + g.head * D + r] = dV[q];
        }
      }
    }
  }
}

// dense bias images from the (225, heads) table:
//   biasT[h][key][query] = biasN[h][query][key] = table[rpi(query,key)][h]
//   rpi(q,k) = (qy-ky+7)*15 + (qx-kx+7)      (network_swinir.py:116-128,156-162)
__global__ void k_bias_expand(const float* __restrict__ table, float* __restrict__ biasT,
                              float* __restrict__ biasN, int heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * 4096) return;
  const int hd = i / 4096, a = (i >> 6) & 63, b = i & 63;   // element [hd][a][b]
  // biasN: a = query, b = key
  const int idxN = ((a >> 3) - (b >> 3) + 7) * 15 + ((a & 7) - (b & 7) + 7);
  biasN[i] = table[idxN * heads + hd];
  // biasT: a = key, b = query
  const int idxT = ((b >> 3) - (a >> 3) + 7) * 15 + ((b & 7) - (a & 7) + 7);
  biasT[i] = table[idxT * heads + hd];
}
// dtable[idx][h] = sum over (query,key) with rpi == idx of dbiasT[h][key][query]
__global__ void k_bias_grad(const float* __restrict__ dbiasT, float* __restrict__ dtable, int heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 225 * heads) return;
  const int hd = i % heads, idx = i / heads;
  const int dy = idx / 15 - 7, dx = idx % 15 - 7;   // query - key
  float a = 0.f;
  for (int ky = 0; ky < 8; ++ky) {
    const int qy = ky + dy;
    if (qy < 0 || qy > 7) continue;
    for (int kx = 0; kx < 8; ++kx) {
      const int qx = kx + dx;
      if (qx < 0 || qx > 7) continue;
      a += dbiasT[(long)hd * 4096 + (ky * 8 + kx) * 64 + qy * 8 + qx];
    }
  }
  dtable[i] = a;
}

}  // namespace

extern "C" {

int srhip_bias_expand(const float* table, float* biasT, float* biasN, int heads, void* stream) {
  hipLaunchKernelGGL(k_bias_expand, dim3(sr_cdiv(heads * 4096, 256)), dim3(256), 0,
                     (hipStream_t)stream, table, biasT, biasN, heads);
  SR_LAUNCH_CHECK("bias_expand");
  return 0;
}

int srhip_bias_grad(const float* dbiasT, float* dtable, int heads, void* stream) {
  hipLaunchKernelGGL(k_bias_grad, dim3(sr_cdiv(225 * heads, 256)), dim3(256), 0, (hipStream_t)stream,
                     dbiasT, dtable, heads);
  SR_LAUNCH_CHECK("bias_grad");
  return 0;
}

static int wattn_check(int B, int H, int W, int C, int heads, int shift) {
  SR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
             "window_attention: H, W must be positive multiples of the 8x8 window (H=%d W=%d)", H, W);
  SR_REQUIRE(heads > 0 && C % heads == 0, "window_attention: C %% heads != 0");
  SR_REQUIRE(shift == 0 || shift == 4, "window_attention: shift must be 0 or 4 (got %d)", shift);
  SR_REQUIRE(shift == 0 || (H > 8 && W > 8), "window_attention: shifted windows need H, W > 8");
  const int D = C / heads;
  SR_REQUIRE(D == 30 || D == 10 || D == 16 || D == 32, "window_attention: head dim %d not built", D);
  return 0;
}

int srhip_window_attention_fwd(const float* qkv, float* out, const float* biasT, int B, int H, int W,
                               int C, int heads, int shift, void* stream) {
  int rc = wattn_check(B, H, W, C, heads, shift);
  if (rc) return rc;
  const int D = C / heads;
  const long total = (long)B * (H / 8) * (W / 8) * heads;
  const float scale = 1.0f / sqrtf((float)D);
  dim3 grid(sr_cdiv(total, 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define SR_WA(D_) \
  if (D == D_) hipLaunchKernelGGL((k_wattn_fwd<D_>), grid, blk, 0, st, qkv, out, biasT, total, H, W, C, heads, shift, scale);
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("window_attention_fwd");
  return 0;
}

// dbiasT must be zero on entry (accumulated with atomics).
// dbiasT must be zero on entry (accumulated with atomics).
int srhip_window_attention_bwd(const float* qkv, const float* dout, float* dqkv, const float* biasT,
                               const float* biasN, float* dbiasT, int B, int H, int W, int C, int heads,
                               int shift, void* stream) {
  int rc = wattn_check(B, H, W, C, heads, shift);
  if (rc) return rc;
  const int D = C / heads;
  const long total = (long)B * (H / 8) * (W / 8) * heads;
  const float scale = 1.0f / sqrtf((float)D);
  dim3 grid(sr_cdiv(total, 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define SR_WA(D_) \
  if (D == D_) hipLaunchKernelGGL((k_wattn_bwd<D_>), grid, blk, 0, st, qkv, dout, dqkv, biasT, biasN, dbiasT, total, H, W, C, heads, shift, scale);
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("window_attention_bwd");
  return 0;
}

}  // extern "C"
