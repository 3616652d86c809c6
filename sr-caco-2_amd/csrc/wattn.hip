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
#include <stdlib.h>
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

// Staging of one [64 tokens][D] matrix (q, k, v or dO of one head) into this wave's
// LDS region: lane = token.  A lane reads its own row (D floats, 8-byte aligned, 94 %
// of a 128-byte line for D = 30) with float2 loads at immediate offsets and writes LDS
// row `lane` -- no per-load address arithmetic and no cross-lane token shuffles (the
// earlier coalesced mapping spent a third of the kernel's vector instructions on
// them), and the 30-dword row stride makes the ds_write_b64 conflict free.
template <int D>
__device__ __forceinline__ void wa_stage_load(float2 (&regs)[D / 2], const float* __restrict__ gsrc,
                                              long row_pitch, int mytok, int lane) {
  const float* rowp = gsrc + (long)mytok * row_pitch;
#pragma unroll
  for (int i = 0; i < D / 2; ++i) regs[i] = ldg_f2(rowp + 2 * i);
}
template <int D>
__device__ __forceinline__ void wa_stage_store(float* __restrict__ lds, const float2 (&regs)[D / 2], int lane) {
#pragma unroll
  for (int i = 0; i < D / 2; ++i) *(float2*)(lds + lane * D + 2 * i) = regs[i];
}
template <int D>
__device__ __forceinline__ void wa_stage(float* __restrict__ lds, const float* __restrict__ gsrc,
                                         long row_pitch, int mytok, int lane) {
  float2 regs[D / 2];
  wa_stage_load<D>(regs, gsrc, row_pitch, mytok, lane);
  wa_stage_store<D>(lds, regs, lane);
}

// Relative-position bias images in LANE ORDER.  The dense images are only ever read
// as MFMA accumulator tiles: lane l of tile (a, b) needs the 16 values of its
// accumulator registers.  Stored as img[head][a][b][lane][16] they are four
// global_load_dwordx4 per tile at immediate offsets instead of sixteen scattered dwords.
//   imgT (fwd, bwd_q; tile (kb, qb)): value of key = mfma_row(q,l) + 32*kb, query = (l&31) + 32*qb
//   imgN (bwd_kv;     tile (qb, kb)): value of query = mfma_row(q,l) + 32*qb, key = (l&31) + 32*kb
// both = table[rpi(query, key)][head], rpi(q,k) = (qy-ky+7)*15 + (qx-kx+7)
// (network_swinir.py:116-128,156-162).
__device__ __forceinline__ int wa_img_index(int a, int b, int lane, int q) {
  return (((a * 2 + b) * 64 + lane) * 16) + q;
}
// the bias-GRADIENT image is accumulated with float atomics: register-major order
// [a][b][q][lane], so that one wave instruction adds to 256 contiguous bytes (lane-major
// would touch 64 cache lines per instruction: measured 8x slower)
__device__ __forceinline__ int wa_dimg_index(int a, int b, int lane, int q) {
  return (((a * 2 + b) * 16 + q) * 64) + lane;
}
__device__ __forceinline__ int wa_rpi(int query, int key) {
  return ((query >> 3) - (key >> 3) + 7) * 15 + ((query & 7) - (key & 7) + 7);
}

// Logical block index for XCD locality (sr_xcd_block, common.h): the waves that share cache
// lines -- the heads of one window: a head's 120-byte slice of a 2160-byte token row straddles
// the lines of its neighbours -- have consecutive logical indices and so meet in one L2
// instead of fetching the shared lines into several.
__device__ __forceinline__ int wa_block(int remap) {
  return remap ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
}

template <int D>
__global__ void __launch_bounds__(256, 3) k_wattn_fwd(const float* __restrict__ qkv, float* __restrict__ out,
                                                   const float* __restrict__ biasT, long total, int H,
                                                   int W, int C, int heads, int shift, float scale, int remap) {
  constexpr int HD = D / 2;
  __shared__ __attribute__((aligned(16))) float smem[4][64 * D];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  const long gid = wa_block(remap) * 4L + wv;
  if (gid >= total) return;               // no block-level barrier below
  const WaGeom g = wa_decode(gid, heads, W / 8, H / 8, shift);
  const int C3 = 3 * C;
  const int mytok = wa_token(g, lane, H, W, shift);
  // ONE 64 x D staging buffer per wave (7.7 KB at D = 30), used for Q, then K, then V: the MFMA
  // fragments of Q and K live in registers, so the buffer is free again as soon as they are read.
  // With two buffers a block held 61 KB -> 8 waves per CU and the 3072 (window, head) waves of the
  // benchmark ran as 1.5 rounds; with one buffer and <= 168 registers 12 waves per CU = one round.
  float* Ks = smem[wv];
  const float* hb = qkv + g.head * D;
  float2 vreg[D / 2], kreg[D / 2];
  wa_stage_load<D>(vreg, hb, C3, mytok, lane);            // Q
  wa_stage_load<D>(kreg, hb + C, C3, mytok, lane);        // K
  wa_stage_store<D>(Ks, vreg, lane);
  __builtin_amdgcn_wave_barrier();
  float qf[2][HD], kf[2][HD];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int t = 0; t < HD; ++t) qf[blk][t] = Ks[(r + 32 * blk) * D + h * HD + t];
  __builtin_amdgcn_wave_barrier();
  wa_stage_store<D>(Ks, kreg, lane);
  __builtin_amdgcn_wave_barrier();
  wa_stage_load<D>(vreg, hb + 2 * C, C3, mytok, lane);    // V on its way while S^T is computed
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int t = 0; t < HD; ++t) kf[blk][t] = Ks[(r + 32 * blk) * D + h * HD + t];
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
  wa_stage_store<D>(Ks, vreg, lane);
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
        float s = T[kb][qb][q] * scale + bt[wa_img_index(kb, qb, lane, q)];
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
        const float e = __expf(T[kb][qb][q] - mx);
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

// ---------------------------------------------------------------------------
// Backward as two kernels, each wave owning HALF a (window, head):
//   bwd_q : (window, head, query block qb) -> softmax stats (to a workspace), dQ of
//           its 32 queries, d(bias) tiles (kb, qb) as global float atomics
//   bwd_kv: (window, head, key block kb)   -> dK, dV of its 32 keys (reads the stats)
// A wave needs one full and one half [tokens][D] matrix in LDS at a time (11.5 KB)
// and 2 x 2 accumulator tiles less than a whole-window wave: 12 waves per CU instead
// of 8, and the 6144 waves of the README shape fill the chip in two even rounds
// (whole-window waves were 3 per SIMD with 2 resident: the second round ran half
// empty, and SQ counters showed 57 % of the wave cycles stalled on issue).
// ---------------------------------------------------------------------------
// half matrix: rows 32*blk .. 32*blk+31 of a head matrix; lane (r, h) moves float2
// 8h .. 8h+7 of row r
template <int D>
__device__ __forceinline__ void wa_stage_half_load(float2 (&regs)[8], const float* __restrict__ gsrc,
                                                   long row_pitch, int tok, int h) {
  const float* rowp = gsrc + (long)tok * row_pitch;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = min(h * 8 + i, D / 2 - 1);           // D/2 <= 16 float2 per row
    regs[i] = ldg_f2(rowp + 2 * f);
  }
}
template <int D>
__device__ __forceinline__ void wa_stage_half_store(float* __restrict__ lds, const float2 (&regs)[8], int r, int h) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = h * 8 + i;
    if (f < D / 2) *(float2*)(lds + r * D + 2 * f) = regs[i];
  }
}

constexpr int WA_NW = 2;   // windows per bwd_q wave

template <int D>
__global__ void __launch_bounds__(256, 3) k_wattn_bwd_q(
    const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
    const float* __restrict__ biasT, float* __restrict__ dbias_part, float* __restrict__ stats, int nwin,
    int H, int W, int C, int heads, int shift, float scale, int remap) {
  constexpr int HD = D / 2;
  static_assert(D <= 32, "head dim");
  constexpr int SM = 4 * (64 + 32) * D > 4096 ? 4 * (64 + 32) * D : 4096;   // >= two d(bias) tile pairs
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  float* As = smem + wv * ((64 + 32) * D);     // full matrix: K, V, K again
  float* Bs = As + 64 * D;                     // half matrix: Q, dO rows of this query block
  const int lb = wa_block(remap);
  const int head = lb % heads;
  const int item = (lb / heads) * 4 + wv;
  const int qb = item & 1;
  // WA_NW consecutive windows per wave, one after the other: their d(bias) tiles are
  // summed in registers first
  f32x16 dsacc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) dsacc[a][q] = 0.f;
  for (int rep = 0; rep < WA_NW; ++rep) {
  const int widx = (item >> 1) * WA_NW + rep;
  if (widx < nwin) {
    __builtin_amdgcn_wave_barrier();
    const int nWx = W / 8, nWy = H / 8;
    WaGeom g;
    g.head = head;
    g.wx = widx % nWx;
    g.wy = (widx / nWx) % nWy;
    g.b = widx / (nWx * nWy);
    g.last_row = shift > 0 && g.wy == nWy - 1;
    g.last_col = shift > 0 && g.wx == nWx - 1;
    const int C3 = 3 * C;
    const int mytok = wa_token(g, lane, H, W, shift);           // token of key / staging row `lane`
    const int qtok = wa_token(g, r + 32 * qb, H, W, shift);     // token of this lane's query
    const float lane_mask = (g.last_col && h != ((lane >> 2) & 1)) ? -100.f : 0.f;
    const int fo0 = r * D + h * HD, fo1 = (r + 32) * D + h * HD;
    f32x16 T[2], G[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) { T[a][q] = 0.f; G[a][q] = 0.f; }
    // S^T = K.Q^T  (V and dO are fetched into registers meanwhile)
    float2 ra[D / 2], rb[8];
    wa_stage<D>(As, qkv + C + head * D, C3, mytok, lane);
    wa_stage_half_load<D>(rb, qkv + head * D, C3, qtok, h);
    wa_stage_half_store<D>(Bs, rb, r, h);
    __builtin_amdgcn_wave_barrier();
    wa_stage_load<D>(ra, qkv + 2 * C + head * D, C3, mytok, lane);
    wa_stage_half_load<D>(rb, dout + head * D, C, qtok, h);
#pragma unroll 3
    for (int t = 0; t < HD; ++t) {
      const float k0 = As[fo0 + t], k1 = As[fo1 + t], q0 = Bs[fo0 + t];
      T[0] = mfma32(k0, q0, T[0]);
      T[1] = mfma32(k1, q0, T[1]);
    }
    // dP^T = V.dO^T  (K is fetched again meanwhile: column pattern for dQ)
    __builtin_amdgcn_wave_barrier();
    wa_stage_store<D>(As, ra, lane);
    wa_stage_half_store<D>(Bs, rb, r, h);
    __builtin_amdgcn_wave_barrier();
    wa_stage_load<D>(ra, qkv + C + head * D, C3, mytok, lane);
#pragma unroll 3
    for (int t = 0; t < HD; ++t) {
      const float v0 = As[fo0 + t], v1 = As[fo1 + t], g0 = Bs[fo0 + t];
      G[0] = mfma32(v0, g0, G[0]);
      G[1] = mfma32(v1, g0, G[1]);
    }
    __builtin_amdgcn_wave_barrier();
    wa_stage_store<D>(As, ra, lane);
    __builtin_amdgcn_wave_barrier();
    const float* bt = biasT + (long)head * 4096;
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float tile_mask = (g.last_row && kb != qb) ? -100.f : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float s = T[kb][q] * scale + bt[wa_img_index(kb, qb, lane, q)];
        s += tile_mask; s += lane_mask;
        T[kb][q] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float e = __expf(T[kb][q] - mx);
        T[kb][q] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float dl = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        T[kb][q] *= inv;                     // P^T
        dl += T[kb][q] * G[kb][q];
      }
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0) {                            // softmax statistics of query r + 32*qb
      float* sp = stats + ((long)qtok * heads + head) * 3;
      sp[0] = mx; sp[1] = inv; sp[2] = dl;
    }
    f32x16 dQ;
#pragma unroll
    for (int q = 0; q < 16; ++q) dQ[q] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int key = mfma_row(q, lane) + 32 * kb;
        const float ds = T[kb][q] * (G[kb][q] - dl);
        dsacc[kb][q] += ds;
        const float kc = r < D ? As[key * D + r] : 0.f;
        dQ = mfma32(ds, kc, dQ);
      }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int tok = wa_token(g, mfma_row(q, lane) + 32 * qb, H, W, shift);
      if (r < D) dqkv[(long)tok * C3 + head * D + r] = dQ[q] * scale;
    }
  }
  }
  // bias gradient: NO atomics.  The block's two window groups (waves wv and wv + 2 own the same
  // query block) meet in LDS and the block stores ONE partial d(bias) tile pair per query block --
  // plain 256-byte-per-instruction stores into part[block-in-head][head][4096] -- which the tail
  // blocks of the key pass sum in fp64 (k_wattn_bwd_kv).  Deterministic, and the sum over all
  // windows no longer depends on the arrival order of 12.6 M float atomics per launch
  // (MI355X_MICROARCH.md "Global float atomics": 1.3 TB/s chip-wide, 5x below plain stores).
  if (dbias_part) {
    __syncthreads();                                   // every wave is done with its staging region
    if (wv >= 2) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; ++q) smem[(wv - 2) * 2048 + (kb * 16 + q) * 64 + lane] = dsacc[kb][q];
    }
    __syncthreads();
    if (wv < 2) {
      float* dst = dbias_part + ((long)(lb / heads) * heads + head) * 4096;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          dst[wa_dimg_index(kb, qb, lane, q)] = dsacc[kb][q] + smem[wv * 2048 + (kb * 16 + q) * 64 + lane];
    }
  }
}

template <int D>
__global__ void __launch_bounds__(256, 3) k_wattn_bwd_kv(
    const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
    const float* __restrict__ biasN, const float* __restrict__ stats, long total, int H, int W, int C,
    int heads, int shift, float scale, int remap, int nmain, const float* __restrict__ dbias_part,
    int nparts, float* __restrict__ dbiasT) {
  constexpr int HD = D / 2;
  __shared__ __attribute__((aligned(16))) float smem[4 * ((64 + 32) * D + 192)];
  const int ntail = (int)gridDim.x - nmain;
  if ((int)blockIdx.x < ntail) {
    // the FIRST blocks of the grid (they start at once and overlap the main blocks): d(bias) image =
    // sum of the query pass's partial tiles (complete: that launch precedes this one on the
    // stream), one thread per image element, eight loads in flight, fp64 accumulation
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int img = heads * 4096;
    if (e < img) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const float* src = dbias_part + e;
      int p = 0;
      for (; p + 8 <= nparts; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ldg_f(src + (long)(p + u) * img);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
      }
      for (; p < nparts; ++p) a[0] += (double)ldg_f(src + (long)p * img);
      dbiasT[e] = (float)(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    }
    return;
  }
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  const int mb = (int)blockIdx.x - ntail;
  const long item = (remap ? sr_xcd_block(mb, nmain) : mb) * 4L + wv;       // (window, head) x key block
  if (item >= 2 * total) return;                // no block-level barrier below
  const int kb = (int)(item & 1);
  const WaGeom g = wa_decode(item >> 1, heads, W / 8, H / 8, shift);
  const int head = g.head;
  const int C3 = 3 * C;
  const int mytok = wa_token(g, lane, H, W, shift);             // token of query / staging row `lane`
  const int ktok = wa_token(g, r + 32 * kb, H, W, shift);       // token of this lane's key
  const float lane_mask = (g.last_col && h != ((lane >> 2) & 1)) ? -100.f : 0.f;
  float* As = smem + wv * ((64 + 32) * D + 192);   // full matrix: Q, dO, Q again
  float* Bs = As + 64 * D;                         // half matrix: K, V rows of this key block
  float* st = Bs + 32 * D;                         // [3][64]: max, 1/sum, delta of query = position
  const int fo0 = r * D + h * HD, fo1 = (r + 32) * D + h * HD;
  f32x16 S[2], G[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) { S[a][q] = 0.f; G[a][q] = 0.f; }
  // S = Q.K^T (rows = queries)
  float2 ra[D / 2], rb[8];
  wa_stage<D>(As, qkv + head * D, C3, mytok, lane);
  wa_stage_half_load<D>(rb, qkv + C + head * D, C3, ktok, h);
  wa_stage_half_store<D>(Bs, rb, r, h);
  {
    const float* sp = stats + ((long)mytok * heads + head) * 3;
    st[lane] = sp[0]; st[64 + lane] = sp[1]; st[128 + lane] = sp[2];
  }
  __builtin_amdgcn_wave_barrier();
  wa_stage_load<D>(ra, dout + head * D, C, mytok, lane);          // dO and V on their way during S
  wa_stage_half_load<D>(rb, qkv + 2 * C + head * D, C3, ktok, h);
#pragma unroll 3
  for (int t = 0; t < HD; ++t) {
    const float q0 = As[fo0 + t], q1 = As[fo1 + t], k0 = Bs[fo0 + t];
    S[0] = mfma32(q0, k0, S[0]);
    S[1] = mfma32(q1, k0, S[1]);
  }
  // dP = dO.V^T  (Q is fetched again meanwhile: column pattern for dK)
  __builtin_amdgcn_wave_barrier();
  wa_stage_store<D>(As, ra, lane);
  wa_stage_half_store<D>(Bs, rb, r, h);
  __builtin_amdgcn_wave_barrier();
  wa_stage_load<D>(ra, qkv + head * D, C3, mytok, lane);
#pragma unroll 3
  for (int t = 0; t < HD; ++t) {
    const float g0 = As[fo0 + t], g1 = As[fo1 + t], v0 = Bs[fo0 + t];
    G[0] = mfma32(g0, v0, G[0]);
    G[1] = mfma32(g1, v0, G[1]);
  }
  // P and dS in place (S -> P, G -> dS), then dV = P^T.dO with dO still in LDS
  const float* bn = biasN + (long)head * 4096;
  f32x16 dK, dV;
#pragma unroll
  for (int q = 0; q < 16; ++q) { dK[q] = 0.f; dV[q] = 0.f; }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float tile_mask = (g.last_row && kb != qb) ? -100.f : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int qry = mfma_row(q, lane) + 32 * qb;       // window position of the query
      const float mx = st[qry], inv = st[64 + qry], dl = st[128 + qry];
      float s = S[qb][q] * scale + bn[wa_img_index(qb, kb, lane, q)];
      s += tile_mask; s += lane_mask;                    // lane_mask is symmetric in (query, key)
      const float pv = __expf(s - mx) * inv;
      G[qb][q] = pv * (G[qb][q] - dl);
      const float gc = r < D ? As[qry * D + r] : 0.f;
      dV = mfma32(pv, gc, dV);
    }
  }
  // dK = dS^T.Q with Q back in LDS
  __builtin_amdgcn_wave_barrier();
  wa_stage_store<D>(As, ra, lane);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int qry = mfma_row(q, lane) + 32 * qb;
      const float qc = r < D ? As[qry * D + r] : 0.f;
      dK = mfma32(G[qb][q], qc, dK);
    }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int tok = wa_token(g, mfma_row(q, lane) + 32 * kb, H, W, shift);
    if (r < D) {
      dqkv[(long)tok * C3 + C + head * D + r] = dK[q] * scale;
      dqkv[(long)tok * C3 + 2 * C + head * D + r] = dV[q];
    }
  }
}

// lane-ordered bias images from the (225, heads) table (see wa_img_index)
__global__ void k_bias_expand(const float* __restrict__ table, float* __restrict__ biasT,
                              float* __restrict__ biasN, int heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * 4096) return;
  const int hd = i / 4096, e = i & 4095;
  const int q = e & 15, lane = (e >> 4) & 63, b = (e >> 10) & 1, a = e >> 11;
  const int row = mfma_row(q, lane), col = lane & 31;
  // imgT: tile (kb = a, qb = b): key = row + 32a, query = col + 32b
  biasT[i] = table[wa_rpi(col + 32 * b, row + 32 * a) * heads + hd];
  // imgN: tile (qb = a, kb = b): query = row + 32a, key = col + 32b
  biasN[i] = table[wa_rpi(row + 32 * a, col + 32 * b) * heads + hd];
}
// dtable[idx][h] = sum over (query,key) with rpi == idx of the bias-gradient image
// (wa_dimg_index order); one wave per table entry, one lane per key position
struct BiasGradBatch {          // up to 8 attention blocks (images heads*4096 floats apart in one buffer)
  float* dtable[8];
};
__global__ void k_bias_grad(const float* __restrict__ dbiasT, float* __restrict__ dtable, int heads,
                            BiasGradBatch batch, long img_stride) {
  if (batch.dtable[0]) {        // batched form: blockIdx.y = attention block
    dbiasT += blockIdx.y * img_stride;
    dtable = batch.dtable[blockIdx.y];
  }
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= 225 * heads) return;
  const int hd = i % heads, idx = i / heads;
  const int dy = idx / 15 - 7, dx = idx % 15 - 7;   // query - key
  const int ky = lane >> 3, kx = lane & 7;
  const int qy = ky + dy, qx = kx + dx;
  float a = 0.f;
  if (qy >= 0 && qy < 8 && qx >= 0 && qx < 8) {
    const int key = lane, query = qy * 8 + qx;
    const int kb = key >> 5, rowin = key & 31, qb = query >> 5;
    const int h = (rowin >> 2) & 1, q = (rowin & 3) + 4 * (rowin >> 3);     // mfma_row(q, 32h + c) == rowin
    a = dbiasT[(long)hd * 4096 + wa_dimg_index(kb, qb, (query & 31) + 32 * h, q)];
  }
  const double ad = wave_sum_d((double)a);
  if (lane == 0) dtable[i] = (float)ad;
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
  BiasGradBatch none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(k_bias_grad, dim3(sr_cdiv(225 * heads, 4)), dim3(256), 0, (hipStream_t)stream,
                     dbiasT, dtable, heads, none, 0L);
  SR_LAUNCH_CHECK("bias_grad");
  return 0;
}

int srhip_bias_grad_batched(const float* dbiasT, long image_stride, float* const* dtables, int nblocks, int heads,
                            void* stream) {
  SR_REQUIRE(nblocks >= 1 && nblocks <= 8, "bias_grad_batched: 1..8 blocks per launch (got %d)", nblocks);
  BiasGradBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < nblocks; ++i) {
    SR_REQUIRE(dtables[i] != nullptr, "bias_grad_batched: table %d missing", i);
    b.dtable[i] = dtables[i];
  }
  hipLaunchKernelGGL(k_bias_grad, dim3(sr_cdiv(225 * heads, 4), nblocks), dim3(256), 0, (hipStream_t)stream,
                     dbiasT, (float*)nullptr, heads, b, image_stride);
  SR_LAUNCH_CHECK("bias_grad_batched");
  return 0;
}

static int wa_remap() {            // SRHIP_WA_XCD=0: plain block order (A/B switch)
  static const int on = [] { const char* e = sr_getenv("SRHIP_WA_XCD"); return !(e && e[0] == '0'); }();
  return on;
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
  // (a variant with one wave per 32-query half -- 3 whole rounds of waves instead of
  // 1.5 -- measured 35 % slower: K and V are then staged twice)
#define SR_WA(D_) \
  if (D == D_) hipLaunchKernelGGL((k_wattn_fwd<D_>), grid, blk, 0, st, qkv, out, biasT, total, H, W, C, heads, shift, scale, wa_remap());
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("window_attention_fwd");
  return 0;
}

static int wa_bwd_q_parts(int nwin) {     // query-pass blocks per head = partial d(bias) tile sets per head
  return sr_cdiv(2 * sr_cdiv(nwin, WA_NW), 4);
}

long srhip_window_attention_bwd_ws(int B, int H, int W, int heads) {
  // floats: softmax max, 1/sum, delta per (token, head) + the query pass's partial d(bias) tiles
  const int nwin = B * (H / 8) * (W / 8);
  return 3L * B * H * W * heads + (long)wa_bwd_q_parts(nwin) * heads * 4096;
}

// dbiasT (may be NULL) is overwritten with the bias-gradient image.
int srhip_window_attention_bwd(const float* qkv, const float* dout, float* dqkv, const float* biasT,
                               const float* biasN, float* dbiasT, float* workspace, int B, int H, int W,
                               int C, int heads, int shift, void* stream) {
  int rc = wattn_check(B, H, W, C, heads, shift);
  if (rc) return rc;
  SR_REQUIRE(workspace != nullptr, "window_attention_bwd: workspace required");
  const int D = C / heads;
  const int nwin = B * (H / 8) * (W / 8);
  const long total = (long)nwin * heads;
  const float scale = 1.0f / sqrtf((float)D);
  hipStream_t st = (hipStream_t)stream;
  // half a (window, head) per wave; bwd_q waves take WA_NW windows each
  const int nparts = wa_bwd_q_parts(nwin), nmain = sr_cdiv(2 * total, 4);
  const int ntail = dbiasT ? sr_cdiv(heads * 4096, 256) : 0;
  dim3 blk(256), gq(heads * nparts), gkv(nmain + ntail);
  float* part = dbiasT ? workspace + 3L * B * H * W * heads : nullptr;
#define SR_WA(D_) \
  if (D == D_) { \
    hipLaunchKernelGGL((k_wattn_bwd_q<D_>), gq, blk, 0, st, qkv, dout, dqkv, biasT, part, workspace, nwin, H, W, C, heads, shift, scale, wa_remap()); \
    hipLaunchKernelGGL((k_wattn_bwd_kv<D_>), gkv, blk, 0, st, qkv, dout, dqkv, biasN, workspace, total, H, W, C, heads, shift, scale, wa_remap(), nmain, part, nparts, dbiasT); \
  }
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("window_attention_bwd");
  return 0;
}

}  // extern "C"
