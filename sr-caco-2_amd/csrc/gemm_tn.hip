// TN contraction on the exact-f32 MFMA: weight gradients.
//
//   out[i][j] = sum_m pa(A)[m][i] * pb(B)[m'][j]
//
//   Linear : A = dY [M][NI], B = X [M][NJ]               -> dW [NI][NJ]
//   Conv   : A = dY [pixels][Cout], B = X NHWC, m' = pixel shifted by the tap
//            -> dWp [tap][Cout][Cin]
//
// The reduce dimension (tokens / pixels) is split into S slices; every block
// writes its partial tile into part[s][tap][NI][NJ] with plain stores and a
// small second kernel (reduce.hip) sums the slices -- bitwise reproducible,
// unlike float atomics.  Column sums of A (bias gradients) ride along: the
// A chunk is already in LDS.
//
// Replaces autograd's weight-gradient GEMMs for nn.Linear / nn.Conv2d
// (network_swinir.py:34-36,131-133,544; network_nlsn.py:38-41).
//
// LDS holds the chunk token-major ([32 tokens][192 cols]) exactly as it lies in
// HBM; a fragment is then one ds_read_b32 per MFMA with consecutive lanes on
// consecutive columns (conflict free); lane half h takes token 2s+h.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TK = 32;  // tokens per staged chunk

template <int WI, int WJ>
__device__ __forceinline__ void tn_body(const TnArgs& p, const int s, const int tile, const int tap) {
  constexpr int BI = 64 * WI, BJ = 64 * WJ;
  constexpr int A_IT = TK * BI / 4 / 256, B_IT = TK * BJ / 4 / 256;
  static_assert(TK * BI / 4 % 256 == 0 && TK * BJ / 4 % 256 == 0, "tile");
  __shared__ __attribute__((aligned(16))) float smem[TK * (BI + BJ)];
  float* As = smem;
  float* Bs = smem + TK * BI;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
  const int nbj = (p.NJ + p.j_tile - 1) / p.j_tile;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * p.i_tile, j0 = bj * p.j_tile;
  const int ivalid = min(p.i_tile, p.NI - i0), jvalid = min(p.j_tile, p.NJ - j0);
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const int dy = p.conv ? tap / 3 - 1 : 0, dx = p.conv ? tap % 3 - 1 : 0;

  f32x4 ra[A_IT], rb[B_IT];
  float rsa[A_IT];
  float2 rst[B_IT];               // {mean, rstd} of the B row (or a neutral pair)

  auto load = [&](int mc) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / (BI / 4), c4 = idx - row * (BI / 4);
      const int gm = mc + row, gc = c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const bool in = gm < m_end && gc < ivalid;
      if (in) v = ldg_f4(p.A + (long)gm * p.lda + i0 + gc);
      const float* sp = (in && p.a_rowscale) ? p.a_rowscale + gm / p.a_rowscale_rows : k_sr_neutral + 1;
      rsa[it] = ldg_f(sp);               // unconditional load (see k_sr_neutral)
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / (BJ / 4), c4 = idx - row * (BJ / 4);
      const int gm = mc + row, gc = c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const float* sp = k_sr_neutral;
      if (gm < m_end && gc < jvalid) {
        long src = gm;
        bool ok = true;
        if (p.conv) {
          const int x = gm % p.Wd, t = gm / p.Wd;
          const int y = t % p.H, b = t / p.H;
          const int yy = y + dy, xx = x + dx;
          ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
          src = ((long)b * p.H + yy) * p.Wd + xx;
        }
        if (ok) {
          v = ldg_f4(p.B + src * p.ldb + j0 + gc);
          if (p.b_mode == 1) sp = p.ln_stats + 2 * src;
        } else {
          sp = k_sr_neutral + 2;   // {0,0}: padded pixels stay exactly 0 under any prologue
        }
      }
      rst[it] = ldg_f2(sp);   // unconditional load (see k_sr_neutral)
      rb[it] = v;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int idx = tid + it * 256;
      f32x4 v = ra[it];
      v.x *= rsa[it]; v.y *= rsa[it]; v.z *= rsa[it]; v.w *= rsa[it];
      *(f32x4*)(As + idx * 4) = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / (BJ / 4);
      f32x4 v = rb[it];
      if (p.b_mode == 1) {
        const float mu = rst[it].x, rs = rst[it].y;
        v.x = (v.x - mu) * rs; v.y = (v.y - mu) * rs; v.z = (v.z - mu) * rs; v.w = (v.w - mu) * rs;
      } else if (p.b_mode == 2) {
        // rows past the slice end hold zeros and gelu(0) = 0
        v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
      }
      (void)row;
      *(f32x4*)(Bs + idx * 4) = v;
    }
  };

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  float colsum = 0.f;
  const bool do_colsum = p.part_colsum && bj == 0 && tap == 0;

  if (m_begin < m_end) load(m_begin);
  for (int mc = m_begin; mc < m_end; mc += TK) {
    __syncthreads();
    store();
    __syncthreads();
    if (mc + TK < m_end) load(mc + TK);
    if (do_colsum && tid < BI) {
      float cs = 0.f;
#pragma unroll 8
      for (int t = 0; t < TK; ++t) cs += As[t * BI + tid];
      colsum += cs;
    }
    // fragments are fetched one token pair ahead of the MFMAs that use them
    // (register double buffer), so LDS latency hides behind the matrix pipe
    float fa[2][WI], fb[2][WJ];
    const float* ar0 = As + h * BI + wi * WI * 32 + r;
    const float* br0 = Bs + h * BJ + wj * WJ * 32 + r;
#pragma unroll
    for (int i = 0; i < WI; ++i) fa[0][i] = ar0[i * 32];
#pragma unroll
    for (int j = 0; j < WJ; ++j) fb[0][j] = br0[j * 32];
#pragma unroll
    for (int sp = 0; sp < TK / 2; ++sp) {
      const int cur = sp & 1, nxt = cur ^ 1;
      if (sp + 1 < TK / 2) {
#pragma unroll
        for (int i = 0; i < WI; ++i) fa[nxt][i] = ar0[2 * (sp + 1) * BI + i * 32];
#pragma unroll
        for (int j = 0; j < WJ; ++j) fb[nxt][j] = br0[2 * (sp + 1) * BJ + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = mfma32(fa[cur][i], fb[cur][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float* out = p.part + ((long)(s * (p.conv ? 9 : 1) + tap) * p.NI) * p.NJ;
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int col = (wj * WJ + j) * 32 + r;
      if (col >= jvalid) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (wi * WI + i) * 32 + mfma_row(q, lane);
        if (row < ivalid) out[(long)(i0 + row) * p.NJ + j0 + col] = acc[i][j][q];
      }
    }
  if (do_colsum && tid < ivalid) p.part_colsum[(long)s * p.NI + i0 + tid] = colsum;
}

template <int WI, int WJ>
__global__ void __launch_bounds__(256, 2) k_tn(TnArgs p) {
  tn_body<WI, WJ>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several weight-gradient problems over the same token range in ONE launch
// (the four Linear layers of a Swin block): enough blocks to fill the chip at a
// moderate slice count, so the partial-sum traffic stays small.
struct TnGroup {
  TnArgs p[4];
  int tile_start[5];
  int n;
};
template <int WI, int WJ>
__global__ void __launch_bounds__(256, 2) k_tn_grouped(TnGroup g) {
  const int t = blockIdx.y;
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.n && t >= g.tile_start[i]) k = i;
  // select with constant indices (a runtime-indexed struct array would go to scratch)
  if (k == 0) tn_body<WI, WJ>(g.p[0], blockIdx.x, t - g.tile_start[0], 0);
  else if (k == 1) tn_body<WI, WJ>(g.p[1], blockIdx.x, t - g.tile_start[1], 0);
  else if (k == 2) tn_body<WI, WJ>(g.p[2], blockIdx.x, t - g.tile_start[2], 0);
  else tn_body<WI, WJ>(g.p[3], blockIdx.x, t - g.tile_start[3], 0);
}

int pick_tile(int n, int* w) {
  if (n % 180 == 0) { *w = 3; return 180; }
  if (n <= 64) { *w = 1; return 64; }
  if (n <= 128 || n % 128 == 0) { *w = 2; return 128; }
  *w = 3; return 192;
}

}  // namespace

int sr_tn_plan(int M, int NI, int NJ, int conv, int* S, long* part_floats) {
  return sr_tn_plan_t(M, NI, NJ, conv, 512, S, part_floats);
}

int sr_tn_plan_t(int M, int NI, int NJ, int conv, int target, int* S, long* part_floats) {
  int wi, wj;
  const int ti = pick_tile(NI, &wi), tj = pick_tile(NJ, &wj);
  const long tiles = (long)sr_cdiv(NI, ti) * sr_cdiv(NJ, tj) * (conv ? 9 : 1);
  // one block per CU and more (k_tn<3,3> holds 1 block/CU); at least 128 tokens each
  long s = target / tiles;               // f32 kernel: two whole rounds of 256 blocks; bx3: one
  const long smax = (M + 127) / 128;
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  *S = (int)s;
  *part_floats = s * (conv ? 9 : 1) * (long)NI * NJ;
  return 0;
}

int sr_tn_group_plan(int M, int ntiles, int* S) { return sr_tn_group_plan_t(M, ntiles, 512, S); }

int sr_tn_group_plan_t(int M, int ntiles, int dflt_target, int* S) {
  // one whole round of 512 blocks (2 blocks per CU, so one block's staging and
  // index math overlap the other's MFMAs); env SRHIP_TN_BLOCKS overrides
  const char* e = sr_getenv("SRHIP_TN_BLOCKS");
  const long target = e ? atol(e) : dflt_target;
  long s = target / ntiles;
  if (s < 1) s = 1;
  const long smax = (M + 127) / 128;
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  *S = (int)s;
  return 0;
}

int sr_tn_tiles(int NI, int NJ) {
  int wi, wj;
  return sr_cdiv(NI, pick_tile(NI, &wi)) * sr_cdiv(NJ, pick_tile(NJ, &wj));
}

int sr_gemm_tn_grouped(TnArgs* probs, int n, hipStream_t st) {
  SR_REQUIRE(n >= 1 && n <= 4, "gemm_tn_grouped: 1..4 problems (got %d)", n);
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  int wi = 1, wj = 1, tiles = 0;
  for (int k = 0; k < n; ++k) {
    TnArgs& p = probs[k];
    SR_REQUIRE(p.NI % 4 == 0 && p.NJ % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0,
               "gemm_tn_grouped: NI, NJ, lda, ldb must be multiples of 4");
    SR_REQUIRE(p.M == probs[0].M && p.S == probs[0].S && !p.conv,
               "gemm_tn_grouped: problems must share M and S");
    int a, b;
    p.i_tile = pick_tile(p.NI, &a);
    p.j_tile = pick_tile(p.NJ, &b);
    if (a > wi) wi = a;
    if (b > wj) wj = b;
    int rps = sr_cdiv(p.M, p.S);
    p.rows_per_slice = (rps + TK - 1) / TK * TK;
    g.tile_start[k] = tiles;
    tiles += sr_cdiv(p.NI, p.i_tile) * sr_cdiv(p.NJ, p.j_tile);
    g.p[k] = p;
  }
  g.tile_start[n] = tiles;
  // one kernel instantiation for the whole group: the largest wave tile; smaller
  // problems run with masked columns
  if (wi != wj) wi = wj = (wi > wj ? wi : wj);
  dim3 grid(probs[0].S, tiles, 1);
  if (wi == 1) hipLaunchKernelGGL((k_tn_grouped<1, 1>), grid, dim3(256), 0, st, g);
  else if (wi == 2) hipLaunchKernelGGL((k_tn_grouped<2, 2>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_tn_grouped<3, 3>), grid, dim3(256), 0, st, g);
  SR_LAUNCH_CHECK("k_tn_grouped");
  return 0;
}

int sr_gemm_tn(TnArgs& p, hipStream_t st) {
  SR_REQUIRE(p.NI % 4 == 0 && p.NJ % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0,
             "gemm_tn: NI, NJ, lda, ldb must be multiples of 4");
  SR_REQUIRE(p.M > 0 && p.S > 0, "gemm_tn: empty problem");
  int wi, wj;
  p.i_tile = pick_tile(p.NI, &wi);
  p.j_tile = pick_tile(p.NJ, &wj);
  int rps = sr_cdiv(p.M, p.S);
  rps = (rps + TK - 1) / TK * TK;
  p.rows_per_slice = rps;
  dim3 grid(p.S, sr_cdiv(p.NI, p.i_tile) * sr_cdiv(p.NJ, p.j_tile), p.conv ? 9 : 1);
#define SR_TN_CASE(WI_, WJ_) \
  if (wi == WI_ && wj == WJ_) { hipLaunchKernelGGL((k_tn<WI_, WJ_>), grid, dim3(256), 0, st, p); }
  SR_TN_CASE(1, 1) SR_TN_CASE(1, 2) SR_TN_CASE(1, 3)
  SR_TN_CASE(2, 1) SR_TN_CASE(2, 2) SR_TN_CASE(2, 3)
  SR_TN_CASE(3, 1) SR_TN_CASE(3, 2) SR_TN_CASE(3, 3)
#undef SR_TN_CASE
  SR_LAUNCH_CHECK("k_tn");
  return 0;
}
