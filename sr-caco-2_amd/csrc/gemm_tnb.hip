// TN contraction (weight gradients) with 3-way bf16 split operands on the bf16
// MFMA -- the f32-accurate scheme of gemm_ntb.hip applied to
//
//   out[i][j] = sum_m pa(A)[m][i] * pb(B)[m'][j]        (see gemm_tn.hip)
//
// The reduce index (tokens / pixels) is the MFMA k, so both operands must reach
// the matrix core column-major: lane (r, h) of a 32x32x16 MFMA supplies 8
// consecutive TOKENS of column r.  The transposition happens in the global load
// itself: a lane owns one column and loads it for 8 consecutive tokens (every
// load instruction reads 64 consecutive floats of one token row -- coalesced),
// splits the 8 values into bf16 (h, m, l) and writes one 16-byte unit per plane.
// Everything that depends on the token only (DropPath row scale, LayerNorm
// statistics, the tap-shifted source pixel of the conv) is wave-uniform and
// lives in SGPRs.
//
// LDS: per plane and column two 16-byte units (tokens 0-7, 8-15 of a 16-token
// chunk), unit (col, u) at slot 2*col + (u ^ ((col>>3)&1)): the 16 lanes of a
// ds_read/write_b128 phase (16 consecutive columns, same u) hit 16 distinct
// slots.  Two chunk buffers (one barrier per chunk): chunk c+1 is split and
// stored while chunk c feeds the MFMAs.
//
// Same slicing / partial-tile output / reducers as gemm_tn.hip.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TKB = 16;   // tokens per chunk = one k step of the 32x32x16 MFMA

__device__ const float k_tnb_zero_row[256] = {0.f};   // source row of tokens that contribute nothing

__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ int unit_slot(int col, int u) { return 2 * col + (u ^ ((col >> 3) & 1)); }

template <int W>
__device__ __forceinline__ void tnb_body(const TnArgs& p, const int s, const int tile, const int tap,
                                         unsigned char* smem) {
  constexpr int BC = 64 * W;                 // columns per operand tile
  constexpr int NU = 2 * 2 * BC;             // units per chunk (A and B, two token octets each)
  constexpr int IT = NU / 256;               // = W
  constexpr int PLANE = 2 * BC * 32;         // bytes per plane per chunk buffer (A cols then B cols)
  constexpr int BUF = 3 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
  const int nbj = (p.NJ + p.j_tile - 1) / p.j_tile;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * p.i_tile, j0 = bj * p.j_tile;
  const int ivalid = min(p.i_tile, p.NI - i0), jvalid = min(p.j_tile, p.NJ - j0);
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const int dy = p.conv ? tap / 3 - 1 : 0, dx = p.conv ? tap % 3 - 1 : 0;
  const bool do_colsum = p.part_colsum && bj == 0 && tap == 0;

  // unit (it): idx = tid + 256*it = u * (2*BC) + c, c < BC: A column c, else B column c - BC.
  // 2*BC and BC are multiples of 64, so operand and u are uniform per wave.
  float rv[IT][8];
  float cs[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) cs[it] = 0.f;
  // per-token data of the chunk in flight, held by lane (token & 15) and broadcast
  // with v_readlane: source row of each operand (-1: contributes zeros), DropPath
  // scale of the A row, LayerNorm statistics of the B row
  int t_rowA = -1, t_rowB = -1;
  float t_scale = 1.f;
  float2 t_stats = {0.f, 1.f};

  auto load = [&](int mc) {
    {
      const int gm = mc + (lane & 15);
      const bool in = gm < m_end;
      t_rowA = in ? gm : -1;
      int srow = gm;
      bool ok = in;
      if (p.conv) {
        const int x = gm % p.Wd, tq = gm / p.Wd;
        const int y = tq % p.H, b = tq / p.H;
        const int yy = y + dy, xx = x + dx;
        ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
        srow = (b * p.H + yy) * p.Wd + xx;
      }
      t_rowB = ok ? srow : -1;
      const float* sp = (in && p.a_rowscale) ? p.a_rowscale + gm / p.a_rowscale_rows : k_sr_neutral + 1;
      t_scale = ldg_f(sp);
      const float* tp = (ok && p.b_mode == 1) ? p.ln_stats + 2 * (long)srow : k_sr_neutral;
      t_stats = ldg_f2(tp);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int wbase = __builtin_amdgcn_readfirstlane((wave << 6) + 256 * it);   // idx of lane 0
      const int u = wbase / (2 * BC), cw = wbase - u * (2 * BC);
      const bool isB = cw >= BC;
      const int c = (cw - (isB ? BC : 0)) + lane;
      const float* P = isB ? p.B + j0 : p.A + i0;          // uniform
      const long ld = isB ? p.ldb : p.lda;
      const int cc = min(c, (isB ? jvalid : ivalid) - 1);
      const int rows = isB ? t_rowB : t_rowA;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = __builtin_amdgcn_readlane(rows, 8 * u + t);
        const float* base = row >= 0 ? P + (long)row * ld : k_tnb_zero_row;   // uniform
        rv[it][t] = ldg_f(base + cc);
      }
    }
  };

  auto store = [&](unsigned char* buf) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int wbase = __builtin_amdgcn_readfirstlane((wave << 6) + 256 * it);
      const int u = wbase / (2 * BC), cw = wbase - u * (2 * BC);
      const bool isB = cw >= BC;
      const int c = cw + lane;                 // column inside the [A | B] chunk row
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = rv[it][t];
      if (!isB) {
        if (p.a_rowscale) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            v[t] *= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_scale), 8 * u + t));
        }
        if (do_colsum) {
          float q = 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t) q += v[t];
          cs[it] += q;
        }
      } else if (p.b_mode == 1) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_stats.x), 8 * u + t));
          const float rs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_stats.y), 8 * u + t));
          v[t] = (v[t] - mu) * rs;              // zero-filled tokens carry {0, 1}
        }
      } else if (p.b_mode == 2) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = gelu_f(v[t]);        // gelu(0) = 0 for the zero fill
      }
      unsigned qh[4], qm[4], ql[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) split3_pair(v[2 * t], v[2 * t + 1], qh[t], qm[t], ql[t]);
      const u32x4 ph = {qh[0], qh[1], qh[2], qh[3]}, pm = {qm[0], qm[1], qm[2], qm[3]},
                  pl = {ql[0], ql[1], ql[2], ql[3]};
      unsigned char* dst = buf + unit_slot(c, u) * 16;
      *(u32x4*)(dst) = ph;
      *(u32x4*)(dst + PLANE) = pm;
      *(u32x4*)(dst + 2 * PLANE) = pl;
    }
  };

  f32x16 acc[W][W];
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  int a_off[W], b_off[W];
#pragma unroll
  for (int i = 0; i < W; ++i) a_off[i] = unit_slot((wi * W + i) * 32 + r, h) * 16;
#pragma unroll
  for (int j = 0; j < W; ++j) b_off[j] = unit_slot(BC + (wj * W + j) * 32 + r, h) * 16;

  const int nch = (m_end - m_begin + TKB - 1) / TKB;
  if (nch > 0) {
    load(m_begin);
    store(smem);
    if (nch > 1) load(m_begin + TKB);
  }
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    unsigned char* cur = smem + (c & 1) * BUF;
    if (c + 1 < nch) store(smem + ((c + 1) & 1) * BUF);
    if (c + 2 < nch) load(m_begin + (c + 2) * TKB);
    u32x4 fa[W][3];
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[i][pl] = *(const u32x4*)(cur + pl * PLANE + a_off[i]);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      u32x4 fb0 = *(const u32x4*)(cur + b_off[j]);
      u32x4 fb1 = *(const u32x4*)(cur + PLANE + b_off[j]);
      u32x4 fb2 = *(const u32x4*)(cur + 2 * PLANE + b_off[j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][1], fb1, acc[i][j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][0], fb2, acc[i][j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][2], fb0, acc[i][j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][0], fb1, acc[i][j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][1], fb0, acc[i][j]);
#pragma unroll
      for (int i = 0; i < W; ++i) acc[i][j] = mfma_bf(fa[i][0], fb0, acc[i][j]);
    }
    __syncthreads();
  }

  float* out = p.part + ((long)(s * (p.conv ? 9 : 1) + tap) * p.NI) * p.NJ;
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int col = (wj * W + j) * 32 + r;
      if (col >= jvalid) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (wi * W + i) * 32 + mfma_row(q, lane);
        if (row < ivalid) out[(long)(i0 + row) * p.NJ + j0 + col] = acc[i][j][q];
      }
    }

  if (do_colsum) {          // a column's two token octets live in different threads: meet in LDS
    float* red = (float*)smem;               // [2][BC]; the chunk buffers are dead now
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int wbase = __builtin_amdgcn_readfirstlane((wave << 6) + 256 * it);
      const int u = wbase / (2 * BC), cw = wbase - u * (2 * BC);
      if (cw < BC) red[u * BC + cw + lane] = cs[it];
    }
    __syncthreads();
    if (tid < ivalid) p.part_colsum[(long)s * p.NI + i0 + tid] = red[tid] + red[BC + tid];
  }
}

template <int W>
__global__ void __launch_bounds__(256, 2) k_tnb(TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  tnb_body<W>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

struct TnbGroup {
  TnArgs p[4];
  int tile_start[5];
  int n;
};
template <int W>
__global__ void __launch_bounds__(256, 2) k_tnb_grouped(TnbGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = blockIdx.y;
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < g.n && t >= g.tile_start[i]) k = i;
  if (k == 0) tnb_body<W>(g.p[0], blockIdx.x, t - g.tile_start[0], 0, smem);
  else if (k == 1) tnb_body<W>(g.p[1], blockIdx.x, t - g.tile_start[1], 0, smem);
  else if (k == 2) tnb_body<W>(g.p[2], blockIdx.x, t - g.tile_start[2], 0, smem);
  else tnb_body<W>(g.p[3], blockIdx.x, t - g.tile_start[3], 0, smem);
}

int pick_tile(int n, int* w) {
  if (n % 180 == 0) { *w = 3; return 180; }
  if (n <= 64) { *w = 1; return 64; }
  if (n <= 128 || n % 128 == 0) { *w = 2; return 128; }
  *w = 3; return 192;
}

template <typename K>
int reserve_lds(K kern, int bytes, const char* name) {
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return sr_fail(-5, "%s: cannot reserve %d B of LDS: %s", name, bytes, hipGetErrorString(e));
  return 0;
}

constexpr int lds_bytes(int w) { return 2 * 3 * 2 * 64 * w * 32; }

}  // namespace

int sr_gemm_tnb_grouped(TnArgs* probs, int n, hipStream_t st) {
  SR_REQUIRE(n >= 1 && n <= 4, "gemm_tn_grouped_bx3: 1..4 problems (got %d)", n);
  TnbGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  int w = 1, tiles = 0;
  for (int k = 0; k < n; ++k) {
    TnArgs& p = probs[k];
    SR_REQUIRE(p.M == probs[0].M && p.S == probs[0].S && !p.conv,
               "gemm_tn_grouped_bx3: problems must share M and S");
    int a, b;
    p.i_tile = pick_tile(p.NI, &a);
    p.j_tile = pick_tile(p.NJ, &b);
    if (a > w) w = a;
    if (b > w) w = b;
    const int rps = sr_cdiv(p.M, p.S);
    p.rows_per_slice = (rps + TKB - 1) / TKB * TKB;
    g.tile_start[k] = tiles;
    tiles += sr_cdiv(p.NI, p.i_tile) * sr_cdiv(p.NJ, p.j_tile);
    g.p[k] = p;
  }
  g.tile_start[n] = tiles;
  dim3 grid(probs[0].S, tiles, 1);
  static bool attr[4] = {false, false, false, false};
#define SR_TNB_G(W_)                                                                      \
  if (w == W_) {                                                                          \
    if (!attr[W_]) {                                                                      \
      if (int rc = reserve_lds(k_tnb_grouped<W_>, lds_bytes(W_), "k_tnb_grouped")) return rc; \
      attr[W_] = true;                                                                    \
    }                                                                                     \
    hipLaunchKernelGGL((k_tnb_grouped<W_>), grid, dim3(256), lds_bytes(W_), st, g);       \
  }
  SR_TNB_G(1) SR_TNB_G(2) SR_TNB_G(3)
#undef SR_TNB_G
  SR_LAUNCH_CHECK("k_tnb_grouped");
  return 0;
}

int sr_gemm_tnb(TnArgs& p, hipStream_t st) {
  SR_REQUIRE(p.M > 0 && p.S > 0, "gemm_tn_bx3: empty problem");
  int wi, wj;
  p.i_tile = pick_tile(p.NI, &wi);
  p.j_tile = pick_tile(p.NJ, &wj);
  const int w = wi > wj ? wi : wj;
  const int rps = sr_cdiv(p.M, p.S);
  p.rows_per_slice = (rps + TKB - 1) / TKB * TKB;
  dim3 grid(p.S, sr_cdiv(p.NI, p.i_tile) * sr_cdiv(p.NJ, p.j_tile), p.conv ? 9 : 1);
  static bool attr[4] = {false, false, false, false};
#define SR_TNB(W_)                                                                        \
  if (w == W_) {                                                                          \
    if (!attr[W_]) {                                                                      \
      if (int rc = reserve_lds(k_tnb<W_>, lds_bytes(W_), "k_tnb")) return rc;             \
      attr[W_] = true;                                                                    \
    }                                                                                     \
    hipLaunchKernelGGL((k_tnb<W_>), grid, dim3(256), lds_bytes(W_), st, p);               \
  }
  SR_TNB(1) SR_TNB(2) SR_TNB(3)
#undef SR_TNB
  SR_LAUNCH_CHECK("k_tnb");
  return 0;
}
