// TN contraction (weight gradients) with 3-way bf16 split operands on the bf16
// MFMA -- the f32-accurate scheme of gemm_ntb.hip applied to
//
//   out[i][j] = sum_m pa(A)[m][i] * pb(B)[m'][j]        (see gemm_tn.hip)
//
// The reduce index (tokens / pixels) is the MFMA k, so both operands must reach
// the matrix core column-major: lane (r, h) of a 32x32x16 MFMA supplies 8
// consecutive TOKENS of column r.  A producer lane owns W adjacent columns (the tile
// is 64*W columns wide: all 64 lanes work; W = 3 -> global_load_dwordx3, a wave reads
// 768 contiguous bytes of a token row per instruction) and loads them for 2 x 8
// consecutive tokens; the 8 x W register block IS the transpose: column j of it is
// one 16-byte LDS unit per bf16 plane after the split.  Everything that depends on
// the token only (DropPath row scale, LayerNorm statistics, the tap-shifted source
// pixel of the conv) is computed by lane (token & 15) and broadcast with v_readlane.
// (First version: 4 columns per lane on 48 of the 64 lanes -- a third more vector
// instructions per wave for the same tile: producers 4350 -> 3700 cycles per chunk.)
//
// LDS: per plane and column four 16-byte units (token octets 0-3 of a 32-token
// chunk), swizzled (unit_slot) so that BOTH access patterns are bank-conflict free:
// the consumer's ds_read_b128 (16 consecutive columns, same octet) and the producer's
// ds_write_b128 (lanes W columns apart).  Two chunk buffers, one barrier per chunk.
//
// Roles: a block is 8 waves = 4 CONSUMER waves (2x2, each W x W MFMA tiles of 32x32;
// nothing but ds_read + MFMA) and 4 PRODUCER waves (global loads, prologue, split, LDS
// stores for the next chunk).  One block per CU puts one consumer and one producer on
// every SIMD, so the matrix core and the vector ALU / memory pipes work at the same
// time; with symmetric waves the three phases ran back to back (measured: loads 55 us
// + split 50 us + MFMA 64 us = the 175 us of the launch).
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

constexpr int TKB = 32;   // tokens per chunk = two k steps of the 32x32x16 MFMA

__device__ const float k_tnb_zero_row[256] = {0.f};   // source row of tokens that contribute nothing

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tn_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// x = h + l in fp16 (round to nearest; the residual is exact in f32), two values per call (v_cvt_pk_f16_f32)
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
  const tn_f16x2 hv = __builtin_convertvector(f32x2{x0, x1}, tn_f16x2);
  const float r0 = x0 - (float)hv.x, r1 = x1 - (float)hv.y;
  const tn_f16x2 lv = __builtin_convertvector(f32x2{r0, r1}, tn_f16x2);
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}

// A column holds four 16-byte units per plane (token octets 0..3 of a 32-token chunk);
// unit (col, u) sits at slot 4*P(col) + (u ^ g(col)).  A 16-lane phase of ds_*_b128 is
// conflict free when its 16 units differ in (P & 3, u ^ g), so that pair must be a
// bijection of the column bits that vary inside a phase -- for the consumer's reads 16
// consecutive columns, for the producer's writes 16 columns W apart (a lane owns W
// adjacent columns).  W odd: (P & 3, g) = col & 15 (W is coprime with 16).  W = 2:
// (col ^ (col >> 4)) & 15 (b0 fixed, b1..b4 vary).  Both verified exhaustively.
template <int W>
__device__ __forceinline__ int unit_slot(int col, int u) {
  if (W == 2) {
    const int x = col ^ (col >> 4);
    return 4 * ((col & ~3) | (x & 3)) + (u ^ ((x >> 2) & 3));
  }
  return 4 * col + (u ^ ((col >> 2) & 3));
}

// W adjacent f32 columns of one token row as one register vector (4-byte aligned for W = 3)
typedef float tnb_f32x3 __attribute__((ext_vector_type(3)));
typedef tnb_f32x3 __attribute__((aligned(4))) tnb_f32x3_u;
template <int W> struct ColVec;
template <> struct ColVec<1> {
  typedef float T;
  static __device__ __forceinline__ T ldg(const float* p) { return ldg_f(p); }
  static __device__ __forceinline__ T ldg(const float* base, unsigned off) {
    return *(sr_gptr_f)((const char*)base + off);
  }
  static __device__ __forceinline__ float at(T v, int) { return v; }
};
template <> struct ColVec<2> {
  typedef sr_f32x2 T;
  static __device__ __forceinline__ T ldg(const float* p) { return *(sr_gptr_f2)p; }
  static __device__ __forceinline__ T ldg(const float* base, unsigned off) {
    return *(sr_gptr_f2)((const char*)base + off);
  }
  static __device__ __forceinline__ float at(T v, int j) { return v[j]; }
};
template <> struct ColVec<3> {
  typedef tnb_f32x3 T;
  static __device__ __forceinline__ T ldg(const float* p) {
    return *(const __attribute__((address_space(1))) tnb_f32x3_u*)p;
  }
  // uniform base (SGPR pair) + 32-bit per-lane byte offset: the global_load saddr form, no 64-bit
  // vector add per load
  static __device__ __forceinline__ T ldg(const float* base, unsigned off) {
    return *(const __attribute__((address_space(1))) tnb_f32x3_u*)((const char*)base + off);
  }
  static __device__ __forceinline__ float at(T v, int j) { return v[j]; }
};

template <int W, int DBG = 0>     // DBG (timing experiments, env SRHIP_TN_DBG): 1 no MFMA, 2 no producer work
__device__ __forceinline__ void tnb_body(const TnArgs& p, const int s, const int tile, const int tap,
                                         unsigned char* smem) {
  constexpr int BC = 64 * W;                 // columns per operand tile
  constexpr int PLANE = 2 * BC * 64;         // bytes per plane per chunk buffer (A cols then B cols, 4 units each)
  constexpr int BUF = 3 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;                // index inside the role
  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
  const int nbj = (p.NJ + p.j_tile - 1) / p.j_tile;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * p.i_tile, j0 = bj * p.j_tile;
  const int ivalid = min(p.i_tile, p.NI - i0), jvalid = min(p.j_tile, p.NJ - j0);
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const int dy = p.conv ? tap / 3 - 1 : 0, dx = p.conv ? tap % 3 - 1 : 0;
  const bool do_colsum = p.part_colsum && bj == 0 && tap == 0;

  // producer wave pw = wave: operand (pw & 1: 0 = A, 1 = B), token half hh = pw >> 1
  // (octets 2*hh and 2*hh+1 of the chunk); lane owns the W adjacent columns W*lane ..
  // (64 lanes x W = the whole operand tile: every lane works)
  struct Stage {            // one chunk in flight in the producer's registers
    typename ColVec<W>::T rv[2][8];   // [octet][token] x W columns
    float scale;            // A: DropPath scale of token (lane & 15) of the half
    float2 stats;           // B: LayerNorm statistics of token (lane & 15)
  };
  Stage sg0, sg1;
  const bool isB = wave & 1;
  const int hh = wave >> 1;
  // clamped: a lane past the valid width re-reads the last W valid columns (its tile columns
  // are never written out); a lane that straddles it is handled in store()
  const int opvalid = isB ? jvalid : ivalid;
  const int colq = max(min(W * lane, opvalid - W), 0);
  const int ps_f = p.NI >> 2;
  unsigned colb = (unsigned)colq * 4u;         // byte offset of this lane's columns in a token row
  if (!isB && p.ps) {
    const int ig = i0 + colq, sp = ig / ps_f, cc = ig - sp * ps_f;
    colb = (unsigned)((((sp >> 1) * 2 * p.Wd + (sp & 1)) * (int)p.lda + cc) * 4);
  }
  const int dsh = W * lane - colq;             // > 0: this lane's load was shifted left
  const bool ragged = opvalid % W != 0;        // uniform: some lane straddles the valid width
  const float* const pA = p.A;
  const float* const pB = p.B;
  const long ldA = p.lda, ldB = p.ldb;
  // conv + PixelShuffle(2) gradient (p.ps): kernel column i = sp*F + c is channel c of sub-pixel sp of the image
  // [batch][2H][2Wd][F], F = NI / 4.  A lane's W adjacent columns share one sub-pixel (F % W == 0, checked by the
  // dispatcher): the token decides the pixel (2y, 2x), the lane's columns a constant byte offset to (i, j) and c
  const float* const opP = isB ? pB + j0 : (p.ps ? pA : pA + i0);    // this wave's operand (uniform)
  const long opLd = isB ? ldB : ldA;
  float cs[W];
#pragma unroll
  for (int j = 0; j < W; ++j) cs[j] = 0.f;

  auto load = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int gm = mc + 16 * hh + (lane & 15);   // token of this lane's per-token data
    const float* P = opP;                                  // uniform
    const long ld = opLd;
    if (!p.conv && mc + TKB <= m_end) {          // interior chunk of a Linear problem (uniform): no bookkeeping
      if (!isB) sg.scale = ldg_f(p.a_rowscale ? p.a_rowscale + gm / p.a_rowscale_rows : k_sr_neutral + 1);
      else sg.stats = ldg_f2(p.b_mode == 1 ? p.ln_stats + 2 * (long)gm : k_sr_neutral);
      const float* q = P + (long)(mc + 16 * hh) * ld;       // uniform, advanced per token
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(q, colb);
        q += ld;
      }
      return;
    }
    const bool in = gm < m_end;
    int srow = gm;
    bool ok = in;
    if (isB && p.conv) {
      const int x = gm % p.Wd, tq = gm / p.Wd;
      const int y = tq % p.H, b = tq / p.H;
      const int yy = y + dy, xx = x + dx;
      ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
      srow = (b * p.H + yy) * p.Wd + xx;
    }
    if (!isB && p.ps) {      // token (b, y, x) of the low-resolution grid -> pixel (2y + i, 2x + j) of the shuffled image
      const int x = gm % p.Wd, tq = gm / p.Wd;
      const int y = tq % p.H, b = tq / p.H;
      srow = (b * 2 * p.H + 2 * y) * 2 * p.Wd + 2 * x;
    }
    const int t_row = ok ? srow : -1;
    if (!isB) {
      const float* sp = (in && p.a_rowscale) ? p.a_rowscale + gm / p.a_rowscale_rows : k_sr_neutral + 1;
      sg.scale = ldg_f(sp);
    } else {
      const float* tp = (ok && p.b_mode == 1) ? p.ln_stats + 2 * (long)srow : k_sr_neutral;
      sg.stats = ldg_f2(tp);
    }
    // all 16 tokens present, source rows consecutive (conv interior)
    const int row0 = __builtin_amdgcn_readlane(t_row, 0);
    const int step = (!isB && p.ps) ? 2 : 1;            // shuffled gradient image: consecutive tokens are two pixels apart
    const bool dense = __all(t_row == row0 + step * (lane & 15) && row0 >= 0);
    if (dense) {
      const float* q = P + (long)row0 * ld;                 // uniform, advanced per token
      const long adv = step * ld;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(q, colb);
        q += adv;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = __builtin_amdgcn_readlane(t_row, t);
        const float* base = row >= 0 ? P + (long)row * ld : k_tnb_zero_row;   // uniform
        sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(base, row >= 0 ? colb : 0u);
      }
    }
  };

  auto bcast = [&](float v, int t) __attribute__((always_inline)) -> float {          // per-token scalar of token t of the half
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
  };
  auto store = [&](unsigned char* buf, const Stage& sg) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {               // the two token octets of this wave's half
      // (token 2t, 2t+1) pairs of one column: the form the split wants (packed f32 math).
      // One uniform branch per octet selects the prologue; the loops inside are straight-line.
      sr_f32x2 x[4][W];
      if (!ragged) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j)
            x[t][j] = sr_f32x2{ColVec<W>::at(sg.rv[o][2 * t], j), ColVec<W>::at(sg.rv[o][2 * t + 1], j)};
      } else {                                  // (uniform) the lane that straddles the valid width loaded
#pragma unroll                                  // shifted left by dsh columns: move its columns back in place
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j) {
            float e0 = ColVec<W>::at(sg.rv[o][2 * t], W - 1), e1 = ColVec<W>::at(sg.rv[o][2 * t + 1], W - 1);
#pragma unroll
            for (int d = W - 2; d >= 0; --d)
              if (j + d < W && dsh == d) {
                e0 = ColVec<W>::at(sg.rv[o][2 * t], j + d);
                e1 = ColVec<W>::at(sg.rv[o][2 * t + 1], j + d);
              }
            x[t][j] = sr_f32x2{e0, e1};
          }
      }
      if (!isB) {
        if (p.a_rowscale) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float k0 = bcast(sg.scale, 8 * o + 2 * t), k1 = bcast(sg.scale, 8 * o + 2 * t + 1);
#pragma unroll
            for (int j = 0; j < W; ++j) { x[t][j].x *= k0; x[t][j].y *= k1; }      // scalar on purpose (see common.h)
          }
        }
        if (do_colsum) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < W; ++j) cs[j] += x[t][j].x + x[t][j].y;
        }
      } else if (p.b_mode == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float mu0 = bcast(sg.stats.x, 8 * o + 2 * t), mu1 = bcast(sg.stats.x, 8 * o + 2 * t + 1);
          const float rs0 = bcast(sg.stats.y, 8 * o + 2 * t), rs1 = bcast(sg.stats.y, 8 * o + 2 * t + 1);
#pragma unroll
          for (int j = 0; j < W; ++j) {             // zero-filled tokens carry {0, 1}
            x[t][j].x = (x[t][j].x - mu0) * rs0;
            x[t][j].y = (x[t][j].y - mu1) * rs1;
          }
        }
      } else if (p.b_mode == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j) {           // gelu(0) = 0 for the zero fill
            x[t][j].x = gelu_f(x[t][j].x); x[t][j].y = gelu_f(x[t][j].y);
          }
      }
#pragma unroll
      for (int j = 0; j < W; ++j) {             // column j of the 8 x W block = one unit per plane
        unsigned qh[4], qm[4], ql[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) split3_pair(x[t][j].x, x[t][j].y, qh[t], qm[t], ql[t]);
        const u32x4 ph = {qh[0], qh[1], qh[2], qh[3]}, pm = {qm[0], qm[1], qm[2], qm[3]},
                    pl = {ql[0], ql[1], ql[2], ql[3]};
        unsigned char* dst = buf + unit_slot<W>((isB ? BC : 0) + W * lane + j, 2 * hh + o) * 16;
        *(u32x4*)(dst) = ph;
        *(u32x4*)(dst + PLANE) = pm;
        *(u32x4*)(dst + 2 * PLANE) = pl;
      }
    }
  };

  auto touch = [&](const Stage& sg) __attribute__((always_inline)) {   // DBG 4: the column maxima of a stage, thrown away
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int j = 0; j < W; ++j) mx = fmaxf(mx, fabsf(ColVec<W>::at(sg.rv[t >> 3][t & 7], j)));
    asm volatile("" ::"v"(mx));
  };

  f32x16 acc[W][W];
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  int a_off[W][2], b_off[W][2];      // [tile][k step]: lane half h takes octet 2*step + h
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[i][ks] = unit_slot<W>((wi * W + i) * 32 + r, 2 * ks + h) * 16;
#pragma unroll
  for (int j = 0; j < W; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[j][ks] = unit_slot<W>(BC + (wj * W + j) * 32 + r, 2 * ks + h) * 16;

  // chunk count rounded up to even: the producer loop below is straight-line code per
  // pair of chunks (no conditional stage updates -- those made the compiler wait for
  // loads still in flight).  Chunks past the slice end load the zero row.
  const int nch = ((m_end - m_begin + 2 * TKB - 1) / (2 * TKB)) * 2;
  // both roles execute exactly nch + 1 barriers
  if (producer) {
    __builtin_amdgcn_s_setprio(2);                    // staging waves first: they are the critical path of a chunk
    // two chunks in flight: chunk c+1 is split and stored from one register stage
    // while the loads of chunk c+3 fill it again (c+2 sits in the other stage), so a
    // load has two chunk periods to land
    load(m_begin, sg0);
    load(m_begin + TKB, sg1);
    store(smem, sg0);
    load(m_begin + 2 * TKB, sg0);
    __syncthreads();
    long t_store = 0, t_load = 0, t_bar = 0;      // DBG 3: s_memtime stamps of the producer phases
    for (int c = 0; c < nch; c += 2) {
      const long s0 = DBG == 3 ? (long)__builtin_amdgcn_s_memtime() : 0;
      if (DBG != 2) store(smem + BUF, sg1);                 // chunk c+1
      const long s1 = DBG == 3 ? (long)__builtin_amdgcn_s_memtime() : 0;
      if (DBG != 2) load(m_begin + (c + 3) * TKB, sg1);
      if (DBG == 4) touch(sg0);                  // experiment: chunk c+2 must have landed one half-iteration early
      const long s2 = DBG == 3 ? (long)__builtin_amdgcn_s_memtime() : 0;
      __syncthreads();
      const long s3 = DBG == 3 ? (long)__builtin_amdgcn_s_memtime() : 0;
      if (DBG != 2) {
        store(smem, sg0);                       // chunk c+2
        load(m_begin + (c + 4) * TKB, sg0);
      }
      if (DBG == 4) touch(sg1);
      __syncthreads();
      if (DBG == 3) { t_store += s1 - s0; t_load += s2 - s1; t_bar += s3 - s2; }
    }
    if (DBG == 3 && blockIdx.x == 0 && tile == 0 && lane == 0) {   // cycles per chunk, by producer wave
      float* o = p.part + wave * 4;
      o[0] = (float)t_store / (nch / 2); o[1] = (float)t_load / (nch / 2); o[2] = (float)t_bar / (nch / 2);
      o[3] = (float)nch;
    }
  } else {
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const unsigned char* cur = smem + (c & 1) * BUF;
      if (DBG != 1) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 fa[W][3];
#pragma unroll
          for (int i = 0; i < W; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[i][pl] = *(const u32x4*)(cur + pl * PLANE + a_off[i][ks]);
#pragma unroll
          for (int j = 0; j < W; ++j) {
            u32x4 fb0 = *(const u32x4*)(cur + b_off[j][ks]);
            u32x4 fb1 = *(const u32x4*)(cur + PLANE + b_off[j][ks]);
            u32x4 fb2 = *(const u32x4*)(cur + 2 * PLANE + b_off[j][ks]);
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
        }
      }
      __syncthreads();
    }

  float* out = p.part + ((long)(s * (p.conv ? 9 : 1) + tap) * p.NI) * p.NJ;
  if (DBG == 3) return;     // stamp build: keep the stamps in part[]
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int col = (wj * W + j) * 32 + r;
      if (col >= jvalid) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (wi * W + i) * 32 + mfma_row(q, lane);
        if (row < ivalid) {
          const int io = i0 + row;               // p.ps: kernel row sp*F + c is torch channel c*4 + sp
          out[(long)(p.ps ? (io % ps_f) * 4 + io / ps_f : io) * p.NJ + j0 + col] = acc[i][j][q];
        }
      }
    }
  }

  if (do_colsum) {          // a column's two token octets live in different threads: meet in LDS
    float* red = (float*)smem;               // [2][BC]; the chunk buffers are dead now
    if (producer && !isB) {
#pragma unroll
      for (int j = 0; j < W; ++j) red[hh * BC + W * lane + j] = cs[j];
    }
    __syncthreads();
    if (tid < ivalid) {
      const int io = i0 + tid;
      p.part_colsum[(long)s * p.NI + (p.ps ? (io % ps_f) * 4 + io / ps_f : io)] = red[tid] + red[BC + tid];
    }
  }
}

// ---------------------------------------------------------------------------
// Linear problems (conv == 0) on TWO fp16 planes and three products: the scheme of tnb_body3<.., true> in the general
// tile.  A column's scale must be ONE value for its 32-token chunk, and in tnb_body the two token halves of a column are
// staged by different waves -- so the producer roles are re-cut here: wave = operand x COLUMN half (32 W columns), lane =
// column W-tuple (lane & 31) x token half (lane >> 5).  The two halves of a column sit in lanes l and l ^ 32 of one
// wave: one cross-lane exchange per column and chunk gives both the same maximum, both keep the same running
// power-of-two scale.  Per-token scalars (DropPath scale, LayerNorm statistics) are carried by lane (token & 31) and
// read with two v_readlane + a select.  Control block in the place of the third plane: factors [2][BC], four flag
// words (one per producer wave); final 2^-s [2][BC] in buffer 0.
// ---------------------------------------------------------------------------
// BM >= 0: the operand prologues fixed at compile time (b_mode = BM, the A row scale present iff AR) -- the grouped launch
// picks the instantiation per problem (block-uniform), so the staging loop carries no mode branches; BM < 0: read from p.
// CONV: the 3x3 conv weight gradient of tap `tap` (B rows are the tap-shifted pixels of a [batch][H][Wd] image, zero outside;
// plain operands: BM = 0, no row scale, no PixelShuffle form; both operands below 4 GB: 32-bit row offsets).  A chunk's
// source rows differ between the lane halves, so its loads take a per-lane row offset: lane (token & 31) works out the
// token's source row (-1: none), a load reads it back with two v_readlane + a select, clamps it and store() zeroes the
// tokens that had none -- no branch around a load anywhere, one form for interior and ragged chunks.
template <int W, int BM = -1, bool AR = false, bool NR = false, bool CONV = false>
__device__ __forceinline__ void tnb_body_h(const TnArgs& p, const int s, const int tile, unsigned char* smem, const int tap = 0) {
  const int bmode = BM >= 0 ? BM : p.b_mode;
  const bool arow = BM >= 0 ? AR : (p.a_rowscale != nullptr);
  constexpr int BC = 64 * W, HC = 32 * W;
  constexpr int PLANE = 2 * BC * 64;
  constexpr int BUF = 3 * PLANE;
  constexpr int CTRL = 2 * PLANE;
  constexpr int SINV = 2 * PLANE + 4096;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
  const int nbj = (p.NJ + p.j_tile - 1) / p.j_tile;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * p.i_tile, j0 = bj * p.j_tile;
  const int ivalid = min(p.i_tile, p.NI - i0), jvalid = min(p.j_tile, p.NJ - j0);
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const bool do_colsum = p.part_colsum && bj == 0 && tap == 0;
  const int cdy = CONV ? tap / 3 - 1 : 0, cdx = CONV ? tap % 3 - 1 : 0;

  struct Stage {
    typename ColVec<W>::T rv[2][8];
    float scale;
    float2 stats;
    int trow;                 // CONV: source row of token (lane & 31) of the chunk, -1 = none (outside the image / the slice)
  };
  Stage sg0, sg1;
  const bool isB = wave & 1;
  const int ch = wave >> 1;                    // column half of the operand tile
  const int lt = lane & 31, th = lane >> 5;    // column W-tuple, token half
  const int opvalid = isB ? jvalid : ivalid;
  const int gcol = HC * ch + W * lt;           // this lane's first column in the operand tile
  const int colq = max(min(gcol, opvalid - W), 0);
  const int dsh = gcol - colq;
  const bool ragged = NR ? false : opvalid % W != 0;     // NR: every tile width of the problem is a multiple of W
  const float* const opP = isB ? p.B + j0 : p.A + i0;
  const long opLd = isB ? p.ldb : p.lda;
  const unsigned colb = (unsigned)((16L * th * opLd + colq) * 4);
  float cs[W], sc[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { cs[j] = 0.f; sc[j] = 0x1p126f; }

  auto bcast_i = [&](int v, int k) __attribute__((always_inline)) -> int {       // value of token 16 th + k of the chunk
    const int lo = __builtin_amdgcn_readlane(v, k), hi = __builtin_amdgcn_readlane(v, 16 + k);
    return th ? hi : lo;
  };
  const unsigned ld4 = (unsigned)opLd * 4u, colb0 = (unsigned)colq * 4u;
  auto load_c = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int gm = mc + (lane & 31);
    bool ok = gm < m_end;
    int srow = gm;
    const int x = gm % p.Wd, tq = gm / p.Wd;
    const int y = tq % p.H, bq = tq / p.H;
    const int yy = y + (isB ? cdy : 0), xx = x + (isB ? cdx : 0);
    ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
    srow = (bq * p.H + yy) * p.Wd + xx;
    sg.trow = ok ? srow : -1;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = max(bcast_i(sg.trow, t), 0);
      sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(opP, (unsigned)row * ld4 + colb0);
    }
  };
  auto load = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int tokl = mc + (lane & 31);          // the token whose scalars this lane carries
    const bool interior = mc + TKB <= m_end;    // uniform
    const bool tin = interior || tokl < m_end;
    if (!isB) sg.scale = ldg_f((arow && tin) ? p.a_rowscale + tokl / p.a_rowscale_rows : k_sr_neutral + 1);
    else sg.stats = ldg_f2((bmode == 1 && tin) ? p.ln_stats + 2 * (long)tokl : k_sr_neutral);
    const float* q = opP + (long)mc * opLd;     // uniform, advanced per token
    if (interior) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(q, colb);
        q += opLd;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool ok = mc + 16 * th + t < m_end;
        sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(ok ? (const float*)((const char*)q + colb) : k_tnb_zero_row);
        q += opLd;
      }
    }
  };
  // The same for a chunk that lies entirely inside the slice, WITHOUT a branch around any load: the compiler's s_waitcnt
  // placement keeps the order of the loads in flight only along straight-line code -- behind a join (interior / ragged,
  // operand A / B) it assumed every stage register could have been loaded last, and each store() began with vmcnt(1):
  // it waited for the loads issued just before the barrier, the two-stage prefetch never overlapped anything.  The main
  // loop and its preheader use this form only; the generic one serves the last chunks of a slice.
  auto load_i = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int tokl = mc + (lane & 31);
    if (BM < 0 || AR) sg.scale = ldg_f((!isB && arow) ? p.a_rowscale + tokl / p.a_rowscale_rows : k_sr_neutral + 1);
    if (BM < 0 || BM == 1) sg.stats = ldg_f2((isB && bmode == 1) ? p.ln_stats + 2 * (long)tokl : k_sr_neutral);
    const float* q = opP + (long)mc * opLd;     // uniform, advanced per token
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      sg.rv[t >> 3][t & 7] = ColVec<W>::ldg(q, colb);
      q += opLd;
    }
  };
  auto bcast = [&](float v, int k) __attribute__((always_inline)) -> float {   // scalar of token 16 th + k of the chunk
    const int lo = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k);
    const int hi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 + k);
    return __builtin_bit_cast(float, th ? hi : lo);
  };
  // (Experiments, dropped.  The next chunk's loads issued in the MIDDLE of store(), right after the values have left the stage
  // registers, so that they fly under 1.6 store() periods: with the exact waits of load_i 460-473 -> 536-540 us per launch on
  // the README net -- the data then returns into the register file while the producer is busiest; issued at the end of
  // store() it returns while the wave sits at the barrier.  The same move in tnb_body / tnb_body3: 1-2.5 % of the step lost.
  // THREE register stages over the two LDS buffers (a load gets almost three chunk periods): the LayerNorm instantiation
  // needs more than the 256 registers two waves per SIMD leave -- 169 spills, and scratch traffic shares vmcnt with the
  // prefetch.  The launch WITHOUT any operand prologue (no LayerNorm / DropPath arithmetic: wrong results, same accesses):
  // 465 us against 465 -- the staging arithmetic is not the bound, the reads are (1.74 GB per launch at 3.7-3.9 TB/s).)
  auto store = [&](unsigned char* buf, const Stage& sg) {
    sr_f32x2 x[2][4][W];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (!ragged) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j)
            x[o][t][j] = sr_f32x2{ColVec<W>::at(sg.rv[o][2 * t], j), ColVec<W>::at(sg.rv[o][2 * t + 1], j)};
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j) {
            float e0 = ColVec<W>::at(sg.rv[o][2 * t], W - 1), e1 = ColVec<W>::at(sg.rv[o][2 * t + 1], W - 1);
#pragma unroll
            for (int d = W - 2; d >= 0; --d)
              if (j + d < W && dsh == d) {
                e0 = ColVec<W>::at(sg.rv[o][2 * t], j + d);
                e1 = ColVec<W>::at(sg.rv[o][2 * t + 1], j + d);
              }
            x[o][t][j] = sr_f32x2{e0, e1};
          }
      }
      if (CONV) {             // tokens without a source row (outside the image or the slice) count as zeros
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool ok0 = bcast_i(sg.trow, 8 * o + 2 * t) >= 0, ok1 = bcast_i(sg.trow, 8 * o + 2 * t + 1) >= 0;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            x[o][t][j].x = ok0 ? x[o][t][j].x : 0.f;
            x[o][t][j].y = ok1 ? x[o][t][j].y : 0.f;
          }
        }
      }
      if (!isB) {
        if (arow) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float k0 = bcast(sg.scale, 8 * o + 2 * t), k1 = bcast(sg.scale, 8 * o + 2 * t + 1);
#pragma unroll
            for (int j = 0; j < W; ++j) { x[o][t][j].x *= k0; x[o][t][j].y *= k1; }
          }
        }
        if (do_colsum) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < W; ++j) cs[j] += x[o][t][j].x + x[o][t][j].y;
        }
      } else if (bmode == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float mu0 = bcast(sg.stats.x, 8 * o + 2 * t), mu1 = bcast(sg.stats.x, 8 * o + 2 * t + 1);
          const float rs0 = bcast(sg.stats.y, 8 * o + 2 * t), rs1 = bcast(sg.stats.y, 8 * o + 2 * t + 1);
#pragma unroll
          for (int j = 0; j < W; ++j) {
            x[o][t][j].x = (x[o][t][j].x - mu0) * rs0;
            x[o][t][j].y = (x[o][t][j].y - mu1) * rs1;
          }
        }
      } else if (bmode == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < W; ++j) x[o][t][j] = gelu_fast2(x[o][t][j]);     // the MLP forward's own x Phi(x), packed
      }
    }
    // the columns' maxima over the chunk's 32 tokens (this lane's 16 and lane ^ 32's), the running scales, the factors
    float f[W];
    bool chg = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float mx = 0.f;
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fabsf(x[o][t][j].x), fabsf(x[o][t][j].y)));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      f[j] = 1.f;
      if (mx * sc[j] > 60000.f) {
        const float ns = exp2f(fminf(floorf(log2f(16384.f / mx)), 120.f));
        f[j] = ns / sc[j];
        sc[j] = ns;
        chg = true;
      }
    }
    const bool any = __any(chg);
    float* ctrl = (float*)(buf + CTRL);
    if (th == 0) {
#pragma unroll
      for (int j = 0; j < W; ++j) ctrl[(isB ? BC : 0) + gcol + j] = f[j];
    }
    if (lane == 0) ((int*)(ctrl + 2 * BC))[wave] = any ? 1 : 0;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        unsigned qh[4], ql[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) split2_pair(x[o][t][j].x * sc[j], x[o][t][j].y * sc[j], qh[t], ql[t]);
        unsigned char* dst = buf + unit_slot<W>((isB ? BC : 0) + gcol + j, 2 * th + o) * 16;
        *(u32x4*)(dst) = u32x4{qh[0], qh[1], qh[2], qh[3]};
        *(u32x4*)(dst + PLANE) = u32x4{ql[0], ql[1], ql[2], ql[3]};
      }
  };

  f32x16 acc[W][W];
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  int a_off[W][2], b_off[W][2];
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[i][ks] = unit_slot<W>((wi * W + i) * 32 + r, 2 * ks + h) * 16;
#pragma unroll
  for (int j = 0; j < W; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[j][ks] = unit_slot<W>(BC + (wj * W + j) * 32 + r, 2 * ks + h) * 16;

  const int nch = ((m_end - m_begin + 2 * TKB - 1) / (2 * TKB)) * 2;
  if (producer) {
    __builtin_amdgcn_s_setprio(2);                    // the staging waves are the critical path of a chunk: issue before the MFMA waves
    const int n_int = (m_end - m_begin) / TKB;     // chunks that lie entirely inside the slice
    int c = 0;
    if constexpr (CONV) {        // (three register stages fit here -- 224 registers -- and measured 95.0 / 95.8 us against 92.8 / 93.1)
      load_c(m_begin, sg0);
      load_c(m_begin + TKB, sg1);
      store(smem, sg0);
      load_c(m_begin + 2 * TKB, sg0);
      __syncthreads();
      for (; c < nch; c += 2) {
        store(smem + BUF, sg1);
        load_c(m_begin + (c + 3) * TKB, sg1);
        __syncthreads();
        store(smem, sg0);
        load_c(m_begin + (c + 4) * TKB, sg0);
        __syncthreads();
      }
    } else if (n_int >= 5) {                              // (uniform) straight-line loads from the first one on
      load_i(m_begin, sg0);
      load_i(m_begin + TKB, sg1);
      store(smem, sg0);
      load_i(m_begin + 2 * TKB, sg0);
      __syncthreads();
      for (; c + 4 < n_int; c += 2) {              // the loads of chunks c + 3, c + 4: interior
        store(smem + BUF, sg1);
        load_i(m_begin + (c + 3) * TKB, sg1);
        __syncthreads();
        store(smem, sg0);
        load_i(m_begin + (c + 4) * TKB, sg0);
        __syncthreads();
      }
    } else {
      load(m_begin, sg0);
      load(m_begin + TKB, sg1);
      store(smem, sg0);
      load(m_begin + 2 * TKB, sg0);
      __syncthreads();
    }
    for (; c < nch; c += 2) {
      store(smem + BUF, sg1);
      load(m_begin + (c + 3) * TKB, sg1);
      __syncthreads();
      store(smem, sg0);
      load(m_begin + (c + 4) * TKB, sg0);
      __syncthreads();
    }
    if (th == 0) {
#pragma unroll
      for (int j = 0; j < W; ++j) ((float*)(smem + SINV))[(isB ? BC : 0) + gcol + j] = 1.0f / sc[j];
    }
    __syncthreads();
  } else {
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const unsigned char* cur = smem + (c & 1) * BUF;
      const float* ctrl = (const float*)(cur + CTRL);
      const u32x4 fl = *(const u32x4*)(ctrl + 2 * BC);
      if (__builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w)) {   // some column's scale dropped with this chunk
        float fb[W];
#pragma unroll
        for (int j = 0; j < W; ++j) fb[j] = ctrl[BC + (wj * W + j) * 32 + r];
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float fa = ctrl[(wi * W + i) * 32 + mfma_row(q, lane)];
#pragma unroll
            for (int j = 0; j < W; ++j) acc[i][j][q] *= fa * fb[j];
          }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fa[W][2];
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *(const u32x4*)(cur + pl * PLANE + a_off[i][ks]);
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const u32x4 bh = *(const u32x4*)(cur + b_off[j][ks]);
          const u32x4 bl = *(const u32x4*)(cur + PLANE + b_off[j][ks]);
#pragma unroll
          for (int i = 0; i < W; ++i) acc[i][j] = mfma_h(fa[i][1], bh, acc[i][j]);
#pragma unroll
          for (int i = 0; i < W; ++i) acc[i][j] = mfma_h(fa[i][0], bl, acc[i][j]);
#pragma unroll
          for (int i = 0; i < W; ++i) acc[i][j] = mfma_h(fa[i][0], bh, acc[i][j]);
        }
      }
      __syncthreads();
    }
    __syncthreads();                             // the producers' final 2^-s
    const float* sinv = (const float*)(smem + SINV);
    float* out = p.part + ((long)(s * (CONV ? 9 : 1) + tap) * p.NI) * p.NJ;
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int col = (wj * W + j) * 32 + r;
        if (col >= jvalid) continue;
        const float ib = sinv[BC + col];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (wi * W + i) * 32 + mfma_row(q, lane);
          if (row < ivalid) out[(long)(i0 + row) * p.NJ + j0 + col] = acc[i][j][q] * (sinv[row] * ib);
        }
      }
  }

  if (do_colsum) {
    float* red = (float*)smem;                   // [2][BC]; SINV sits behind it
    if (producer && !isB) {
#pragma unroll
      for (int j = 0; j < W; ++j) red[th * BC + gcol + j] = cs[j];
    }
    __syncthreads();
    if (tid < ivalid) p.part_colsum[(long)s * p.NI + i0 + tid] = red[tid] + red[BC + tid];
  }
}

// ---------------------------------------------------------------------------
// 64-wide conv problems, THREE taps (one kernel row dy) per block.  With one tap per block a 64 x 64 tile gives a
// consumer wave 12 MFMAs per 32-token chunk and barrier while the producers load and split 128 columns for it, and
// every tap block splits the same dY tile again (23 % of the bf16x3 rate on the EDSR shapes).  Here the dY tile of a
// chunk is staged ONCE for the three taps dx = -1, 0, +1 (the three shifted X tiles each get their own LDS image: a
// token octet must be 16-byte aligned for the MFMA operand) in 16-TOKEN chunks: four operand tiles x 16 tokens = one
// staging job per producer wave (wave k stages operand k: dY, X(-1), X(0), X(+1)), 18 MFMAs per consumer wave and
// barrier, 48 KB of LDS -- three blocks per CU as before.  (The same with 32-token chunks -- 36 MFMAs per barrier,
// 96 KB, ONE block per CU -- measured 4.5 % SLOWER than one tap per block: eight waves per CU do not hide the
// producers' load latency.)  Plain operands only (no prologue, no row scale), NI and NJ multiples of 64.
// ---------------------------------------------------------------------------
constexpr int TK3 = 16;                         // tokens per chunk = one k step of the 32x32x16 MFMA
#define SR_TNB3_OCC 6                           // waves per SIMD: three 8-wave blocks per CU
__device__ __forceinline__ int unit_slot3(int col, int u) { return 2 * col + (u ^ ((col >> 3) & 1)); }

//
// F16: the operands as TWO fp16 planes and three products (h*h + h*l + l*h on v_mfma_f32_32x32x16_f16) instead of three
// bf16 planes and six.  fp16 has 5 exponent bits, so every operand COLUMN (a channel of dY or of X) carries a power-of-two
// scale 2^s: the producer lane that owns the column keeps it as a running value that only goes down -- when a chunk's
// largest |x| would pass 60000 after scaling, s drops so that the maximum lands in [8192, 16384] and the lane posts the
// (exact, power-of-two) factor next to the chunk's planes; the consumers multiply their accumulators by factor(row) *
// factor(column) before that chunk's MFMAs (a wave-uniform branch on four flag words; rare after the first chunks).  The
// epilogue multiplies by 2^-s(row) * 2^-s(column).  An element far below its column's maximum keeps an ABSOLUTE error of
// 2^-25 in scaled units = 2^-39 of the column maximum: the sum over tokens -- what a weight gradient is -- stays
// f32-grade (error <= ~2^-22 * sum |a| |b|), which per-column scaling could not give an NT product.
template <int DBG = 0, bool F16 = false>
__device__ __forceinline__ void tnb_body3(const TnArgs& p, const int s, const int tile, const int trow,
                                          unsigned char* smem) {
  constexpr int BC = 64, NOP = 4;              // operand tiles per chunk: dY, X(dx = -1), X(0), X(+1)
  constexpr int PLANE = NOP * BC * 32;         // bytes per plane per chunk buffer (two 16-byte token octets per column)
  constexpr int BUF = 3 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
  const int nbj = p.NJ / BC;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * BC, j0 = bj * BC;
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const int dy = trow - 1;
  const bool do_colsum = p.part_colsum && bj == 0 && trow == 0;

  struct Stage { float rv[2][8]; };            // 16 tokens x this lane's column in flight
  const int op = wave;                         // producer wave k stages operand tile k
  const bool isB = op != 0;
  const int ps_f = p.NI >> 2;
  unsigned colb = (unsigned)lane * 4u;
  if (!isB && p.ps) {
    const int ig = i0 + lane, sp = ig / ps_f, cc = ig - sp * ps_f;
    colb = (unsigned)((((sp >> 1) * 2 * p.Wd + (sp & 1)) * (int)p.lda + cc) * 4);
  }
  const float* const P = isB ? p.B + j0 : (p.ps ? p.A : p.A + i0);      // uniform
  const long ld = isB ? p.ldb : p.lda;
  const int step = (!isB && p.ps) ? 2 : 1;
  float cs = 0.f;
  float sc = 0x1p126f;                         // F16: this lane's column scale ("unset": any non-zero chunk sets it)
  constexpr int CTRL = 2 * PLANE;              // F16: per chunk buffer, in the place of the third plane: factors [4][64], flags [4]
  constexpr int SINV = 2 * PLANE + 2048;       // F16: final 2^-s [4][64] (buffer 0)

  auto load = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int gm = mc + (lane & 15);             // token of this lane's per-token data
    const bool in = gm < m_end;
    const int x = gm % p.Wd, tq = gm / p.Wd;
    const int y = tq % p.H, b = tq / p.H;
    int srow = gm;
    bool ok = in;
    if (isB) {
      const int yy = y + dy, xx = x + op - 2;
      ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
      srow = (b * p.H + yy) * p.Wd + xx;
    } else if (p.ps) {
      srow = (b * 2 * p.H + 2 * y) * 2 * p.Wd + 2 * x;
    }
    const int t_row = ok ? srow : -1;
    const int row0 = __builtin_amdgcn_readlane(t_row, 0);
    const bool dense = __all(t_row == row0 + step * (lane & 15) && row0 >= 0);
    if (dense) {
      const float* q = P + (long)row0 * ld;
      const long adv = step * ld;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        sg.rv[t >> 3][t & 7] = ColVec<1>::ldg(q, colb);
        q += adv;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = __builtin_amdgcn_readlane(t_row, t);
        const float* base = row >= 0 ? P + (long)row * ld : k_tnb_zero_row;
        sg.rv[t >> 3][t & 7] = ColVec<1>::ldg(base, row >= 0 ? colb : 0u);
      }
    }
  };
  auto store = [&](unsigned char* buf, const Stage& sg) __attribute__((always_inline)) {
    if constexpr (F16) {
      float mx = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) mx = fmaxf(mx, fabsf(sg.rv[t >> 3][t & 7]));
      float f = 1.f;
      if (mx * sc > 60000.f) {
        const float ns = exp2f(fminf(floorf(log2f(16384.f / mx)), 120.f));
        f = ns / sc;
        sc = ns;
      }
      const bool ch = __any(f != 1.f);
      float* ctrl = (float*)(buf + CTRL);
      if (ch) ctrl[op * BC + lane] = f;
      if (lane == 0) ((int*)(ctrl + NOP * BC))[op] = ch ? 1 : 0;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        unsigned qh[4], ql[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float e0 = sg.rv[o][2 * t], e1 = sg.rv[o][2 * t + 1];
          if (do_colsum) cs += e0 + e1;            // read by the dY wave only
          split2_pair(e0 * sc, e1 * sc, qh[t], ql[t]);
        }
        unsigned char* dst = buf + unit_slot3(op * BC + lane, o) * 16;
        *(u32x4*)(dst) = u32x4{qh[0], qh[1], qh[2], qh[3]};
        *(u32x4*)(dst + PLANE) = u32x4{ql[0], ql[1], ql[2], ql[3]};
      }
      return;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      unsigned qh[4], qm[4], ql[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float e0 = sg.rv[o][2 * t], e1 = sg.rv[o][2 * t + 1];
        if (do_colsum) cs += e0 + e1;              // read by the dY wave only
        split3_pair(e0, e1, qh[t], qm[t], ql[t]);
      }
      const u32x4 ph = {qh[0], qh[1], qh[2], qh[3]}, pm = {qm[0], qm[1], qm[2], qm[3]},
                  pl = {ql[0], ql[1], ql[2], ql[3]};
      unsigned char* dst = buf + unit_slot3(op * BC + lane, o) * 16;
      *(u32x4*)(dst) = ph;
      *(u32x4*)(dst + PLANE) = pm;
      *(u32x4*)(dst + 2 * PLANE) = pl;
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
  const int a_off = unit_slot3(wi * 32 + r, h) * 16;
  int b_off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) b_off[k] = unit_slot3((1 + k) * BC + wj * 32 + r, h) * 16;

  // (Round 6, measured and dropped: THREE register stages -- a chunk's rows requested three chunk periods ahead.  The role
  // ablations say the staging path is the longer pole (tools/mb_tnb3.py at 8 x 256 x 256, 64 -> 256 channels: all 960 us | no
  // MFMAs 739 | no staging 538), but the third stage does not fit: 127 spilled registers at the 80 of three blocks per CU, 50
  // at the 128 of two -- scratch traffic shares vmcnt with the prefetch: EDSR x8 1,810 -> 1,143 patches/s.)
  const int nch = ((m_end - m_begin + 2 * TK3 - 1) / (2 * TK3)) * 2;     // even
  if (producer) {
    __builtin_amdgcn_s_setprio(2);                    // staging waves first: they are the critical path of a chunk
    Stage sg0, sg1;                              // chunk parity
    load(m_begin, sg0);
    load(m_begin + TK3, sg1);
    store(smem, sg0);
    load(m_begin + 2 * TK3, sg0);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
      if (DBG != 2 && DBG != 5) {
        store(smem + BUF, sg1);                  // chunk c+1
        load(m_begin + (c + 3) * TK3, sg1);
      }
      __syncthreads();
      if (DBG != 2 && DBG != 5) {
        store(smem, sg0);                        // chunk c+2
        load(m_begin + (c + 4) * TK3, sg0);
      }
      __syncthreads();
    }
    if constexpr (F16) {
      ((float*)(smem + SINV))[op * BC + lane] = 1.0f / sc;
      __syncthreads();
    }
  } else {
    __syncthreads();
    u32x4 g5[8];
    if (DBG == 5) {
#pragma unroll
      for (int q = 0; q < 8; ++q) g5[q] = *(const u32x4*)(smem + q * 1024 + lane * 16);
    }
    for (int c = 0; c < nch; ++c) {
      const unsigned char* cur = smem + (c & 1) * BUF;
      if constexpr (F16) {
        if (DBG != 1) {
          const float* ctrl = (const float*)(cur + CTRL);
          const u32x4 fl = *(const u32x4*)(ctrl + NOP * BC);
          if (__builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w)) {       // some column's scale dropped with this chunk: bring the sums along
            float fa_[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) fa_[q] = fl.x ? ctrl[wi * 32 + mfma_row(q, lane)] : 1.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const unsigned on = k == 0 ? fl.y : (k == 1 ? fl.z : fl.w);
              const float fb = on ? ctrl[(1 + k) * BC + wj * 32 + r] : 1.f;
#pragma unroll
              for (int q = 0; q < 16; ++q) acc[k][q] *= fa_[q] * fb;
            }
          }
          // DBG 5 (experiments): no fragment reads -- the operands are whatever the registers of `g5` hold (read once in front
          // of the loop): what remains per chunk is the flag word, nine MFMAs and the barrier
          const u32x4 ah = DBG == 5 ? g5[0] : *(const u32x4*)(cur + a_off), al = DBG == 5 ? g5[1] : *(const u32x4*)(cur + PLANE + a_off);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const u32x4 bh = DBG == 5 ? g5[2 + k] : *(const u32x4*)(cur + b_off[k]);
            const u32x4 bl = DBG == 5 ? g5[5 + k] : *(const u32x4*)(cur + PLANE + b_off[k]);
            acc[k] = mfma_h(al, bh, acc[k]);
            acc[k] = mfma_h(ah, bl, acc[k]);
            acc[k] = mfma_h(ah, bh, acc[k]);
          }
        }
        __syncthreads();
        continue;
      }
      if (DBG != 1) {
        u32x4 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[pl] = *(const u32x4*)(cur + pl * PLANE + a_off);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const u32x4 fb0 = *(const u32x4*)(cur + b_off[k]);
          const u32x4 fb1 = *(const u32x4*)(cur + PLANE + b_off[k]);
          const u32x4 fb2 = *(const u32x4*)(cur + 2 * PLANE + b_off[k]);
          acc[k] = mfma_bf(fa[1], fb1, acc[k]);
          acc[k] = mfma_bf(fa[0], fb2, acc[k]);
          acc[k] = mfma_bf(fa[2], fb0, acc[k]);
          acc[k] = mfma_bf(fa[0], fb1, acc[k]);
          acc[k] = mfma_bf(fa[1], fb0, acc[k]);
          acc[k] = mfma_bf(fa[0], fb0, acc[k]);
        }
      }
      __syncthreads();
    }
    const int col = wj * 32 + r;
    if constexpr (F16) {
      __syncthreads();                             // the producers' final 2^-s
      const float* sinv = (const float*)(smem + SINV);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float ib = sinv[(1 + k) * BC + col];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[k][q] *= sinv[wi * 32 + mfma_row(q, lane)] * ib;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float* out = p.part + ((long)(s * 9 + 3 * trow + k) * p.NI) * p.NJ;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int io = i0 + wi * 32 + mfma_row(q, lane);       // p.ps: kernel row sp*F + c is torch channel c*4 + sp
        out[(long)(p.ps ? (io % ps_f) * 4 + io / ps_f : io) * p.NJ + j0 + col] = acc[k][q];
      }
    }
  }

  if (do_colsum) {
    float* red = (float*)smem;                   // the chunk buffers are dead now
    if (producer && op == 0) red[lane] = cs;
    __syncthreads();
    if (tid < BC) {
      const int io = i0 + tid;
      p.part_colsum[(long)s * p.NI + (p.ps ? (io % ps_f) * 4 + io / ps_f : io)] = red[tid];
    }
  }
}

// ---------------------------------------------------------------------------
// 64-wide conv problems, ALL NINE taps per block (round 6; two fp16 planes / three products only).
//
// What the three-tap form costs (role ablations, tools/mb_tnb3.py, 8 x 256 x 256 pixels, 64 -> 256 channels: everything
// 960 us | consumers without their MFMAs 739 | producers without loads and stores 538 -- against 186 us of matrix time):
//   * staging: a tap row's block stages dY and three shifted X tiles for its three taps -- twelve operand tiles per chunk
//     position for the nine taps, every one split (VALU) and written to LDS, and the staging waves run at the latency of their
//     own loads (two chunks ahead, 85 registers per wave with three blocks per CU: no room for a third stage);
//   * LDS: a consumer wave reads 8 KB of fragments for nine MFMAs.
// Here a tap (dy, dx) pairs the dY tile shifted by -dx ALONG its row with the X tile of row y + dy:
//     dW[dy, dx][co][ci] = sum_q dY[q - (0, dx)][co] * X[q + (dy, 0)][ci]          (q = p + (0, dx))
// -- SIX operand tiles per chunk (dY for dx = -1, 0, +1; X for dy = -1, 0, +1; the X tiles are plain rows: aligned, always
// dense) feed all nine taps: half the loads, splits and LDS writes per MFMA.  One block per CU, 14 waves: six staging waves
// (one operand tile each, THREE register stages: a chunk's rows are requested three chunk periods ahead) and eight matrix
// waves = 2 x 2 quadrants of the 64 x 64 tile x two tap groups (taps 0-4 / 5-8: 15 + 12 MFMAs per SIMD and chunk, the same
// on all four SIMDs); a matrix wave reads ten fragments (three dY shifts, two X rows; 10 KB) for 15 / 12 MFMAs.  Partial
// sums leave in the layout of the three-tap form ([slice][tap][NI][NJ]): same reducers.
// ---------------------------------------------------------------------------
constexpr int T9_OPS = 6;
constexpr int T9_PLANE = T9_OPS * 64 * 32;        // bytes per plane and chunk buffer
constexpr int T9_CTRL = 2 * T9_PLANE;             // factors [6][64] floats, then the six flag words (+ 2 pad)
constexpr int T9_BUF = 2 * T9_PLANE + 2048;
constexpr int T9_SINV = 2 * T9_BUF;               // final 2^-s [6][64]
constexpr int T9_LDS = 2 * T9_BUF + 2048;
constexpr int T9_THREADS = 14 * 64;

// DBG (experiments build, SRHIP_TN_DBG; results are wrong on purpose): 1 = no MFMAs | 2 = staging waves neither load nor store
// | 6 = matrix waves read no fragments (one register quad stands in for all of them) | 7 = 2 + 6 | 8 = 7 without any barrier
// in the chunk loops | 9 = 7 with the staging waves gone (they return at once: barriers among the eight matrix waves only)
template <int H2, int DBG = 0>
__device__ __forceinline__ void tnb9_consume(const TnArgs& p, const int s, const int i0, const int j0, const int nch,
                                             unsigned char* smem, const int wi, const int wj, const int lane) {
  // tap t = 5 H2 + u: (dy index, dx index) = (t / 3, t % 3)
  constexpr int NU = H2 ? 4 : 5;
  constexpr int DY0 = H2 ? 1 : 0;                  // the two X rows this wave reads: DY0, DY0 + 1
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;
  int a_off[3], b_off[2];
#pragma unroll
  for (int d = 0; d < 3; ++d) a_off[d] = unit_slot3(d * 64 + wi * 32 + r, h) * 16;
#pragma unroll
  for (int d = 0; d < 2; ++d) b_off[d] = unit_slot3((3 + DY0 + d) * 64 + wj * 32 + r, h) * 16;
  __syncthreads();
  long long ts0 = 0, ts1 = 0;
  if (DBG == 3) ts0 = (long long)wall_clock64();
  for (int c = 0; c < nch; ++c) {
    const unsigned char* cur = smem + (c & 1) * T9_BUF;
    const float* ctrl = (const float*)(cur + T9_CTRL);
    const u32x4 f0 = *(const u32x4*)(ctrl + T9_OPS * 64);
    const sr_u32x2 f1 = *(const sr_u32x2*)(ctrl + T9_OPS * 64 + 4);
    if (__builtin_amdgcn_readfirstlane(f0.x | f0.y | f0.z | f0.w | f1.x | f1.y)) {   // some column's scale dropped with this chunk
      const unsigned fl[6] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y};
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        constexpr int T0 = 5 * H2;
        const int t = T0 + u, dyi = t / 3, dxi = t - 3 * dyi;
        if (fl[dxi] | fl[3 + dyi]) {
          const float fb = fl[3 + dyi] ? ctrl[(3 + dyi) * 64 + wj * 32 + r] : 1.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[u][q] *= (fl[dxi] ? ctrl[dxi * 64 + wi * 32 + mfma_row(q, lane)] : 1.f) * fb;
        }
      }
    }
    // All ten fragments first, then the three product terms as three PASSES over the wave's units: an accumulator is touched
    // again only NU MFMAs later.  (Term by term per unit -- three MFMAs in a row into the same accumulator -- the matrix waves
    // alone, without staging and without fragment reads, ran 695 of the launch's 794 us: a dependent 32x32x16 MFMA waits out
    // the whole latency of the one in front of it.)
    u32x4 ah[3], al[3], bh[2], bl[2];
    const u32x4 dummy = u32x4{(unsigned)lane, 0x3c003c00u, 0x3c003c00u, (unsigned)c};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      bh[d] = (DBG >= 6) ? dummy : *(const u32x4*)(cur + b_off[d]);
      bl[d] = (DBG >= 6) ? dummy : *(const u32x4*)(cur + T9_PLANE + b_off[d]);
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      ah[dx] = (DBG >= 6) ? dummy : *(const u32x4*)(cur + a_off[dx]);
      al[dx] = (DBG >= 6) ? dummy : *(const u32x4*)(cur + T9_PLANE + a_off[dx]);
    }
    if (DBG != 1) {
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int t = 5 * H2 + u, dyi = t / 3, dxi = t - 3 * dyi;
          if (DBG == 10) {            // (experiment: the bf16 form of the same MFMA shape on the same registers, no barriers)
            acc[u] = mfma_bf(term == 0 ? al[dxi] : ah[dxi], term == 1 ? bl[dyi - DY0] : bh[dyi - DY0], acc[u]);
          } else if (term == 0) acc[u] = mfma_h(al[dxi], bh[dyi - DY0], acc[u]);
          else if (term == 1) acc[u] = mfma_h(ah[dxi], bl[dyi - DY0], acc[u]);
          else acc[u] = mfma_h(ah[dxi], bh[dyi - DY0], acc[u]);
        }
      }
    }
    if (DBG != 8 && DBG != 10) __syncthreads();
  }
  if (DBG == 3) ts1 = (long long)wall_clock64();
  if (DBG != 9) __syncthreads();                   // the staging waves' final 2^-s
  const float* sinv = (const float*)(smem + T9_SINV);
  const int col = wj * 32 + r;
  const int ps_f = p.NI >> 2;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int t = 5 * H2 + u, dyi = t / 3, dxi = t - 3 * dyi;
    const float ib = sinv[(3 + dyi) * 64 + col];
    float* out = p.part + ((long)(s * 9 + t) * p.NI) * p.NJ;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int io = i0 + wi * 32 + mfma_row(q, lane);         // p.ps: kernel row sp*F + c is torch channel c*4 + sp
      out[(long)(p.ps ? (io % ps_f) * 4 + io / ps_f : io) * p.NJ + j0 + col] = acc[u][q] * (sinv[dxi * 64 + wi * 32 + mfma_row(q, lane)] * ib);
    }
  }
  if (DBG == 3 && lane == 0 && p.part_colsum) {      // stamps (10-ns units) of block 0 .. 3's matrix waves: loop start, loop end, done
    const long long ts2 = (long long)wall_clock64();
    if (blockIdx.x < 4) {
      long long* d = (long long*)p.part_colsum + ((long)blockIdx.x * 8 + (H2 * 4 + wi * 2 + wj)) * 4;
      d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = nch;
    }
  }
}

template <int DBG = 0>
__device__ __forceinline__ void tnb_body9(const TnArgs& p, const int s, const int tile, unsigned char* smem) {
  constexpr int BC = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 8;
  const int nbj = p.NJ / BC;
  const int bi = tile / nbj, bj = tile - bi * nbj;
  const int i0 = bi * BC, j0 = bj * BC;
  const int m_begin = s * p.rows_per_slice;
  const int m_end = min(p.M, m_begin + p.rows_per_slice);
  const int nch = ((m_end - m_begin + 3 * TK3 - 1) / (3 * TK3)) * 3;      // a multiple of 3: three register stages
  if (!producer) {
    const int q4 = wave & 3;
    if (wave < 4) tnb9_consume<0, DBG>(p, s, i0, j0, nch, smem, q4 >> 1, q4 & 1, lane);
    else tnb9_consume<1, DBG>(p, s, i0, j0, nch, smem, q4 >> 1, q4 & 1, lane);
    if (p.part_colsum && bj == 0) __syncthreads();            // the column sums' exchange below
    return;
  }
  if (DBG == 9) return;
  // ---------------- staging wave `op`: 0..2 = dY shifted by dx = op - 1 along the row, 3..5 = X of row y + (op - 4)
  const int op = wave - 8;
  const bool isB = op >= 3;
  const int sh = isB ? op - 4 : op - 1;
  const bool do_colsum = p.part_colsum && bj == 0 && op == 1;
  struct Stage { float rv[2][8]; };
  const int ps_f = p.NI >> 2;
  unsigned colb = (unsigned)lane * 4u;
  if (!isB && p.ps) {
    const int ig = i0 + lane, sp = ig / ps_f, cc = ig - sp * ps_f;
    colb = (unsigned)((((sp >> 1) * 2 * p.Wd + (sp & 1)) * (int)p.lda + cc) * 4);
  }
  const float* const P = isB ? p.B + j0 : (p.ps ? p.A : p.A + i0);      // uniform
  const long ld = isB ? p.ldb : p.lda;
  const int step = (!isB && p.ps) ? 2 : 1;
  float cs = 0.f;
  float sc = 0x1p126f;
  // The chunks are loaded in order, 16 tokens apart: lane (token & 15) carries its token's (x, y, image) and steps it --
  // three integer divisions per chunk otherwise (a third of the staging wave's instructions).
  int cx, cy, cb;
  {
    const int gm0 = m_begin + (lane & 15);
    cx = gm0 % p.Wd;
    const int tq = gm0 / p.Wd;
    cy = tq % p.H;
    cb = tq / p.H;
  }
  // last pixel row a dense chunk may start its 16th token on (PixelShuffle form: a lane's column offset reaches up to one
  // shuffled row + one pixel further)
  const int last_row = (!isB && p.ps) ? p.batch * p.H * p.Wd * 4 - 1 - (2 * p.Wd + 1) : p.batch * p.H * p.Wd - 1;
  auto load = [&](int mc, Stage& sg) __attribute__((always_inline)) {
    const int gm = mc + (lane & 15);
    const int x = cx, y = cy, b = cb;
    cx += 16;
    while (cx >= p.Wd) { cx -= p.Wd; if (++cy == p.H) { cy = 0; ++cb; } }
    bool ok = gm < m_end;
    int srow;
    if (isB) {
      const int yy = y + sh;
      ok = ok && yy >= 0 && yy < p.H;
      srow = (b * p.H + yy) * p.Wd + x;
    } else {
      const int xx = x - sh;
      ok = ok && xx >= 0 && xx < p.Wd;
      srow = p.ps ? (b * 2 * p.H + 2 * y) * 2 * p.Wd + 2 * xx : (b * p.H + y) * p.Wd + xx;
    }
    // Dense form: the valid tokens of the chunk sit on ONE arithmetic row sequence base + step t -- then all sixteen rows
    // are read from it (a token without a source -- the pixel before a row's first, the row above the image -- reads its
    // neighbour's row instead, in bounds) and the invalid ones are zeroed afterwards: no per-token address, no branch
    // around a load.  Anything else (a chunk across an image's last row with dy = +1, the operand's first / last rows) takes
    // the per-token form.
    const int cand = srow - step * (lane & 15);
    int base = ok ? cand : -2147483647 - 1;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) base = max(base, __shfl_xor(base, o, 64));
    const unsigned okmask = (unsigned)__ballot(ok) & 0xffffu;
    const bool dense = okmask != 0u && __all(!ok || cand == base) && __builtin_amdgcn_readfirstlane(base) >= 0 &&
                       __builtin_amdgcn_readfirstlane(base) + 15 * step <= last_row;
    if (dense) {
      const float* q = P + (long)__builtin_amdgcn_readfirstlane(base) * ld;
      const long adv = step * ld;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float v = ColVec<1>::ldg(q, colb);
        sg.rv[t >> 3][t & 7] = v;
        q += adv;
      }
      if (okmask != 0xffffu) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (!((okmask >> t) & 1u)) sg.rv[t >> 3][t & 7] = 0.f;
      }
    } else {
      const int t_row = ok ? srow : -1;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = __builtin_amdgcn_readlane(t_row, t);
        const float* base_p = row >= 0 ? P + (long)row * ld : k_tnb_zero_row;
        sg.rv[t >> 3][t & 7] = ColVec<1>::ldg(base_p, row >= 0 ? colb : 0u);
      }
    }
  };
  auto store = [&](unsigned char* buf, const Stage& sg) __attribute__((always_inline)) {
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) mx = fmaxf(mx, fabsf(sg.rv[t >> 3][t & 7]));
    float f = 1.f;
    if (mx * sc > 60000.f) {
      const float ns = exp2f(fminf(floorf(log2f(16384.f / mx)), 120.f));
      f = ns / sc;
      sc = ns;
    }
    const bool ch = __any(f != 1.f);
    float* ctrl = (float*)(buf + T9_CTRL);
    if (ch) ctrl[op * BC + lane] = f;
    if (lane == 0) ((int*)(ctrl + T9_OPS * BC))[op] = ch ? 1 : 0;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      unsigned qh[4], ql[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float e0 = sg.rv[o][2 * t], e1 = sg.rv[o][2 * t + 1];
        if (do_colsum) cs += e0 + e1;
        split2_pair(e0 * sc, e1 * sc, qh[t], ql[t]);
      }
      unsigned char* dst = buf + unit_slot3(op * BC + lane, o) * 16;
      *(u32x4*)(dst) = u32x4{qh[0], qh[1], qh[2], qh[3]};
      *(u32x4*)(dst + T9_PLANE) = u32x4{ql[0], ql[1], ql[2], ql[3]};
    }
  };
  __builtin_amdgcn_s_setprio(2);
  Stage sa, sb, sc3;
  load(m_begin, sa);
  load(m_begin + TK3, sb);
  load(m_begin + 2 * TK3, sc3);
  if (lane == 0 && op == 0) { int* fw = (int*)(smem + T9_CTRL) + T9_OPS * BC; fw[6] = 0; fw[7] = 0; fw = (int*)(smem + T9_BUF + T9_CTRL) + T9_OPS * BC; fw[6] = 0; fw[7] = 0; }
  store(smem, sa);
  load(m_begin + 3 * TK3, sa);
  __syncthreads();
  // chunk c + K + 1 goes to buffer (c + K + 1) & 1 from the stage that holds it; that stage then takes chunk c + K + 4
#define SR_T9STEP(K_, ST_)                                                                                   \
  if (DBG != 2 && DBG < 7) { store(smem + (((c + (K_) + 1) & 1) ? T9_BUF : 0), ST_); load(m_begin + (c + (K_) + 4) * TK3, ST_); } \
  if (DBG != 8 && DBG != 10) __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; c += 3) {
    SR_T9STEP(0, sb) SR_T9STEP(1, sc3) SR_T9STEP(2, sa)
  }
#undef SR_T9STEP
  ((float*)(smem + T9_SINV))[op * BC + lane] = 1.0f / sc;
  __syncthreads();
  if (p.part_colsum && bj == 0) {
    float* red = (float*)smem;                   // the chunk buffers are dead: every matrix wave is past its last chunk
    if (op == 1) red[lane] = cs;
    __syncthreads();
    if (op == 2) {
      const int io = i0 + lane;
      p.part_colsum[(long)s * p.NI + (p.ps ? (io % ps_f) * 4 + io / ps_f : io)] = red[lane];
    }
  }
}

template <int DBG = 0>
__global__ void __launch_bounds__(T9_THREADS) k_tnb9(TnArgs p, int tiles, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  tnb_body9<DBG>(p, L / tiles, L % tiles, smem);
}

// ---------------------------------------------------------------------------
// The STRIP form of the nine-tap block (round 6, third form; images whose width is a multiple of 64).
//
// What the token-order form above still pays (role ablations, 8 x 256 x 256, 64 -> 256, same box: everything 786 us | matrix
// waves alone 362 | only TWO of the six staging waves working 702 | only ONE 656): a staging wave needs ~1.2 us for the ~350
// instructions of its 16-token chunk (load addresses, dense-chunk test, maxima, split, LDS writes), the block waits for it at
// every chunk's barrier, and there are SIX tiles per chunk because the +-1 shifts of dY and the +-1 rows of X are staged as
// copies.  Here a block walks a 64-pixel-wide STRIP of the image downwards, one pixel row (four 16-token chunks) per barrier:
//   * X row y + 1 is staged ONCE and stays in a four-row ring: it is the dy = +1 operand of row y, dy = 0 of y + 1 and
//     dy = -1 of y + 2;
//   * dY row y is staged ONCE with a one-pixel halo on either side (the neighbour strip's pixel, zero at the image's edge);
//     the matrix waves build the dx = +-1 operands in registers: a lane's k-octet plus the neighbour octet's edge token,
//     funnel-shifted by 16 bits (v_alignbit_b32, 16 per chunk and wave);
// -- 130 tokens staged per 64 tokens of product instead of 384, by FOUR staging waves (one per SIMD) that work on 32 / 33
// tokens per barrier with one row-address computation (no per-token coordinates: a strip's rows are arithmetic sequences of
// the flattened (image, y) row index, in the PixelShuffle form too), one row ahead.  Rows outside the
// image (dy = -1 above the first row, +1 below the last): the matrix waves leave those taps out for that row.  A lane gets its
// octet's neighbour tokens from a small per-octet edge array (first | last token, one dword per column: conflict-free, where
// the octets' own dwords 32 bytes apart would be read four lanes to a bank); the halo pixels are that array's outer entries.
// 108 KB of LDS (two dY rows, four X rows); sixteen waves, one block per CU: per SIMD one staging wave and the three matrix
// waves of one 32 x 32 quadrant, one tap ROW each (nine MFMAs per chunk from two X and two dY fragment reads; 48 accumulator
// registers, so the compiler can keep several chunks' reads in flight inside the 128 of a 16-wave block).
// Slices are row ranges of one strip (S >= the strip count).
//
// EXPONENTS.  A column's two planes hold x * sc, sc a power of two FIXED before the block's first row and never changed
// while it runs: the matrix waves carry no per-chunk state.  (The running exponents of the token-order form cost its matrix
// waves six flag words per chunk turned into a scalar -- v_readfirstlane -- for an almost-never-taken branch: a VALU -> SGPR
// transfer behind MFMAs waits until the wave's matrix instructions have drained, the pipe idles once per chunk: its
// MFMA-only loop runs 635 us with the check and 244 without.)  fp16 leaves room for a guess: the column's largest magnitude
// of the block's first row (but at least 2^-8 of the tile's) is put into [16, 32) -- a later pixel may be 2,000 x larger
// before the high plane overflows, and the pair resolves 2^-25 absolute, 2^-29 of that first maximum (f32 itself: 2^-24
// relative).  The two waves that stage
// halves of the same columns agree on it through LDS.  The staging waves track every column's TRUE maximum as they go; a
// block whose guess did not hold (some column's maximum x sc outside [1/8, 60000]) says so in a word behind the call's
// partial sums and leaves its maxima there, and a SECOND PASS (k_tnb9s<.., 1>: the same grid, unflagged blocks leave at
// once) runs exactly those blocks again with exact exponents and overwrites their partial sums.  (The second run as a loop
// inside the kernel cost the first pass its registers: 111 -> 128 + scratch, 102 -> 141 us on a 8 x 128 x 128 problem.)
// ---------------------------------------------------------------------------
constexpr int S9_SUB = 64 * 32;                   // bytes per chunk sub-buffer and plane: 64 columns x two octets x 16 B
constexpr int S9_DYPL = 4 * S9_SUB;               // a dY row, one plane
constexpr int S9_EDGE = 10 * 256;                 // .. and its octets' edge tokens, one plane: [octet -1 .. 8][64 columns] dwords
constexpr int S9_DYE = 2 * S9_DYPL;               // (first token | last token << 16); octets -1 and 8 are the halo pixels
constexpr int S9_DYROW = 2 * S9_DYPL + 2 * S9_EDGE;
constexpr int S9_XPL = 4 * S9_SUB;
constexpr int S9_XROW = 2 * S9_XPL;
constexpr int S9_X0 = 2 * S9_DYROW;               // the X ring (four rows) behind the two dY rows
constexpr int S9_MISC = S9_X0 + 4 * S9_XROW;
constexpr int S9_SINV = S9_MISC;                  // [2][64]: 2^-s of the dY / X columns
constexpr int S9_MAX = S9_MISC + 512;             // [4][64]: the staging waves' column maxima
constexpr int S9_CS = S9_MISC + 1536;             // [2][64]: column sums of the two dY halves
constexpr int S9_RETRY = S9_MISC + 2048;
constexpr int S9_LDS = S9_MISC + 2304;
constexpr int S9_THREADS = 16 * 64;
static_assert(S9_LDS <= 160 * 1024, "LDS of one CU");

// matrix wave of tap row DYI (dy = DYI - 1; taps 3 DYI + dx index) and quadrant (wi, wj) of the 64 x 64 tile
template <int DYI, int DBG = 0>
__device__ __forceinline__ void tnb9s_consume(const TnArgs& p, const int s, const int i0, const int j0, const int r0, const int r1,
                                              const int nsteps, unsigned char* smem, const int wi, const int wj, const int lane) {
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[3];
  const int a_self = unit_slot3(wi * 32 + r, h) * 16;                     // + c * S9_SUB: the lane's octet 2 c + h of chunk c
  const int a_edge = S9_DYE + h * 256 + (wi * 32 + r) * 4;                // + c * 512: edge tokens of octet 2 c + h - 1, + 512: 2 c + h + 1
  const int b_self = unit_slot3(wj * 32 + r, h) * 16;
  {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;
    __syncthreads();                                 // (the staging waves' maxima)
    __syncthreads();                                 // rows r0 - 1 .. r0 + 1 of X and r0 of dY are in place
    int y = r0 % p.H;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
      const int row = r0 + k;
      // the wave's X row lies outside the image (dy = -1 above the first row, +1 below the last): nothing to add
      const bool inside = DYI == 1 || (DYI == 0 ? y > 0 : y < p.H - 1);
      if (row < r1 && inside) {
        const unsigned char* dyb = smem + (row & 1) * S9_DYROW;
        const unsigned char* xb = smem + S9_X0 + ((row + DYI - 1 + 4) & 3) * S9_XROW + b_self;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x4 dummy = u32x4{(unsigned)lane, 0x3c003c00u, 0x3c003c00u, (unsigned)c};
          const u32x4 bh = (DBG >= 6) ? dummy : *(const u32x4*)(xb + c * S9_SUB);
          const u32x4 bl = (DBG >= 6) ? dummy : *(const u32x4*)(xb + S9_XPL + c * S9_SUB);
          u32x4 a[2][3];                             // [plane][dx index]: dx index 2 (dx = +1) pairs token q with dY[q - 1], 0 with dY[q + 1]
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const u32x4 d = (DBG >= 6) ? dummy : *(const u32x4*)(dyb + pl * S9_DYPL + a_self + c * S9_SUB);
            const unsigned pv = (DBG >= 6) ? 0u : *(const unsigned*)(dyb + pl * S9_EDGE + a_edge + c * 512);         // last token: high half
            const unsigned nx = (DBG >= 6) ? 0u : *(const unsigned*)(dyb + pl * S9_EDGE + a_edge + c * 512 + 512);   // first token: low half
            a[pl][1] = d;
            a[pl][2] = u32x4{__builtin_amdgcn_alignbit(d.x, pv, 16), __builtin_amdgcn_alignbit(d.y, d.x, 16),
                             __builtin_amdgcn_alignbit(d.z, d.y, 16), __builtin_amdgcn_alignbit(d.w, d.z, 16)};
            a[pl][0] = u32x4{__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16),
                             __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(nx, d.w, 16)};
          }
          if (DBG != 1) {
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
              for (int u = 0; u < 3; ++u) {
                if (term == 0) acc[u] = mfma_h(a[1][u], bh, acc[u]);
                else if (term == 1) acc[u] = mfma_h(a[0][u], bl, acc[u]);
                else acc[u] = mfma_h(a[0][u], bh, acc[u]);
              }
            }
          }
        }
      }
      __syncthreads();
      y = (y + 1 == p.H) ? 0 : y + 1;
    }
    __syncthreads();                                 // the staging waves' 2^-s and maxima
    __syncthreads();                                 // (their verdict on the exponents)
  }
  const float* sinv = (const float*)(smem + S9_SINV);
  const int col = wj * 32 + r;
  const int ps_f = p.NI >> 2;
  const float ib = sinv[64 + col];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    float* out = p.part + ((long)(s * 9 + 3 * DYI + u) * p.NI) * p.NJ;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int il = wi * 32 + mfma_row(q, lane), io = i0 + il;   // p.ps: kernel row sp*F + c is torch channel c*4 + sp
      if (io < p.NI && j0 + col < p.NJ)
        out[(long)(p.ps ? (io % ps_f) * 4 + io / ps_f : io) * p.NJ + j0 + col] = acc[u][q] * (sinv[il] * ib);
    }
  }
}

// words per block behind the partial sums (S9_AUX floats each): [0] = 1 if some column's exponent did not hold in pass 0,
// [1 .. 4] = why, per staging wave (1: a column overflowed, 2: one fell below the resolution), [64 ..): their column maxima
constexpr int S9_AUX = 64 + 4 * 64;

template <int DBG = 0>
__device__ __forceinline__ void tnb_body9s(const TnArgs& p, const int s, const int tile, const int tiles, const int pass,
                                           unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  float* const aux = p.aux + (long)(s * tiles + tile) * S9_AUX;
  // pass 1 (the same grid right behind pass 0): only the blocks whose exponents did not hold run again, with exact ones
  if (pass && !__builtin_amdgcn_readfirstlane(*(const int*)aux)) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbj = (p.NJ + 63) / 64;                // channel counts that are no multiple of 64: the last tile's columns beyond
  const int bi = tile / nbj, bj = tile - bi * nbj; // NI / NJ are zeros in the planes and are not written
  const int i0 = bi * 64, j0 = bj * 64;
  // slice -> (strip, row range of the flattened (image, y) index)
  // strip k owns the slices [k S / nstrips, (k + 1) S / nstrips): at least one each (S >= nstrips)
  const int nstrips = p.Wd >> 6;
  int strip = 0;
  while ((strip + 1) * p.S / nstrips <= s) ++strip;
  const int s_first = strip * p.S / nstrips, sps = (strip + 1) * p.S / nstrips - s_first;
  const int ss = s - s_first;
  const int RT = p.batch * p.H;
  const int rps = (RT + sps - 1) / sps;
  const int r0 = min(RT, ss * rps), r1 = min(RT, r0 + rps);
  const int nsteps = r1 - r0;
  const int x0 = strip * 64;
  const bool colsum = p.part_colsum && bj == 0;
  if (wave < 12) {              // waves q, q + 4, q + 8 (one SIMD): quadrant q of the tile, tap rows dy = -1, 0, +1
    const int q4 = wave & 3;
    if (wave < 4) tnb9s_consume<0, DBG>(p, s, i0, j0, r0, r1, nsteps, smem, q4 >> 1, q4 & 1, lane);
    else if (wave < 8) tnb9s_consume<1, DBG>(p, s, i0, j0, r0, r1, nsteps, smem, q4 >> 1, q4 & 1, lane);
    else tnb9s_consume<2, DBG>(p, s, i0, j0, r0, r1, nsteps, smem, q4 >> 1, q4 & 1, lane);
    if (colsum) __syncthreads();
    return;
  }
  // ---------------- staging wave sw: 0 / 1 = dY tokens [-1, 32) / [32, 65) of the strip's row, 2 / 3 = X tokens [0, 32) / [32, 64)
  const int sw = wave - 12;
  const bool isB = sw >= 2;
  const int half = sw & 1;
  struct Stage { float v[32]; float halo; };
  const int ps_f = p.NI >> 2;
  unsigned colb = (unsigned)lane * 4u;
  if (!isB && p.ps) {
    const int ig = i0 + lane, sp = ig / ps_f, cc = ig - sp * ps_f;
    colb = (unsigned)((((sp >> 1) * 2 * p.Wd + (sp & 1)) * (int)p.lda + cc) * 4);
  }
  const bool psA = !isB && p.ps;
  const bool lane_ok = isB ? j0 + lane < p.NJ : i0 + lane < p.NI;       // the lane's column exists
  const float* const P = isB ? p.B + j0 : (p.ps ? p.A : p.A + i0);      // uniform
  const long ld = isB ? p.ldb : p.lda;
  const long adv = psA ? 2 * ld : ld;              // floats between consecutive tokens of a row
  const int last_need = isB ? min(RT - 1, r1) : r1 - 1;      // last row of the operand this slice reads
  const int hx = half ? x0 + 64 : x0 - 1;          // the dY waves' halo pixel
  const bool halo_ok = !isB && hx >= 0 && hx < p.Wd;
  float cs = 0.f, sc = 1.f, mxrun = 0.f;
  auto pick = [](float m) __attribute__((always_inline)) { return exp2f(fminf(fmaxf(4.f - floorf(log2f(m)), -120.f), 120.f)); };
  auto load = [&](int rr, Stage& g) __attribute__((always_inline)) {
    if (rr >= 0 && rr <= last_need) {              // uniform
      const long tok = psA ? 4L * rr * p.Wd : (long)rr * p.Wd;            // the row's first pixel
      const float* q = P + (tok + (psA ? 2 : 1) * (x0 + 32 * half)) * ld;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        g.v[t] = lane_ok ? ColVec<1>::ldg(q, colb) : 0.f;
        q += adv;
      }
      g.halo = (halo_ok && lane_ok) ? ColVec<1>::ldg(P + (tok + (psA ? 2 : 1) * hx) * ld, colb) : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < 32; ++t) g.v[t] = 0.f;
      g.halo = 0.f;
    }
  };
  auto stage_max = [&](const Stage& g) __attribute__((always_inline)) {
    float m = fabsf(g.halo);
#pragma unroll
    for (int t = 0; t < 32; ++t) m = fmaxf(m, fabsf(g.v[t]));
    return m;
  };
  // row rr of the operand -> its ring buffer; planes PL apart; this wave's octets 4 half .. 4 half + 3
  auto store = [&](int rr, const Stage& g) __attribute__((always_inline)) {
    mxrun = fmaxf(mxrun, stage_max(g));
    unsigned char* rb = isB ? smem + S9_X0 + ((rr + 4) & 3) * S9_XROW : smem + (rr & 1) * S9_DYROW;
    const int PL = isB ? S9_XPL : S9_DYPL;
    unsigned* const edge = (unsigned*)(smem + (rr & 1) * S9_DYROW + S9_DYE) + (1 + 4 * half) * 64 + lane;   // (dY) octet 4 half
    rb += 2 * half * S9_SUB;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      unsigned qh[4], ql[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float e0 = g.v[8 * o + 2 * t], e1 = g.v[8 * o + 2 * t + 1];
        if (!isB) cs += e0 + e1;
        split2_pair(e0 * sc, e1 * sc, qh[t], ql[t]);
      }
      unsigned char* dst = rb + (o >> 1) * S9_SUB + unit_slot3(lane, o & 1) * 16;
      *(u32x4*)(dst) = u32x4{qh[0], qh[1], qh[2], qh[3]};
      *(u32x4*)(dst + PL) = u32x4{ql[0], ql[1], ql[2], ql[3]};
      if (!isB) {                                  // the octet's edge tokens: first | last << 16
        edge[o * 64] = (qh[0] & 0xffffu) | (qh[3] & 0xffff0000u);
        edge[o * 64 + S9_EDGE / 4] = (ql[0] & 0xffffu) | (ql[3] & 0xffff0000u);
      }
    }
    if (!isB) {                                    // halo pixel: the last token of octet -1 / the first of octet 8
      unsigned hh, hl;
      split2_pair(g.halo * sc, 0.f, hh, hl);
      unsigned* const e = (unsigned*)(smem + (rr & 1) * S9_DYROW + S9_DYE) + (half ? 9 * 64 : 0) + lane;
      e[0] = half ? (hh & 0xffffu) : (hh << 16);
      e[S9_EDGE / 4] = half ? (hl & 0xffffu) : (hl << 16);
    }
  };
  // (No s_setprio here: staging first -- the token-order form's choice -- 601 us, matrix waves first 591, none 580.)

  const int lead = isB ? 2 : 1;                    // step k stores row r0 + k + lead
  float* mxs = (float*)(smem + S9_MAX);
  Stage g0;                      // ONE register stage (a second one spills at the 128 registers of a 16-wave block): a row's
                                 // requests leave behind the previous row's stores and have the rest of the step to arrive
  {
    load(r0, g0);
    mxs[sw * 64 + lane] = pass ? aux[64 + sw * 64 + lane] : stage_max(g0);
    if (lane == 0 && sw == 0) *(volatile int*)(smem + S9_RETRY) = 0;
    __syncthreads();
    {
      const float m2 = fmaxf(mxs[sw * 64 + lane], mxs[(sw ^ 1) * 64 + lane]);      // both halves of the column
      float mw = m2;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
      // pass 0: the first row says little about a SPARSE column (a ReLU channel: 6 % of EDSR's body columns stay zero
      // throughout, others show one barely positive pixel in the first row and ordinary ones later), so no column is taken
      // for more than 2^8 below the tile's maximum on that evidence: what follows may be 100 x larger than the tile's
      // first row or 10^4 x smaller.  (The column's own first-row maximum alone -- the tile's for an all-zero one -- ran
      // nearly every block of EDSR's body convs twice; and ONE flagged block costs the second pass a whole block's time.)
      if (pass) sc = m2 > 0.f ? pick(m2) : 1.f;
      else { const float m = fmaxf(m2, mw * 0x1p-8f); sc = m > 0.f ? pick(m) : 1.f; }
    }
    store(r0, g0);
    if (isB) {
      load(r0 - 1, g0);
      store(r0 - 1, g0);
      load(r0 + 1, g0);
      store(r0 + 1, g0);
    }
    load(r0 + lead, g0);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
      if (DBG != 2 && DBG < 7) { store(r0 + k + lead, g0); load(r0 + k + lead + 1, g0); }
      __syncthreads();
    }
    if (half == 0) ((float*)(smem + S9_SINV))[(isB ? 64 : 0) + lane] = 1.0f / sc;
    mxs[sw * 64 + lane] = mxrun;
    if (colsum && !isB) ((float*)(smem + S9_CS))[half * 64 + lane] = cs;
    __syncthreads();
    if (!pass) {
      // did the guess hold?  By the column's maximum over BOTH halves of the strip's rows (a half whose own pixels are all
      // tiny is below the column's resolution rightly)
      aux[64 + sw * 64 + lane] = mxrun;            // pass 1 takes both halves' maxima from here
      const float mcol = fmaxf(mxrun, mxs[(sw ^ 1) * 64 + lane]);
      const float top = mcol * sc;
      const bool over = __any(top > 60000.f), under = __any(mcol > 0.f && top < 0.125f);
      if ((over || under) && lane == 0) *(volatile int*)(smem + S9_RETRY) = 1;
      if (lane == 0) ((int*)aux)[1 + sw] = (over ? 1 : 0) | (under ? 2 : 0);      // (for tools/t9s_retry_stats.py: which wave, why)
    }
    __syncthreads();
    if (!pass && sw == 0 && lane == 0) *(int*)aux = *(volatile const int*)(smem + S9_RETRY);
  }
  if (colsum) {
    __syncthreads();
    if (sw == 0 && lane_ok) {
      const float* c2 = (const float*)(smem + S9_CS);
      const int io = i0 + lane;
      p.part_colsum[(long)s * p.NI + (p.ps ? (io % ps_f) * 4 + io / ps_f : io)] = c2[lane] + c2[64 + lane];
    }
  }
}

// PASS 0: one block per (slice, tile).  PASS 1: the same grid again; a block whose flag is clear leaves at once (6 us for a
// launch without a flagged block), the others run with exact exponents.
template <int DBG = 0, int PASS = 0>
__global__ void __launch_bounds__(S9_THREADS) k_tnb9s(TnArgs p, int tiles, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  tnb_body9s<DBG>(p, L / tiles, L % tiles, tiles, PASS, smem);
}

template <int DBG = 0, bool F16 = false>
__global__ void __launch_bounds__(512, SR_TNB3_OCC) k_tnb3(TnArgs p, int tiles, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int trow = L % 3, t2 = L / 3;
  tnb_body3<DBG, F16>(p, t2 / tiles, t2 % tiles, trow, smem);
}

// Conv launches are one-dimensional with the block -> (slice, tile, tap) map made here: the 9 taps x i-tiles of one
// slice read the same rows of dY and X, and the hardware deals consecutive block indices to the 8 XCDs round robin --
// with (slice, tile, tap) on the grid axes every XCD's L2 fetched every slice for itself (rocprofv3 PMC: 2.5 GB per
// launch against 0.29 GB algorithmic on the EDSR x8 upsampler problems, 19 GB against 2.2 GB on the x4 body batch,
// i.e. 3.6-6 TB/s of L2 misses: the kernel's bound).  sr_xcd_block gives an XCD a contiguous run of logical indices:
// the blocks that share a slice meet in ONE L2.  SRHIP_TN_XCD=0: the old order.
template <int W>
__global__ void __launch_bounds__(512, 1) k_tnb(TnArgs p, int tiles, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (!p.conv) { tnb_body<W>(p, blockIdx.x, blockIdx.y, 0, smem); return; }
  const int L = xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tap = L % 9, t2 = L / 9;
  tnb_body<W>(p, t2 / tiles, t2 % tiles, tap, smem);
}

// 3x3 conv weight gradient on 64 W-column tiles, two fp16 planes / three products (tnb_body_h<.., CONV>): SwinIR's 180-channel
// convs (W = 3), DRRN's 128-channel ones (W = 2).  Same block -> (slice, tile, tap) map as k_tnb.
template <int W>
__global__ void __launch_bounds__(512, 1) k_tnb_hc(TnArgs p, int tiles, int xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tap = L % 9, t2 = L / 9;
  const bool even = p.NI % W == 0 && p.NJ % W == 0 && p.i_tile % W == 0 && p.j_tile % W == 0;
  if (even) tnb_body_h<W, 0, false, true, true>(p, t2 / tiles, t2 % tiles, smem, tap);
  else tnb_body_h<W, 0, false, false, true>(p, t2 / tiles, t2 % tiles, smem, tap);
}

// Up to TNB_GROUP_MAX Linear problems over the same rows in one launch: the four of a Swin block, or the 4 x depth of a
// whole RSTB layer (SwinIREngine defers them to the layer's end: 48 tiles need 5 reduce slices to fill the chip instead
// of 32 -- a sixth of the partial-sum traffic and of the reducer's work per block).  The block's problem is selected by
// a chain of uniform conditional copies with CONSTANT indices (a runtime-indexed kernel-argument array would be copied
// to scratch), so the body exists once.
constexpr int TNB_GROUP_MAX = 24;
struct TnbGroup {
  TnArgs p[TNB_GROUP_MAX];
  int tile_start[TNB_GROUP_MAX + 1];
  int n, S, xcd;
};
// block -> (slice, tile).  xcd: the tiles of ONE slice (they read the same token rows: the i-tiles of a problem share all
// of X, its j-tiles all of dY) get consecutive logical indices on one XCD, i.e. run side by side on one L2; 0: the old
// order (slice fastest: a slice's tiles are S blocks apart and meet nowhere -- 2.43 GB fetched per launch for 1.7 GB of
// operands on the README net)
__device__ __forceinline__ void tnb_group_block(const TnbGroup& g, int& slice, int& t) {
  const int tiles = g.tile_start[g.n];
  if (g.xcd) {
    const int L = sr_xcd_block(blockIdx.x, gridDim.x);
    t = L % tiles;
    slice = L / tiles;
  } else {
    slice = blockIdx.x % g.S;
    t = blockIdx.x / g.S;
  }
}
template <int W, int DBG = 0>
__global__ void __launch_bounds__(512, 1) k_tnb_grouped(TnbGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int slice, t;
  tnb_group_block(g, slice, t);
  TnArgs p = g.p[0];
  int t0 = 0;
#pragma unroll
  for (int i = 1; i < TNB_GROUP_MAX; ++i)
    if (i < g.n && t >= g.tile_start[i]) { p = g.p[i]; t0 = g.tile_start[i]; }
  tnb_body<W, DBG>(p, slice, t - t0, 0, smem);
}

template <int W>
__global__ void __launch_bounds__(512, 1) k_tnb_grouped_h(TnbGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int slice, t;
  tnb_group_block(g, slice, t);
  TnArgs p = g.p[0];
  int t0 = 0;
#pragma unroll
  for (int i = 1; i < TNB_GROUP_MAX; ++i)
    if (i < g.n && t >= g.tile_start[i]) { p = g.p[i]; t0 = g.tile_start[i]; }
  const bool even = p.NI % W == 0 && p.NJ % W == 0 && p.i_tile % W == 0 && p.j_tile % W == 0;   // no ragged W-tuples
  if (even && p.b_mode == 1 && !p.a_rowscale) tnb_body_h<W, 1, false, true>(p, slice, t - t0, smem);     // LayerNorm-folded Linears
  else if (even && p.b_mode == 0 && p.a_rowscale) tnb_body_h<W, 0, true, true>(p, slice, t - t0, smem); // DropPath-scaled gradients
  else if (even && p.b_mode == 2 && p.a_rowscale) tnb_body_h<W, 2, true, true>(p, slice, t - t0, smem); // fc2: gelu(h) recomputed
  else tnb_body_h<W>(p, slice, t - t0, smem);
}

// Up to TNB_BATCH_MAX conv weight-gradient problems of ONE shape (same image geometry and channel
// counts: the body convs of an EDSR-style net) in one launch.  A single 64 -> 64 problem at 32768
// pixels is 2.4 GFLOP: alone it needs ~85 reduce slices to fill the chip (13 chunks per block, a
// 12.5 MB partial buffer and a 24 us reducer per layer); 33 of them together run 2-3 slices of
// 340+ chunks each.
constexpr int TNB_BATCH_MAX = 40;
struct TnbConvBatch {
  TnArgs base;
  const float* A[TNB_BATCH_MAX];
  const float* B[TNB_BATCH_MAX];
  long part_stride, colsum_stride;     // floats between consecutive problems' partial buffers
  int n, tiles, xcd;
};
template <int W>
__global__ void __launch_bounds__(512, 1) k_tnb_conv_batched(TnbConvBatch g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // one-dimensional grid, XCD-aware (see k_tnb): logical index -> (problem, slice, tile, tap), taps fastest
  const int L = g.xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tap = L % 9;
  int rr = L / 9;
  const int tile = rr % g.tiles; rr /= g.tiles;
  const int sl = rr % g.base.S, k = rr / g.base.S;
  TnArgs p = g.base;
  p.A = g.A[k];
  p.B = g.B[k];
  p.part = g.base.part + (long)k * g.part_stride;
  p.part_colsum = g.base.part_colsum ? g.base.part_colsum + (long)k * g.colsum_stride : nullptr;
  // taps fastest: the nine tap blocks of one (slice, problem) are dispatched together and walk the same
  // rows of dY / X at the same pace (with the taps in grid.z, five ran in the first round of blocks and
  // four re-read everything in the second: EDSR x2, 134 MB per operand, 208 instead of 228 patches/s)
  tnb_body<W>(p, sl, tile, tap, smem);
}

template <int DBG = 0, bool F16 = false>
__global__ void __launch_bounds__(512, SR_TNB3_OCC) k_tnb3_conv_batched(TnbConvBatch g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = g.xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int trow = L % 3;
  int rr = L / 3;
  const int tile = rr % g.tiles; rr /= g.tiles;
  const int sl = rr % g.base.S, k = rr / g.base.S;
  TnArgs p = g.base;
  p.A = g.A[k];
  p.B = g.B[k];
  p.part = g.base.part + (long)k * g.part_stride;
  p.part_colsum = g.base.part_colsum ? g.base.part_colsum + (long)k * g.colsum_stride : nullptr;
  tnb_body3<DBG, F16>(p, sl, tile, trow, smem);
}

__global__ void __launch_bounds__(T9_THREADS) k_tnb9_conv_batched(TnbConvBatch g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = g.xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int rr = L;
  const int tile = rr % g.tiles; rr /= g.tiles;
  const int sl = rr % g.base.S, k = rr / g.base.S;
  TnArgs p = g.base;
  p.A = g.A[k];
  p.B = g.B[k];
  p.part = g.base.part + (long)k * g.part_stride;
  p.part_colsum = g.base.part_colsum ? g.base.part_colsum + (long)k * g.colsum_stride : nullptr;
  tnb_body9<0>(p, sl, tile, smem);
}

template <int PASS>
__global__ void __launch_bounds__(S9_THREADS) k_tnb9s_conv_batched(TnbConvBatch g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto one = [&](int L) __attribute__((always_inline)) {
    int rr = L;
    const int tile = rr % g.tiles; rr /= g.tiles;
    const int sl = rr % g.base.S, k = rr / g.base.S;
    TnArgs p = g.base;
    p.A = g.A[k];
    p.B = g.B[k];
    p.part = g.base.part + (long)k * g.part_stride;
    p.part_colsum = g.base.part_colsum ? g.base.part_colsum + (long)k * g.colsum_stride : nullptr;
    p.aux = g.base.aux + (long)k * g.base.S * g.tiles * S9_AUX;
    tnb_body9s<0>(p, sl, tile, g.tiles, PASS, smem);
  };
  one(g.xcd ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x);
}

// three taps per block (tnb_body3) for this problem?  SRHIP_TN_T3=0: one tap per block
bool tnb_t3_shape(int conv, int NI, int NJ, int w) {
  static const int on = [] { const char* e = sr_getenv("SRHIP_TN_T3"); return !(e && e[0] == '0'); }();
  return on && conv && w == 1 && NI % 64 == 0 && NJ % 64 == 0;
}
bool tnb_t3_ok(const TnArgs& p, int w) {
  return tnb_t3_shape(p.conv, p.NI, p.NJ, w) && !p.a_rowscale && p.b_mode == 0;
}
constexpr int lds_bytes3() { return 2 * 3 * 4 * 64 * 32; }
// all nine taps per block (tnb_body9: two fp16 planes only)?  SRHIP_TN_T9=0 (experiments build): the three-tap form
bool tnb_t9_shape(int conv, int NI, int NJ, int w) {
#ifdef SR_TN_T9_OFF
  return false;                                    // build variant for the same-box A/B (make EXTRA=-DSR_TN_T9_OFF)
#endif
  static const int on = [] { const char* e = sr_getenv("SRHIP_TN_T9"); return !(e && e[0] == '0'); }();
  static const int f16 = [] { const char* e = sr_getenv("SRHIP_TN_F16X2"); return !(e && e[0] == '0'); }();
  return on && f16 && tnb_t3_shape(conv, NI, NJ, w);
}
bool tnb_t9_ok(const TnArgs& p, int w) { return tnb_t9_shape(p.conv, p.NI, p.NJ, w) && tnb_t3_ok(p, w); }
// two fp16 planes / three products in the three-tap kernels (default); SRHIP_TN_F16X2=0: three bf16 planes / six products
bool tnb_f16() {
  static const int on = [] { const char* e = sr_getenv("SRHIP_TN_F16X2"); return !(e && e[0] == '0'); }();
  return on;
}

int pick_tile(int n, int* w) {
  if (n % 180 == 0) { *w = 3; return 180; }
  if (n <= 64) { *w = 1; return 64; }
  if (n <= 128 || n % 128 == 0) { *w = 2; return 128; }
  *w = 3; return 192;
}

// Square block tiles of 64*W columns on both operands.  With unequal operand widths the SMALLER class
// decides (a 256 x 64 problem runs as four 64 x 64 tiles, not as two half-empty 128 x 128 ones).
int pick_w(int NI, int NJ, int* tile) {
  int wi, wj;
  const int ti = pick_tile(NI, &wi), tj = pick_tile(NJ, &wj);
  const int w = wi < wj ? wi : wj;
  *tile = wi == wj ? ti : (w == 3 ? 192 : 64 * w);        // equal classes keep 180-wide tiles for 180 / 360 / 540
  if (wi == wj && ti != tj) *tile = 64 * w;
  return w;
}

// The strip form of the nine-tap block (tnb_body9s): image width a multiple of 64, at least one slice per strip, four rows per
// slice.  Channel counts: every conv of at least 64 channels on either side, in 64-column tiles -- the last one partly empty
// where a count is no multiple of 64.  Against the one-tap-per-block kernels of the wider tiles (launch + reducer, 8 x 64 x 64,
// same box): 180 -> 180 114 -> 90 us, 180 -> 64 107 -> 66, 128 -> 128 85 -> 60 (8 x 128 x 128: 262 -> 142), 256 -> 256
// 247 -> 137, 192 -> 192 107 -> 90, 96 -> 96 75 -> 57.  SRHIP_TN_T9S=0 / SRHIP_TN_T9S_RAGGED=0 (experiments build): off /
// multiples of 64 only.
bool tnb_t9s_shape(int conv, int NI, int NJ) {
#ifdef SR_TN_T9S_OFF
  return false;                                    // build variant for the same-box A/B (make EXTRA=-DSR_TN_T9S_OFF)
#endif
  static const int on = [] { const char* e = sr_getenv("SRHIP_TN_T9S"); return !(e && e[0] == '0'); }();
  static const int ragged = [] { const char* e = sr_getenv("SRHIP_TN_T9S_RAGGED"); return e ? atoi(e) : 1; }();
  if (!on || !tnb_f16() || !conv) return false;
  int t;
  if (tnb_t9_shape(conv, NI, NJ, pick_w(NI, NJ, &t))) return true;
  return ragged && NI >= 64 && NJ >= 64;
}
bool tnb_t9s_ok(const TnArgs& p) {
  return tnb_t9s_shape(p.conv, p.NI, p.NJ) && !p.a_rowscale && p.b_mode == 0 && p.Wd % 64 == 0 && p.S >= p.Wd / 64 &&
         (long)p.batch * p.H * (p.Wd / 64) >= 4L * p.S;      // a slice of fewer than four rows stages more halo rows than rows
}

template <typename K>
int reserve_lds(K kern, int bytes, const char* name) {
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return sr_fail(-5, "%s: cannot reserve %d B of LDS: %s", name, bytes, hipGetErrorString(e));
  return 0;
}

constexpr int lds_bytes(int w) { return 2 * 3 * 2 * 64 * w * 64; }

}  // namespace

int sr_gemm_tnb_grouped(TnArgs* probs, int n, hipStream_t st) {
  SR_REQUIRE(n >= 1 && n <= TNB_GROUP_MAX, "gemm_tn_grouped_bx3: 1..%d problems (got %d)", TNB_GROUP_MAX, n);
  static_assert(sizeof(TnbGroup) <= 4096, "kernel arguments");
  TnbGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  int w = 1, tiles = 0;
  for (int k = 0; k < n; ++k) {
    TnArgs& p = probs[k];
    SR_REQUIRE(p.M == probs[0].M && p.S == probs[0].S && !p.conv,
               "gemm_tn_grouped_bx3: problems must share M and S");
    SR_REQUIRE(p.NI % 4 == 0 && p.NJ % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0,
               "gemm_tn_grouped_bx3: NI, NJ, lda, ldb must be multiples of 4");
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
  g.S = probs[0].S;
  static const int gxcd = [] { const char* e = sr_getenv("SRHIP_TN_GROUP_XCD"); return e ? atoi(e) : 1; }();
  g.xcd = gxcd;
  dim3 grid(probs[0].S * tiles, 1, 1);
  static bool attr[4] = {false, false, false, false};
#define SR_TNB_G(W_)                                                                      \
  if (w == W_) {                                                                          \
    if (!attr[W_]) {                                                                      \
      if (int rc = reserve_lds(k_tnb_grouped<W_>, lds_bytes(W_), "k_tnb_grouped")) return rc; \
      attr[W_] = true;                                                                    \
    }                                                                                     \
    hipLaunchKernelGGL((k_tnb_grouped<W_>), grid, dim3(512), lds_bytes(W_), st, g);       \
  }
#ifdef SRHIP_EXPERIMENTS
  const char* dbg_env = sr_getenv("SRHIP_TN_DBG");
  const int dbg = dbg_env ? atoi(dbg_env) : 0;      // role ablations (results are wrong on purpose)
#else
  constexpr int dbg = 0;
#endif
  // 192-column tiles on two fp16 planes / three products (tnb_body_h); SRHIP_TN_F16X2_LINEAR=0: bf16x3 / six
  static const int f16lin = [] { const char* e = sr_getenv("SRHIP_TN_F16X2_LINEAR"); return e ? atoi(e) : 1; }();
  if (w == 3 && !dbg && f16lin) {
    static bool attr_h = false;
    if (!attr_h) {
      if (int rc = reserve_lds(k_tnb_grouped_h<3>, lds_bytes(3), "k_tnb_grouped_h")) return rc;
      attr_h = true;
    }
    hipLaunchKernelGGL((k_tnb_grouped_h<3>), grid, dim3(512), lds_bytes(3), st, g);
  } else
#ifdef SRHIP_EXPERIMENTS
  if (w == 3 && dbg == 1) { hipLaunchKernelGGL((k_tnb_grouped<3, 1>), grid, dim3(512), lds_bytes(3), st, g); }
  else if (w == 3 && dbg == 2) { hipLaunchKernelGGL((k_tnb_grouped<3, 2>), grid, dim3(512), lds_bytes(3), st, g); }
  else if (w == 3 && dbg == 4) { hipLaunchKernelGGL((k_tnb_grouped<3, 4>), grid, dim3(512), lds_bytes(3), st, g); }
  else if (w == 3 && dbg == 3) { hipLaunchKernelGGL((k_tnb_grouped<3, 3>), grid, dim3(512), lds_bytes(3), st, g); }
  else
#endif
  {
  SR_TNB_G(1) SR_TNB_G(2) SR_TNB_G(3)
  }
#undef SR_TNB_G
  SR_LAUNCH_CHECK("k_tnb_grouped");
  return 0;
}

int sr_gemm_tnb(TnArgs& p, hipStream_t st) {
  SR_REQUIRE(p.M > 0 && p.S > 0, "gemm_tn_bx3: empty problem");
  SR_REQUIRE(p.NI % 4 == 0 && p.NJ % 4 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0,
             "gemm_tn_bx3: NI, NJ, lda, ldb must be multiples of 4");
  int tile;
  const int w = pick_w(p.NI, p.NJ, &tile);
  p.i_tile = p.j_tile = tile;
  SR_REQUIRE(!p.ps || (p.conv && p.NI % 4 == 0 && (p.NI / 4) % w == 0 && p.NI % p.i_tile == 0),
             "conv3x3_wgrad + PixelShuffle(2): Cout/4 = %d must be a multiple of %d (columns per lane), Cout of the tile", p.NI / 4, w);
  const int rps = sr_cdiv(p.M, p.S);
  p.rows_per_slice = (rps + TKB - 1) / TKB * TKB;
  int tiles = sr_cdiv(p.NI, p.i_tile) * sr_cdiv(p.NJ, p.j_tile);
  dim3 grid(p.S, tiles, 1);
  if (p.conv) grid = dim3(p.S * tiles * 9, 1, 1);
  static const int xcd = [] { const char* e = sr_getenv("SRHIP_TN_XCD"); return !(e && e[0] == '0'); }();
  if (tnb_t9s_ok(p) && (!p.ps || p.NI % 64 == 0)) {      // nine taps per block, strip walk, 64-column tiles
    p.i_tile = p.j_tile = 64;
    tiles = sr_cdiv(p.NI, 64) * sr_cdiv(p.NJ, 64);
    p.aux = p.part + (long)p.S * 9 * p.NI * p.NJ;          // (sr_tn_plan_bx3 sized the workspace for it)
    static bool attr9s = false;
    if (!attr9s) {
      if (int rc = reserve_lds(k_tnb9s<0, 0>, S9_LDS, "k_tnb9s")) return rc;
      if (int rc = reserve_lds(k_tnb9s<0, 1>, S9_LDS, "k_tnb9s")) return rc;
      attr9s = true;
    }
#ifdef SRHIP_EXPERIMENTS
    {
      const char* e = sr_getenv("SRHIP_TN_DBG");
      const int dbg = e ? atoi(e) : 0;
#define SR_T9DBG(D_) if (dbg == D_) { reserve_lds(k_tnb9s<D_>, S9_LDS, "k_tnb9s"); hipLaunchKernelGGL(k_tnb9s<D_>, dim3(p.S * tiles), dim3(S9_THREADS), S9_LDS, st, p, tiles, xcd); return 0; }
      SR_T9DBG(1) SR_T9DBG(2) SR_T9DBG(6) SR_T9DBG(7)
#undef SR_T9DBG
    }
#endif
    hipLaunchKernelGGL((k_tnb9s<0, 0>), dim3(p.S * tiles), dim3(S9_THREADS), S9_LDS, st, p, tiles, xcd);
    hipLaunchKernelGGL((k_tnb9s<0, 1>), dim3(p.S * tiles), dim3(S9_THREADS), S9_LDS, st, p, tiles, xcd);
    SR_LAUNCH_CHECK("k_tnb9s");
    return 0;
  }
  if (tnb_t9_ok(p, w)) {       // 64-wide conv problem: all nine taps per block, one block per CU
    static bool attr9 = false;
    if (!attr9) {
      if (int rc = reserve_lds(k_tnb9<0>, T9_LDS, "k_tnb9")) return rc;
      attr9 = true;
    }
#ifdef SRHIP_EXPERIMENTS
    {
      const char* e = sr_getenv("SRHIP_TN_DBG");
      const int dbg = e ? atoi(e) : 0;
#define SR_T9DBG(D_) if (dbg == D_) { reserve_lds(k_tnb9<D_>, T9_LDS, "k_tnb9"); hipLaunchKernelGGL(k_tnb9<D_>, dim3(p.S * tiles), dim3(T9_THREADS), T9_LDS, st, p, tiles, xcd); return 0; }
      SR_T9DBG(1) SR_T9DBG(2) SR_T9DBG(6) SR_T9DBG(7) SR_T9DBG(3) SR_T9DBG(8) SR_T9DBG(9) SR_T9DBG(10) SR_T9DBG(4) SR_T9DBG(5)
#undef SR_T9DBG
    }
#endif
    hipLaunchKernelGGL(k_tnb9<0>, dim3(p.S * tiles), dim3(T9_THREADS), T9_LDS, st, p, tiles, xcd);
    SR_LAUNCH_CHECK("k_tnb9");
    return 0;
  }
  if (tnb_t3_ok(p, w)) {       // 64-wide conv problem: three taps per block
    static bool attr3 = false;
    if (!attr3) {
      if (int rc = reserve_lds(k_tnb3<0>, lds_bytes3(), "k_tnb3")) return rc;
      attr3 = true;
    }
#ifdef SRHIP_EXPERIMENTS
    {   // role ablations (wrong results on purpose; tools/mb_tnb3.py): 1 = consumers skip their MFMAs, 2 = producers neither load nor store
      const char* e = sr_getenv("SRHIP_TN_DBG");
      const int dbg = e ? atoi(e) : 0;
      if (dbg == 1 && tnb_f16()) { hipLaunchKernelGGL((k_tnb3<1, true>), dim3(p.S * tiles * 3), dim3(512), lds_bytes3(), st, p, tiles, xcd); return 0; }
      if (dbg == 2 && tnb_f16()) { hipLaunchKernelGGL((k_tnb3<2, true>), dim3(p.S * tiles * 3), dim3(512), lds_bytes3(), st, p, tiles, xcd); return 0; }
      if (dbg == 5 && tnb_f16()) { hipLaunchKernelGGL((k_tnb3<5, true>), dim3(p.S * tiles * 3), dim3(512), lds_bytes3(), st, p, tiles, xcd); return 0; }
    }
#endif
    if (tnb_f16()) hipLaunchKernelGGL((k_tnb3<0, true>), dim3(p.S * tiles * 3), dim3(512), lds_bytes3(), st, p, tiles, xcd);
    else hipLaunchKernelGGL((k_tnb3<0>), dim3(p.S * tiles * 3), dim3(512), lds_bytes3(), st, p, tiles, xcd);
    SR_LAUNCH_CHECK("k_tnb3");
    return 0;
  }
  // single conv problems with plain operands on 128- and 192-column tiles: two fp16
  // planes / three products (k_tnb_hc); SRHIP_TN_F16X2_CONV3=0: six bf16
  static const int f16c3 = [] { const char* e = sr_getenv("SRHIP_TN_F16X2_CONV3"); return e ? atoi(e) : 1; }();
  // (64-column tiles stay on k_tnb<1>: SwinIR's 180 -> 64 conv 78 us there, 97 us on this body -- the per-lane row offsets
  // weigh more where a lane stages one column)
  if (p.conv && w >= 2 && f16c3 && tnb_f16() && !p.ps && p.b_mode == 0 && !p.a_rowscale &&
      (long)p.M * p.lda < (1L << 30) && (long)p.batch * p.H * p.Wd * p.ldb < (1L << 30)) {
    static bool attr_hc[4] = {false, false, false, false};
#define SR_TNB_HC(W_)                                                                       \
    if (w == W_) {                                                                          \
      if (!attr_hc[W_]) {                                                                   \
        if (int rc = reserve_lds(k_tnb_hc<W_>, lds_bytes(W_), "k_tnb_hc")) return rc;       \
        attr_hc[W_] = true;                                                                 \
      }                                                                                     \
      hipLaunchKernelGGL((k_tnb_hc<W_>), grid, dim3(512), lds_bytes(W_), st, p, tiles, xcd); \
    }
    SR_TNB_HC(2) SR_TNB_HC(3)
#undef SR_TNB_HC
    SR_LAUNCH_CHECK("k_tnb_hc");
    return 0;
  }
  static bool attr[4] = {false, false, false, false};
#define SR_TNB(W_)                                                                        \
  if (w == W_) {                                                                          \
    if (!attr[W_]) {                                                                      \
      if (int rc = reserve_lds(k_tnb<W_>, lds_bytes(W_), "k_tnb")) return rc;             \
      attr[W_] = true;                                                                    \
    }                                                                                     \
    hipLaunchKernelGGL((k_tnb<W_>), grid, dim3(512), lds_bytes(W_), st, p, tiles, xcd);   \
  }
  SR_TNB(1) SR_TNB(2) SR_TNB(3)
#undef SR_TNB
  SR_LAUNCH_CHECK("k_tnb");
  return 0;
}

int sr_conv_wgrad_batched_plan(int n, int M, int NI, int NJ, int* S, long* part_floats_per_item) {
  int tile;
  const int w = pick_w(NI, NJ, &tile);
  const bool t3 = tnb_t3_shape(1, NI, NJ, w);             // three taps per block
  const bool t9s = tnb_t9s_shape(1, NI, NJ);
  const bool t9 = t9s || tnb_t9_shape(1, NI, NJ, w);      // nine: one block per (tile, slice), one block per CU
  if (t9s) tile = 64;
  const long tiles = (long)sr_cdiv(NI, tile) * sr_cdiv(NJ, tile) * (t9 ? 1 : (t3 ? 3 : 9)) * n;
  // Blocks in flight: 3 per CU for 64-wide tiles (49 KB of LDS each), else 1.  The slice count is chosen
  // for WHOLE rounds of blocks -- 33 problems x 9 taps x 3 slices = 891 blocks on 768 slots ran 1.16
  // rounds, i.e. the second round 16 % full (x4: 5.3 ms for what 1.93 rounds do in 3.1) -- among the
  // counts that leave a slice at least 1024 rows; ties go to fewer slices (less partial traffic).
  const long slots = t9 ? 256 : (w == 1 ? 128L * SR_TNB3_OCC : 256);
  long best = 1;
  double best_eff = 0.0;
  for (long s = 1; s <= 64; ++s) {
    if (s > 1 && M / s < 1024) break;
    const long blocks = tiles * s;
    const double eff = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  *S = (int)best;
  *part_floats_per_item = best * 9 * (long)NI * NJ + (t9 ? best * (tiles / n) * S9_AUX : 0);      // + the strip form's per-block words
  return 0;
}

int sr_conv_wgrad_batched_tnb(const TnArgs& base, const float* const* A, const float* const* B, int n,
                              long part_stride, long colsum_stride, hipStream_t st) {
  SR_REQUIRE(n >= 1 && n <= TNB_BATCH_MAX, "conv3x3_wgrad_batched: 1..%d problems (got %d)", TNB_BATCH_MAX, n);
  SR_REQUIRE(base.conv && base.M > 0 && base.S > 0, "conv3x3_wgrad_batched: conv problems only");
  SR_REQUIRE(base.NI % 4 == 0 && base.NJ % 4 == 0 && base.lda % 4 == 0 && base.ldb % 4 == 0,
             "conv3x3_wgrad_batched: Cout, Cin and the pixel pitches must be multiples of 4");
  TnbConvBatch g;
  memset(&g, 0, sizeof(g));
  g.base = base;
  int tile;
  const int w = pick_w(base.NI, base.NJ, &tile);
  g.base.i_tile = g.base.j_tile = tile;
  const int rps = sr_cdiv(base.M, base.S);
  g.base.rows_per_slice = (rps + TKB - 1) / TKB * TKB;
  g.n = n;
  g.tiles = sr_cdiv(base.NI, tile) * sr_cdiv(base.NJ, tile);
  g.part_stride = part_stride;
  g.colsum_stride = colsum_stride;
  for (int k = 0; k < n; ++k) { g.A[k] = A[k]; g.B[k] = B[k]; }
  static const int xcd = [] { const char* e = sr_getenv("SRHIP_TN_XCD"); return !(e && e[0] == '0'); }();
  g.xcd = xcd;
  if (tnb_t9s_ok(g.base)) {    // all nine taps per block, strip walk
    g.base.i_tile = g.base.j_tile = 64;
    g.tiles = sr_cdiv(base.NI, 64) * sr_cdiv(base.NJ, 64);
    g.base.aux = base.part + (long)n * part_stride;        // behind the n problems' partial sums (the plan sized it)
    static bool attr9s = false;
    if (!attr9s) {
      if (int rc = reserve_lds(k_tnb9s_conv_batched<0>, S9_LDS, "k_tnb9s_conv_batched")) return rc;
      if (int rc = reserve_lds(k_tnb9s_conv_batched<1>, S9_LDS, "k_tnb9s_conv_batched")) return rc;
      attr9s = true;
    }
    hipLaunchKernelGGL(k_tnb9s_conv_batched<0>, dim3(base.S * g.tiles * n), dim3(S9_THREADS), S9_LDS, st, g);
    hipLaunchKernelGGL(k_tnb9s_conv_batched<1>, dim3(base.S * g.tiles * n), dim3(S9_THREADS), S9_LDS, st, g);
    SR_LAUNCH_CHECK("k_tnb9s_conv_batched");
    return 0;
  }
  if (tnb_t9_ok(g.base, w)) {  // all nine taps per block
    static bool attr9 = false;
    if (!attr9) {
      if (int rc = reserve_lds(k_tnb9_conv_batched, T9_LDS, "k_tnb9_conv_batched")) return rc;
      attr9 = true;
    }
    hipLaunchKernelGGL(k_tnb9_conv_batched, dim3(base.S * g.tiles * n), dim3(T9_THREADS), T9_LDS, st, g);
    SR_LAUNCH_CHECK("k_tnb9_conv_batched");
    return 0;
  }
  if (tnb_t3_ok(g.base, w)) {  // three taps per block
    static bool attr3 = false;
    if (!attr3) {
      if (int rc = reserve_lds(k_tnb3_conv_batched<0>, lds_bytes3(), "k_tnb3_conv_batched")) return rc;
      attr3 = true;
    }
    if (tnb_f16()) hipLaunchKernelGGL((k_tnb3_conv_batched<0, true>), dim3(base.S * 3 * g.tiles * n), dim3(512), lds_bytes3(), st, g);
    else hipLaunchKernelGGL((k_tnb3_conv_batched<0>), dim3(base.S * 3 * g.tiles * n), dim3(512), lds_bytes3(), st, g);
    SR_LAUNCH_CHECK("k_tnb3_conv_batched");
    return 0;
  }
  dim3 grid(base.S * 9 * g.tiles * n, 1, 1);
  static bool attr[4] = {false, false, false, false};
#define SR_TNB_CB(W_)                                                                          \
  if (w == W_) {                                                                               \
    if (!attr[W_]) {                                                                           \
      if (int rc = reserve_lds(k_tnb_conv_batched<W_>, lds_bytes(W_), "k_tnb_conv_batched")) return rc; \
      attr[W_] = true;                                                                         \
    }                                                                                          \
    hipLaunchKernelGGL((k_tnb_conv_batched<W_>), grid, dim3(512), lds_bytes(W_), st, g);       \
  }
  SR_TNB_CB(1) SR_TNB_CB(2) SR_TNB_CB(3)
#undef SR_TNB_CB
  SR_LAUNCH_CHECK("k_tnb_conv_batched");
  return 0;
}

// Slice count of the single-problem bx3 launch: one 8-wave block per CU for the 128- and 192-wide tiles
// (98 / 147 KB of LDS each), THREE per CU for 64-wide tiles (49 KB) -- at 64 channels the launch was
// short of blocks with the per-CU budget of the wide tiles (28 slices x 9 taps: 212 us; 85 x 9: 147 us).
int sr_tn_plan_bx3(int M, int NI, int NJ, int conv, int* S, long* part_floats) {
  int tile;
  const int w = pick_w(NI, NJ, &tile);
  const bool t3 = tnb_t3_shape(conv, NI, NJ, w);          // three taps per block
  const bool t9s = tnb_t9s_shape(conv, NI, NJ);           // nine taps per block in 64-column tiles, one block per CU
  const bool t9 = t9s || tnb_t9_shape(conv, NI, NJ, w);   // (the slice count of a shape is the strip form's also where an image
  if (t9s) tile = 64;                                     //  width sends the call to another kernel: the plan does not see it)
  const long tiles = (long)sr_cdiv(NI, tile) * sr_cdiv(NJ, tile) * (conv ? (t9 ? 1 : (t3 ? 3 : 9)) : 1);
  static const long t1 = [] { const char* e = sr_getenv("SRHIP_TNB_BLOCKS_W1"); return e ? atol(e) : 768L; }();
  static const long t3b = [] { const char* e = sr_getenv("SRHIP_TNB_BLOCKS_T3"); return e ? atol(e) : 768L; }();
  long s = (t9 ? 256 : (t3 ? t3b : (w == 1 ? t1 : 256))) / tiles;
  const long smax = (M + 127) / 128;
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  *S = (int)s;
  *part_floats = s * (conv ? 9 : 1) * (long)NI * NJ;
  if (t9) *part_floats += s * tiles * S9_AUX;              // the strip form's per-block words (tnb_body9s)
  return 0;
}
