// NT GEMM on the 3-way bf16 split MFMA, 16-wide K stages (operands, prologues and
// epilogues of gemm_ntb.hip; 64-row tiles, 4 waves 2 x 2, wave tile 32 x 32*WN, two
// blocks per CU):
//
//   C[M,N] = epi( pro(A)[M,K] . W[N,K]^T )      W pre-split into bf16 planes
//
// Why 16-wide stages: the K loop of gemm_ntb.hip is co-bound by the W stream through the
// CU's L1 (64 B/clk).  Its 32-wide W chunk is 36.9 KB -- more than the 32 KB L1 -- so the
// two blocks that share a CU each pull the chunk from L2 (s_memtime: the 9 W loads of a
// wave take ~1100 cycles to issue).  A 16-wide stage is 18.4 KB: the second block's
// loads hit in L1.  Two LDS stage buffers and ONE barrier per stage keep the barrier
// count per k where it was; while the 18 MFMAs of stage c run, the wave splits and
// stores stage c+1 into the other buffer and issues the loads of stage c+3.
//
// LDS per stage: 3 planes x (64 A rows + 64*WN W rows) x 32 B; a row holds two 16-byte
// units (k 0-7, 8-15), unit (row, u) at slot 2*row + (u ^ ((row>>3)&1)): the 16 lanes
// of a ds_read_b128 phase (16 consecutive rows, same u) hit 16 distinct slots.
// Registers: two stages of A (f32) and W (bf16 planes) in flight.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "nt_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SK = 16;           // k per stage = one k step of the 32x32x16 MFMA
constexpr int BM = 64;

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int unit_slot(int row, int u) { return 2 * row + (u ^ ((row >> 3) & 1)); }

constexpr int ntp_lds(int wn) {
  const int stages = 2 * 3 * (BM + 64 * wn) * 32;
  const int wide = 4 * 32 * (32 * wn + 8) * 4 + 2 * 2 * 64 * 4;   // nt_epilogue_wide's transposition tiles + row-stat exchange
  return stages > wide ? stages : wide;
}

// AMP = true: reduced-precision inference (one bf16 product of the leading planes; see gemm_ntb.hip)
template <int WN, bool AMP = false>
__global__ void __launch_bounds__(256, 2) k_ntp(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 3;
  constexpr int BN = 64 * WN;
  constexpr int B_N = NPL * BN * 2;              // 16-byte W units per stage (3 planes; AMP: the leading one)
  constexpr int B_IT = (B_N + 255) / 256;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
  constexpr int A_STAGE = 3 * A_PLANE, B_STAGE = 3 * B_PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;                 // [2][A_STAGE]
  unsigned char* const Bs = smem + 2 * A_STAGE;   // [2][B_STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int nst = (p.K + SK - 1) / SK;            // stages
  const int kp_stages = p.Kp / SK;                // W planes are zero padded up to Kp

  // ---- staging invariants: thread = one float4 of A (row tid>>2, k 4*(tid&3)) and up
  //      to B_IT 16-byte W units; rows outside the problem are clamped (never stored)
  const int arow = tid >> 2, ac4 = tid & 3;
  const int agm = min(m0 + arow, p.M - 1);
  const unsigned offA = (unsigned)(agm * (int)p.lda + ac4 * 4) * 4u;
  const float2 rst = ldg_f2(p.a_mode == 1 ? p.ln_stats + 2 * agm : k_sr_neutral);
  const int a_dst = unit_slot(arow, ac4 >> 1) * 16 + (ac4 & 1) * 8;
  unsigned offB[B_IT];
  int b_dst[B_IT];
  const long plane_bytes = (long)p.N * p.Kp * 2;
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = min(tid + it * 256, B_N - 1);
    const int pl = idx / (BN * 2), rem = idx - pl * (BN * 2);
    const int row = rem >> 1, u = rem & 1;
    // planes [Kp/16][N][16]: the stage's rows are one contiguous run
    offB[it] = (unsigned)(pl * plane_bytes + (long)(n0 + min(row, nvalid - 1)) * 32 + u * 16);
    b_dst[it] = pl * B_PLANE + unit_slot(row, u) * 16;
  }

  // loaded values stay untouched until staging (no early waits); stages past the end
  // read a valid address (k = 0 of the row / the last W stage) and are never consumed
  auto load_a = [&](int c) -> f32x4 {
    const int k = c * SK + ac4 * 4;
    const bool oob = k >= p.K;
    return *(const f32x4*)((const char*)p.A + (oob ? offA - ac4 * 16u : offA + (unsigned)c * (SK * 4)));
  };
  auto load_b = [&](int c, u32x4 (&rb)[B_IT]) {
    const char* base = (const char*)p.Wb + (long)min(c, kp_stages - 1) * p.N * 32;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const u32x4*)(base + offB[it]);
  };
  auto store = [&](int c, f32x4 v, const u32x4 (&rb)[B_IT]) {
    unsigned char* sa = As + (c & 1) * A_STAGE;
    unsigned char* sb = Bs + (c & 1) * B_STAGE;
    if (p.a_mode == 1) {
      v.x = (v.x - rst.x) * rst.y; v.y = (v.y - rst.x) * rst.y; v.z = (v.z - rst.x) * rst.y; v.w = (v.w - rst.x) * rst.y;   // scalar on purpose (common.h)
    } else if (p.a_mode == 2) {
      v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
    }
    if (c * SK + ac4 * 4 >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};      // K tail: exact zeros
    unsigned h0, m0_, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0_, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *(u32x2*)(sa + a_dst) = u32x2{h0, h1};
    if (!AMP) {
      *(u32x2*)(sa + A_PLANE + a_dst) = u32x2{m0_, m1};
      *(u32x2*)(sa + 2 * A_PLANE + a_dst) = u32x2{l0, l1};
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (B_N % 256 == 0 || tid + it * 256 < B_N) *(u32x4*)(sb + b_dst[it]) = rb[it];
  };

  f32x16 acc[1][WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[0][j][q] = 0.f;
  const int a_off = unit_slot(wm * 32 + r, h) * 16;
  int b_off[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) b_off[j] = unit_slot((wn * WN + j) * 32 + r, h) * 16;

  auto mma = [&](int c) {
    const unsigned char* sa = As + (c & 1) * A_STAGE;
    const unsigned char* sb = Bs + (c & 1) * B_STAGE;
    u32x4 fa[3], fb[WN][3];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * A_PLANE + a_off);
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *(const u32x4*)(sb + pl * B_PLANE + b_off[j]);
    // small terms first; term-outer so that consecutive MFMAs hit different tiles
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < WN; ++j) acc[0][j] = mfma_bf(fa[PA], fb[j][PB], acc[0][j]);
    if constexpr (AMP) {
      SR_TERM(0, 0)
    } else {
      SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
    }
#undef SR_TERM
  };

  // register sets by stage parity: set P holds stage c (c & 1 == P), loaded two stages ahead
  f32x4 ra0, ra1;
  u32x4 rb0[B_IT], rb1[B_IT];
  ra0 = load_a(0); load_b(0, rb0);
  ra1 = load_a(1); load_b(1, rb1);
  store(0, ra0, rb0);
  ra0 = load_a(2); load_b(2, rb0);
  __syncthreads();
  // timing build (SRHIP_NT_DBG bit 64): s_memtime stamps of the even-stage half of the loop
  const bool stamp = (p.dbg & 64) != 0;
  long tk_store = 0, tk_load = 0, tk_mma = 0, tk_bar = 0;
  for (int c = 0; c < nst; c += 2) {
    // stage c (even) is in LDS buffer 0
    const long s0 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
    store(c + 1, ra1, rb1);
    const long s1 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
    ra1 = load_a(c + 3); load_b(c + 3, rb1);
    const long s2 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
    mma(c);
    const long s3 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    if (stamp) { tk_store += s1 - s0; tk_load += s2 - s1; tk_mma += s3 - s2; tk_bar += (long)__builtin_amdgcn_s_memtime() - s3; }
    if (c + 1 < nst) {                       // block-uniform
      store(c + 2, ra0, rb0);
      ra0 = load_a(c + 4); load_b(c + 4, rb0);
      mma(c + 1);
      __syncthreads();
    }
  }

  if (stamp) {       // cycles per phase of wave 0 of two blocks -> C[0..15] (output is lost)
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3) && blockIdx.y == 0 && tid == 0) {
      float* o = p.C + (blockIdx.x == 0 ? 0 : 8);
      const float n = (float)((nst + 1) / 2);
      o[0] = 0.f; o[1] = (float)tk_bar / n; o[2] = (float)tk_store / n; o[3] = 0.f;
      o[4] = (float)tk_load / n; o[5] = (float)tk_mma / n; o[6] = 1.f; o[7] = 0.f;
    }
    return;
  }
  if (p.epi == 5) {
    nt_epilogue_lnbwd<WN>(p, acc, lane, wm, wn, m0, nvalid, (float*)smem);
    return;
  }
  if (p.wide_epi) {                          // block-uniform (set by the dispatcher)
    nt_epilogue_wide<WN>(p, acc, lane, wave, wm, wn, n0, nvalid, m0, (float*)smem);
    return;
  }
  nt_epilogue<1, WN, false>(p, acc, lane, wm, wn, n0, nvalid, m0, 0, 0, 0);
  if (p.stats_out) nt_row_stats<WN>(p, acc, lane, wm, wn, m0, nvalid, (float*)smem);
}

template <int WN>
int launch_ntp(const NtArgs& p, hipStream_t st) {
  constexpr int LDS = ntp_lds(WN);
  dim3 grid(sr_cdiv(p.M, BM), sr_cdiv(p.N, p.n_tile));
  if (p.amp) hipLaunchKernelGGL((k_ntp<WN, true>), grid, dim3(256), LDS, st, p);
  else hipLaunchKernelGGL((k_ntp<WN>), grid, dim3(256), LDS, st, p);
  SR_LAUNCH_CHECK("k_ntp");
  return 0;
}

}  // namespace

// Caller (gemm_ntb.hip) has validated the operands and set Kp / epi / wide_epi; tile width chosen here.
int sr_gemm_ntp(NtArgs& p, hipStream_t st) {
  // 192-column tiles of the f32-accurate path: W fragments straight from global memory (gemm_ntw.hip); SRHIP_NTW=0: this file's kernel
  static const bool ntw = [] { const char* e = sr_getenv("SRHIP_NTW"); return !(e && e[0] == '0'); }();
  const bool w = (ntw && !p.dbg) || p.wfmt == 1;
  if (p.N % 180 == 0) { p.n_tile = 180; return w ? sr_gemm_ntw(p, st) : launch_ntp<3>(p, st); }
  if (p.N <= 64) { p.n_tile = 64; return launch_ntp<1>(p, st); }
  if (p.N <= 128 || p.N % 128 == 0) { p.n_tile = 128; return launch_ntp<2>(p, st); }
  p.n_tile = 192;
  return w ? sr_gemm_ntw(p, st) : launch_ntp<3>(p, st);
}
