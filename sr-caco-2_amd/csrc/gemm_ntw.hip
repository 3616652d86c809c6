// NT GEMM on the 3-way bf16 split MFMA with the WEIGHT FRAGMENTS STRAIGHT FROM GLOBAL MEMORY
// (192-column tiles: every Linear of a Swin block; operands, prologues and epilogues of gemm_ntp.hip):
//
//   C[M,N] = epi( pro(A)[M,K] . W[N,K]^T )      W pre-split into bf16 planes [Kp/16][N][16]
//
// Why: the K loop of gemm_ntp.hip is bound by the LDS pipe -- per 16-wide stage a block moves 18 KB of W
// through VGPR -> LDS (13 cycles per ds_write_b128) and every wave reads 9 KB of it back; with the W stores
// removed the kernel ran 10-17 % faster.  But the plane layout already IS the MFMA
// operand layout: row n's 16 k values of a sub-chunk are 32 contiguous bytes, so lane (c, g) of
// v_mfma_f32_16x16x32_bf16 (column c of a 16-column tile, k = 8g..8g+7) finds its B fragment as ONE
// 16-byte global load, and a wave's load is two contiguous 512-byte runs.  So:
//   * the four waves split the 192 columns (48 each = 3 column tiles of 16), every wave covers all 64
//     rows (4 row tiles): no two waves load the same W bytes, W never touches LDS;
//   * only A goes through LDS (f32 -> three bf16 planes, as before): 12 KB per 32-wide stage instead of
//     48 KB, and a wave reads 12 fragments per 72 MFMAs instead of 24: LDS traffic per k drops to ~40 %;
//   * one barrier per 32 k (72 MFMAs per wave) instead of two.
// Two W register sets (stage parity): the fragments of stage c+2 are requested as soon as the MFMAs of
// stage c have been issued.
// The accumulators (16x16 tiles, column-split waves) are re-laid through LDS once per block into the
// 32x32 / 2x2-wave layout of nt_epi.h, so every epilogue (bias, residual + DropPath, gelu', LayerNorm
// backward, output-row statistics) is the one gemm_ntp.hip runs, bit for bit.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "nt_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SK = 32;            // k per stage = one k step of the 16x16x32 MFMA
constexpr int BM = 64;
constexpr int A_PLANE = BM * 64;  // 64 rows x 32 k bf16
constexpr int A_STAGE = 3 * A_PLANE;
constexpr int TP = 196;           // pitch of the re-layout tile (floats): 4 rows apart = 16 banks apart
constexpr int NTW_LDS = 4 * 32 * (96 + 8) * 4 + 2 * 2 * 64 * 4;   // nt_epilogue_wide's tiles + row-stat exchange (> 2 A stages, > 64 x TP)

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                 0, 0, 0);
}
// 16-byte unit (row, k octet u) of a stage plane: the 16 lanes of a ds_read_b128 phase (16 rows, one u) and of
// a ds_write_b128 phase (4 rows x 4 octets) each hit 16 different bank groups
__device__ __forceinline__ int a_slot(int row, int u) { return row * 4 + (u ^ ((row >> 2) & 3)); }

__global__ void __launch_bounds__(256, 2) k_ntw(NtArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int nst = (p.K + SK - 1) / SK;
  const int nsub = p.Kp / 16;                     // W planes are zero padded up to Kp (a multiple of 32)

  // ---- A staging: thread = 8 consecutive k of one row (two float4); rows outside the problem are clamped
  const int arow = tid >> 2, akq = tid & 3;
  const int agm = min(m0 + arow, p.M - 1);
  const char* const abase = (const char*)p.A + (long)agm * p.lda * 4;
  const float2 rst = ldg_f2(p.a_mode == 1 ? p.ln_stats + 2 * agm : k_sr_neutral);
  const int a_dst = a_slot(arow, akq) * 16;
  auto load_a = [&](int cs, f32x4 (&v)[2]) {      // stages past the end read k = 0 of the row, never consumed
    const int k = cs * SK + akq * 8;
    v[0] = *(const f32x4*)(abase + (k < p.K ? k * 4 : 0));
    v[1] = *(const f32x4*)(abase + (k + 4 < p.K ? (k + 4) * 4 : 0));
  };
  auto store_a = [&](int cs, f32x4 (&v)[2]) {
    unsigned char* sa = smem + (cs & 1) * A_STAGE + a_dst;
    const int k = cs * SK + akq * 8;
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 x = v[e];
      if (p.a_mode == 1) {
        x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y;
      } else if (p.a_mode == 2) {
        x.x = gelu_f(x.x); x.y = gelu_f(x.y); x.z = gelu_f(x.z); x.w = gelu_f(x.w);
      }
      if (k + 4 * e >= p.K) x = f32x4{0.f, 0.f, 0.f, 0.f};          // K tail: exact zeros
      split3_pair(x.x, x.y, hh[2 * e], mm[2 * e], ll[2 * e]);
      split3_pair(x.z, x.w, hh[2 * e + 1], mm[2 * e + 1], ll[2 * e + 1]);
    }
    *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
    *(u32x4*)(sa + A_PLANE) = u32x4{mm[0], mm[1], mm[2], mm[3]};
    *(u32x4*)(sa + 2 * A_PLANE) = u32x4{ll[0], ll[1], ll[2], ll[3]};
  };

  // ---- W fragments: lane (c, g) of column tile jt reads 16 bytes (g & 1) of row n in sub-chunk 2*stage + (g >> 1);
  //      columns past the tile's width re-read its last valid row (their results are never stored)
  const long plane_bytes = (long)p.N * p.Kp * 2;
  unsigned boff[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
    boff[jt] = (unsigned)((n0 + min(wave * 48 + jt * 16 + c, nvalid - 1)) * 32 + (g & 1) * 16);
  auto load_b = [&](int cs, u32x4 (&fb)[3][3]) {
    const char* base = (const char*)p.Wb + (long)min(2 * cs + (g >> 1), nsub - 1) * p.N * 32;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = a_slot(16 * i + c, g) * 16;

  auto mma = [&](int cs, const u32x4 (&fb)[3][3]) {
    const unsigned char* sa = smem + (cs & 1) * A_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[pl] = *(const u32x4*)(sa + pl * A_PLANE + a_off[i]);
      // the six cross products >= 2^-24, small terms first; term-outer: consecutive MFMAs hit different tiles
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fa[PA], fb[j][PB], acc[i][j]);
      SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
    }
  };

  // register sets by stage parity
  f32x4 ra0[2], ra1[2];
  u32x4 fb0[3][3], fb1[3][3];
  load_a(0, ra0); load_b(0, fb0);
  load_a(1, ra1); load_b(1, fb1);
  store_a(0, ra0);
  load_a(2, ra0);
  __syncthreads();
  for (int cs = 0; cs < nst; cs += 2) {
    store_a(cs + 1, ra1);
    load_a(cs + 3, ra1);
    mma(cs, fb0);
    load_b(cs + 2, fb0);
    __syncthreads();
    if (cs + 1 < nst) {                       // block-uniform
      store_a(cs + 2, ra0);
      load_a(cs + 4, ra0);
      mma(cs + 1, fb1);
      load_b(cs + 3, fb1);
      __syncthreads();
    }
  }

  // ---- re-layout: 16x16 tiles of column-split waves -> 32x32 tiles of the 2 x 2 wave grid of nt_epi.h
  float* const T = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc[i][j][e];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31;
  f32x16 acc2[1][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[0][j][q] = T[(wm * 32 + mfma_row(q, lane)) * TP + (wn * 3 + j) * 32 + r];
  __syncthreads();

  if (p.epi == 5) {
    nt_epilogue_lnbwd<3>(p, acc2, lane, wm, wn, m0, nvalid, (float*)smem);
    return;
  }
  if (p.wide_epi) {                          // block-uniform (set by the dispatcher)
    nt_epilogue_wide<3>(p, acc2, lane, wave, wm, wn, n0, nvalid, m0, (float*)smem);
    return;
  }
  nt_epilogue<1, 3, false>(p, acc2, lane, wm, wn, n0, nvalid, m0, 0, 0, 0);
  if (p.stats_out) nt_row_stats<3>(p, acc2, lane, wm, wn, m0, nvalid, (float*)smem);
}

}  // namespace

// 192-column tiles of the f32-accurate path (gemm_ntp.hip decides): n_tile is set by the caller.
int sr_gemm_ntw(NtArgs& p, hipStream_t st) {
  static_assert(NTW_LDS >= 2 * A_STAGE && NTW_LDS >= BM * TP * 4, "LDS regions");
  dim3 grid(sr_cdiv(p.M, BM), sr_cdiv(p.N, p.n_tile));
  hipLaunchKernelGGL(k_ntw, grid, dim3(256), NTW_LDS, st, p);
  SR_LAUNCH_CHECK("k_ntw");
  return 0;
}
