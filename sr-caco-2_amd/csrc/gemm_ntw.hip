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
// Three W register sets (stage mod 3): the fragments of stage c+3 are requested as soon as the MFMAs of
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

// DBG = true (SRHIP_NTW_DBG=bits, timing experiments only, results are wrong): 1 = W fragments always from stage 0
// (cache-hot), 2 = no MFMAs, 4 = no epilogue, 8 = no A staging stores, 16 = three products instead of six
// AMP = true: reduced-precision inference (one bf16 product of the leading planes; see gemm_ntb.hip)
template <bool DBG, bool AMP = false>
__global__ void __launch_bounds__(256, 2) k_ntw(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // One-dimensional grid: the column tiles of a row tile (they read the same A rows) are neighbours in the launch
  // order and, with the XCD-aware order, in one L2 -- the second and third fetch of an A row is an L2 hit instead of
  // another trip to HBM (SRHIP_NTW_GRID=0: row tile fastest, as a two-dimensional grid dealt it)
  const int ncol = (p.N + p.n_tile - 1) / p.n_tile;
  int bt = (int)blockIdx.x, bcol;
  if (p.xcd_order) {
    bt = sr_xcd_block(bt, gridDim.x);
    bcol = bt % ncol; bt /= ncol;
  } else {
    const int nrow = gridDim.x / ncol;
    bcol = bt / nrow; bt -= bcol * nrow;
  }
  const int m0 = bt * BM;
  const int n0 = bcol * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int nst = (p.K + SK - 1) / SK;
  const int nsub = p.Kp / 16;                     // W planes are zero padded up to Kp (a multiple of 32)
  // Every block walks the SAME weight planes: in lockstep all 512 blocks of a launch ask the L2 for the same few
  // cache lines at the same moment.  Block b therefore starts its K walk at stage rot(b) and wraps around (the
  // blocks of one XCD -- same blockIdx.x mod 8 -- get different rotations): at any moment the launch reads all
  // stages of W.  Sums are f32 either way; only their order differs per row block (deterministic).
  const int rot = p.k_rot ? (int)(((unsigned)bt >> 3) % (unsigned)nst) : 0;
  auto stage_of = [&](int cs) {                   // block-uniform; prefetches run up to 5 stages past the end
    int x = cs + rot;
    while (x >= nst) x -= nst;
    return x;
  };

  // ---- A staging: thread = 8 consecutive k of one row (two float4); rows outside the problem are clamped
  const int arow = tid >> 2, akq = tid & 3;
  const int agm = min(m0 + arow, p.M - 1);
  const char* const abase = (const char*)p.A + (long)agm * p.lda * 4;
  const float2 rst = ldg_f2(p.a_mode == 1 ? p.ln_stats + 2 * agm : k_sr_neutral);
  const int a_dst = a_slot(arow, akq) * 16;
  auto load_a = [&](int cs, f32x4 (&v)[2]) {      // stages past the end wrap around, never consumed
    const int k = stage_of(cs) * SK + akq * 8;
    v[0] = *(const f32x4*)(abase + (k < p.K ? k * 4 : 0));
    v[1] = *(const f32x4*)(abase + (k + 4 < p.K ? (k + 4) * 4 : 0));
  };
  auto store_a = [&](int cs, f32x4 (&v)[2]) {
    unsigned char* sa = smem + (cs & 1) * A_STAGE + a_dst;
    const int k = stage_of(cs) * SK + akq * 8;
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
    if (DBG && (p.dbg & 8)) {
      if (hh[0] + mm[1] + ll[2] + hh[3] + mm[0] + ll[1] + hh[2] + mm[3] + ll[0] + hh[1] + mm[2] + ll[3] == 0x12345u) *(unsigned*)sa = 1u;
      return;
    }
    *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
    if (!AMP) {
      *(u32x4*)(sa + A_PLANE) = u32x4{mm[0], mm[1], mm[2], mm[3]};
      *(u32x4*)(sa + 2 * A_PLANE) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };

  // ---- W fragments: lane (c, g) of column tile jt reads 16 bytes (g & 1) of row n in sub-chunk 2*stage + (g >> 1);
  //      columns past the tile's width re-read its last valid row (their results are never stored).  The address is
  //      a block-uniform base (stage, plane: scalar registers) + a 32-bit per-lane offset that never changes.
  const long plane_bytes = (long)p.N * p.Kp * 2;
  unsigned boff[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
    boff[jt] = (unsigned)(((g >> 1) * p.N + n0 + min(wave * 48 + jt * 16 + c, nvalid - 1)) * 32 + (g & 1) * 16);
  auto load_b = [&](int cs, u32x4 (&fb)[3][3]) {
    const int st = (DBG && (p.dbg & 1)) ? 0 : min(stage_of(cs), nst - 1);        // stages past the end re-read the last one
    const char* base = (const char*)p.Wb + (long)(2 * st) * p.N * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
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
    if (DBG && (p.dbg & 2)) {            // keep the fragments alive without the matrix core
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acc[0][j][pl] += __builtin_bit_cast(float, fb[j][pl].x ^ fb[j][pl].w);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[3];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * A_PLANE + a_off[i]);
      // the six cross products >= 2^-24, small terms first; term-outer: consecutive MFMAs hit different tiles
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else if (DBG && (p.dbg & 16)) {      // three of the six products: what a two-plane (fp16-style) split would cost
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      } else {
        SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  // A: two register sets (stage parity), loaded two stages ahead.  W: THREE fragment sets (stage mod 3), each
  // re-requested for stage c+3 as soon as the MFMAs of stage c are issued: two stages of MFMAs (~2 us) cover the
  // fetch -- with two sets (one stage of cover) the K loop ran at fetch latency + MFMA time per stage.
  f32x4 ra0[2], ra1[2];
  u32x4 fb0[3][3], fb1[3][3], fb2[3][3];
  load_a(0, ra0); load_b(0, fb0);
  if (nst > 1) { load_a(1, ra1); load_b(1, fb1); }
  if (nst > 2) load_b(2, fb2);
  store_a(0, ra0);
  if (nst > 2) load_a(2, ra0);
  __syncthreads();
  // one step: stage cs is in LDS buffer cs & 1; RA = the A registers holding stage cs + 1
  // (loads of stages past the end are skipped -- block-uniform branches: with 6 stages at K = 180 the three
  // prefetches past the end were half as many W bytes again through the L1)
#define SR_STEP(CS, RA, FB)                    \
  if ((CS) + 1 < nst) store_a((CS) + 1, RA);   \
  if ((CS) + 3 < nst) load_a((CS) + 3, RA);    \
  mma((CS), FB);                               \
  if ((CS) + 3 < nst) load_b((CS) + 3, FB);    \
  __syncthreads();
  for (int cs = 0; cs < nst; cs += 6) {        // 6 = lcm(A parity, W sets); every guard is block-uniform
    SR_STEP(cs, ra1, fb0)
    if (cs + 1 < nst) { SR_STEP(cs + 1, ra0, fb1) }
    if (cs + 2 < nst) { SR_STEP(cs + 2, ra1, fb2) }
    if (cs + 3 < nst) { SR_STEP(cs + 3, ra0, fb0) }
    if (cs + 4 < nst) { SR_STEP(cs + 4, ra1, fb1) }
    if (cs + 5 < nst) { SR_STEP(cs + 5, ra0, fb2) }
  }
#undef SR_STEP

  if (DBG && (p.dbg & 4)) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 123456.789f) p.C[0] = sacc;
    return;
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

// ---------------------------------------------------------------------------------------------------------
// Weight operands of format 1 (srhip_gemm_nt_f16x2): the same GEMM on TWO fp16 planes per operand and THREE products
// (h.h + h.l + l.h on v_mfma_f32_16x16x32_f16) instead of three bf16 planes and six.  What makes two planes enough is
// a power-of-two scale per ROW of either operand (a block exponent: exact to apply and to undo): x' = x * 2^s with
// max|x'| in [8192, 16384], x' = h + l, h = fp16(x'), l = fp16(x' - h) carries 22 bits relative to the row's largest
// element; emulated against float64 (tools/split_accuracy.py) the results are indistinguishable from a plain f32 matmul.
// W: planes + per-row 2^-s from the preparation kernel (prep.hip kind 3).  A: the row scale is known a priori behind the
// LayerNorm prologue (|xhat| <= sqrt(K)), otherwise it is a running value over 192-k passes (k_nth2 below).  Half the
// MFMAs, a third less LDS, W fetch and fragment registers than k_ntw.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H_APLANE = A_PLANE;                // 64 rows x 32 k fp16
constexpr int H_ASTAGE = 2 * H_APLANE;

__device__ __forceinline__ f32x4 mfma16h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef _Float16 sr_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
  // packed conversions (v_cvt_pk_f16_f32, round to nearest even); the residuals are exact in f32
  const sr_f16x2 hv = __builtin_convertvector(sr_f32x2{x0, x1}, sr_f16x2);
  const float r0 = x0 - (float)hv.x, r1 = x1 - (float)hv.y;
  const sr_f16x2 lv = __builtin_convertvector(sr_f32x2{r0, r1}, sr_f16x2);
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}

// ---------------------------------------------------------------------------------------------------------
// k_nth2.  The block takes its A rows in PASSES of 192 k: a pass's 12 float4 per thread are
// loaded in one go, the row maximum of the pass comes out of the registers that hold them (four threads share a row: two
// shuffles), the row's scale is the SMALLER of the scale so far and what this pass needs (scales only go down, so nothing
// already accumulated can overflow), the accumulators of the rows whose scale dropped are multiplied by the exact power of
// two that separates the old scale from the new one, and the pass is split into six stage images (two planes: 48 KB, the
// epilogue's tiles reuse the space) that the six stages read without a barrier in between.  K = 180 is one pass.
constexpr int NTH2_LDS = NTW_LDS + 1024;         // + [64] current 2^s, [64] rescale factor of the pass, [64] 2^-s

// AMP = true (srhip_set_matmul_mode(1), inference): the leading planes only, one product
template <bool AMP>
__global__ void __launch_bounds__(256, 2) k_nth2(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const rscale = (float*)(smem + NTW_LDS);        // [64] rescale factor of this pass (1 or 2^-d)
  float* const rinv = rscale + 64;                       // [64] 2^-s, final
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int ncol = (p.N + p.n_tile - 1) / p.n_tile;
  int bt = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int bcol = bt % ncol; bt /= ncol;
  const int m0 = bt * BM;
  const int n0 = bcol * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int nst = (p.K + SK - 1) / SK;

  const int arow = tid >> 2, akq = tid & 3;
  const int agm = min(m0 + arow, p.M - 1);
  const char* const abase = (const char*)p.A + (long)agm * p.lda * 4;
  const float2 rst = ldg_f2(p.a_mode == 1 ? p.ln_stats + 2 * agm : k_sr_neutral);
  const int a_dst = a_slot(arow, akq) * 16;

  const long plane_bytes = (long)p.N * p.Kp * 2;
  const float* const winv_all = (const float*)((const char*)p.Wb + 2 * plane_bytes);
  unsigned boff[3];
  float winv[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    const int col = n0 + min(wave * 48 + jt * 16 + c, nvalid - 1);
    boff[jt] = (unsigned)(((g >> 1) * p.N + col) * 32 + (g & 1) * 16);
    winv[jt] = winv_all[col];
  }
  auto load_b = [&](int cs, u32x4 (&fb)[3][2]) {
    const char* base = (const char*)p.Wb + (long)(2 * cs) * p.N * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };
  u32x4 fb0[3][2], fb1[3][2], fb2[3][2];
  load_b(0, fb0);
  if (nst > 1) load_b(1, fb1);
  if (nst > 2) load_b(2, fb2);

  f32x4 acc[4][3];
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = a_slot(16 * i + c, g) * 16;
  auto mma = [&](int s6, const u32x4 (&fb)[3][2]) {
    const unsigned char* sa = smem + s6 * H_ASTAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * H_APLANE + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16h(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  // a priori scale behind the LayerNorm prologue (|xhat| <= sqrt(K)); otherwise start from "no limit yet"
  const float apriori = exp2f(floorf(log2f(16384.f * rsqrtf((float)p.K))));
  float asc = p.a_mode == 1 ? apriori : 3.0e38f;          // this row's current 2^s (the same in the row's four threads)
  const int npass = (nst + 5) / 6;
  for (int pass = 0; pass < npass; ++pass) {
    const int cs0 = pass * 6;
    if (pass) __syncthreads();                            // every wave is done with the previous pass's images
    {
      f32x4 ra[6][2];
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {                    // stages past the end read k = 0 of the row, zeroed below
        const int k = (cs0 + s6) * SK + akq * 8;
        ra[s6][0] = *(const f32x4*)(abase + (k < p.K ? k * 4 : 0));
        ra[s6][1] = *(const f32x4*)(abase + (k + 4 < p.K ? (k + 4) * 4 : 0));
      }
      float mx = 0.f;
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
        const int k = (cs0 + s6) * SK + akq * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f32x4 x = ra[s6][e];
          if (p.a_mode == 1) {
            x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y;
          } else if (p.a_mode == 2) {
            x.x = gelu_f(x.x); x.y = gelu_f(x.y); x.z = gelu_f(x.z); x.w = gelu_f(x.w);
          }
          if (k + 4 * e >= p.K) x = f32x4{0.f, 0.f, 0.f, 0.f};          // K tail: exact zeros
          ra[s6][e] = x;
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
        }
      }
      float old = asc;
      if (p.a_mode != 1) {
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        const float need = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 3.0e38f;
        asc = fminf(asc, need);
      }
      const float use = asc > 1.0e38f ? 1.f : asc;        // an all-zero row so far: any scale
      if (akq == 0) {
        rscale[arow] = (old > 1.0e38f || old == asc) ? 1.f : asc / old;     // exact power of two <= 1
        rinv[arow] = 1.0f / use;
      }
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x4 x = ra[s6][e];
          split2_pair(x.x * use, x.y * use, hh[2 * e], ll[2 * e]);
          split2_pair(x.z * use, x.w * use, hh[2 * e + 1], ll[2 * e + 1]);
        }
        unsigned char* sa = smem + s6 * H_ASTAGE + a_dst;
        *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
        if (!AMP) *(u32x4*)(sa + H_APLANE) = u32x4{ll[0], ll[1], ll[2], ll[3]};
      }
    }
    __syncthreads();
    if (pass == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (p.a_mode != 1) {                            // rows whose scale dropped: bring what is accumulated to the new scale
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = rscale[16 * i + 4 * g + e];
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j][e] *= f;
        }
    }
#define SR_STAGE(S, FB)                                              \
    if (cs0 + (S) < nst) {                                           \
      mma((S), FB);                                                  \
      if (cs0 + (S) + 3 < nst) load_b(cs0 + (S) + 3, FB);            \
    }
    SR_STAGE(0, fb0) SR_STAGE(1, fb1) SR_STAGE(2, fb2) SR_STAGE(3, fb0) SR_STAGE(4, fb1) SR_STAGE(5, fb2)
#undef SR_STAGE
  }

  __syncthreads();
  float* const T = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ri = rinv[16 * i + 4 * g + e];
#pragma unroll
      for (int j = 0; j < 3; ++j) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc[i][j][e] * (ri * winv[j]);
    }
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
  if (p.wide_epi) {
    nt_epilogue_wide<3>(p, acc2, lane, wave, wm, wn, n0, nvalid, m0, (float*)smem);
    return;
  }
  nt_epilogue<1, 3, false>(p, acc2, lane, wm, wn, n0, nvalid, m0, 0, 0, 0);
  if (p.stats_out) nt_row_stats<3>(p, acc2, lane, wm, wn, m0, nvalid, (float*)smem);
}

// ---------------------------------------------------------------------------------------------------------
// The same idea for the 3x3 conv as implicit GEMM (64-pixel x 192-column tiles: SwinIR's 180 -> 180 convs,
// operands / epilogues of gemm_ntb.hip's k_ntb<1, 3, true>): the halo tile of a 32-channel chunk is split once
// into LDS (6 x 18 pixels, 80-byte pixel pitch: conflict free for the 16-lane fragment phases), and the nine
// taps of the chunk run WITHOUT any barrier -- a tap's W fragments (rows tap*Cout + n of the tap-major pack)
// come straight from global memory into the MFMA operand registers (three fragment sets, one per tap mod 3),
// its A fragments are the halo tile shifted by (dy, dx).  Two barriers per channel chunk instead of one or two
// per (chunk, tap); 18 KB of LDS reads per tap and wave-quad instead of 132 KB of W + A traffic.
constexpr int C_PITCH = 80;                      // bytes per halo pixel and plane (32 bf16 + 16 B pad)
constexpr int C_AROWS = 6 * 18;                  // halo pixels of a 4 x 16 tile
constexpr int C_APLANE = C_AROWS * C_PITCH;
constexpr int C_AN = C_AROWS * 8;                // float4 slots per chunk
constexpr int C_AIT = (C_AN + 255) / 256;
constexpr int NTCW_LDS = BM * TP * 4;            // the re-layout tile (> 3 halo planes)

template <bool AMP>
__global__ void __launch_bounds__(256, 2) k_ntcw(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // neighbouring tiles (shared halos) in one L2
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * 4, x0 = tx * 16;
  const int nkc = (p.K + 31) / 32;

  // ---- halo staging: thread = up to C_AIT float4 (pixel, 4 channels) of the 6 x 18 x 32-channel chunk
  unsigned offA[C_AIT];
  bool inA[C_AIT];
#pragma unroll
  for (int it = 0; it < C_AIT; ++it) {
    const int idx = min(tid + it * 256, C_AN - 1);
    const int row = idx >> 3, c4 = idx & 7;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
  }
  auto load_a = [&](int kc, f32x4 (&ra)[C_AIT]) {       // K-tail lanes read k = 0 of the pixel, zeroed on store
    const char* base = (const char*)(p.A + (long)kc * 32);
#pragma unroll
    for (int it = 0; it < C_AIT; ++it) {
      const int c4 = min(tid + it * 256, C_AN - 1) & 7;
      const bool oob = kc * 32 + c4 * 4 >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - c4 * 16u : offA[it]));
    }
  };
  auto store_a = [&](const f32x4 (&ra)[C_AIT], int kc) {
#pragma unroll
    for (int it = 0; it < C_AIT; ++it) {
      if (C_AN % 256 == 0 || tid + it * 256 < C_AN) {
        const int idx = tid + it * 256;
        f32x4 v = ra[it];
        if (!inA[it] || kc * 32 + (idx & 7) * 4 >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};   // outside the image / K tail: exact zeros
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0_, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        unsigned char* dst = smem + (idx >> 3) * C_PITCH + (idx & 7) * 8;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *(u32x2*)(dst) = u32x2{h0, h1};
        if (!AMP) {
          *(u32x2*)(dst + C_APLANE) = u32x2{m0_, m1};
          *(u32x2*)(dst + 2 * C_APLANE) = u32x2{l0, l1};
        }
      }
    }
  };

  // ---- W fragments of (chunk, tap): rows tap*N + n of the tap-major pack, sub-chunk 2*chunk + (g >> 1)
  const long wrows = 9L * p.N;
  const long plane_bytes = wrows * p.Kp * 2;
  unsigned boff[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
    boff[jt] = (unsigned)(((g >> 1) * wrows + n0 + min(wave * 48 + jt * 16 + c, nvalid - 1)) * 32 + (g & 1) * 16);
  const int niter = nkc * 9;
  auto load_b = [&](int it, u32x4 (&fb)[3][3]) {          // iterations past the end re-read the last one
    const int itc = min(it, niter - 1);
    const int kc = itc / 9, tap = itc - kc * 9;
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = (i * 18 + c) * C_PITCH + 16 * g;     // image row i of the tile, pixel c, octet g

  auto mma = [&](int tap, const u32x4 (&fb)[3][3]) {
    const int toff = ((tap / 3) * 18 + (tap % 3)) * C_PITCH;                  // halo shift of the tap
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[3];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(smem + pl * C_APLANE + a_off[i] + toff);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  f32x4 ra[C_AIT];
  u32x4 fb0[3][3], fb1[3][3], fb2[3][3];
  load_a(0, ra);
  load_b(0, fb0); load_b(1, fb1); load_b(2, fb2);
  for (int kc = 0; kc < nkc; ++kc) {
    if (kc) __syncthreads();                  // every tap of the previous chunk has read the halo tile
    store_a(ra, kc);
    __syncthreads();
    if (kc + 1 < nkc) load_a(kc + 1, ra);     // nine taps to land
    const int it = kc * 9;
#pragma unroll 1
    for (int t3 = 0; t3 < 9; t3 += 3) {
      mma(t3, fb0);     if (it + t3 + 3 < niter) load_b(it + t3 + 3, fb0);
      mma(t3 + 1, fb1); if (it + t3 + 4 < niter) load_b(it + t3 + 4, fb1);
      mma(t3 + 2, fb2); if (it + t3 + 5 < niter) load_b(it + t3 + 5, fb2);
    }
  }

  // ---- re-layout into the 32x32 / 2 x 2 layout of nt_epi.h: tile row 16*y + x <-> its 32-row tile (2 image rows x 16)
  __syncthreads();
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
  nt_epilogue<1, 3, true>(p, acc2, lane, wm, wn, n0, nvalid, 0, img, y0, x0);
}

// ---------------------------------------------------------------------------------------------------------
// k_ntcw on TWO fp16 planes and three products (the scheme of k_nhcw2 below on the 64-pixel x 192-column tile: SwinIR's
// 180 -> 180 convs): one power-of-two scale per weight output channel (preparation job kind 4) and one per halo tile, kept
// as a running value over the channel chunks; the block exponents are undone while the accumulators are re-laid.
template <bool AMP>
__global__ void __launch_bounds__(256, 2) k_nhcw(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 2;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // neighbouring tiles (shared halos) in one L2
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * 4, x0 = tx * 16;
  const int nkc = (p.K + 31) / 32;

  // ---- halo staging: thread = up to C_AIT float4 (pixel, 4 channels) of the 6 x 18 x 32-channel chunk
  unsigned offA[C_AIT];
  bool inA[C_AIT];
#pragma unroll
  for (int it = 0; it < C_AIT; ++it) {
    const int idx = min(tid + it * 256, C_AN - 1);
    const int row = idx >> 3, c4 = idx & 7;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
  }
  auto load_a = [&](int kc, f32x4 (&ra)[C_AIT]) {       // K-tail lanes read k = 0 of the pixel, zeroed on store
    const char* base = (const char*)(p.A + (long)kc * 32);
#pragma unroll
    for (int it = 0; it < C_AIT; ++it) {
      const int c4 = min(tid + it * 256, C_AN - 1) & 7;
      const bool oob = kc * 32 + c4 * 4 >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - c4 * 16u : offA[it]));
    }
  };
  float* const red = (float*)(smem + 3 * C_APLANE);          // [4] wave maxima of the chunk (behind the halo planes)
  float cur = 3.0e38f;                                       // the tile's current 2^s (block-uniform)
  auto clean_a = [&](f32x4 (&ra)[C_AIT], int kc) -> float {  // zero what does not count, return the thread's maximum
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < C_AIT; ++it) {
      const int idx = tid + it * 256;
      f32x4 v = ra[it];
      if (!(C_AN % 256 == 0 || idx < C_AN) || !inA[it] || kc * 32 + (idx & 7) * 4 >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
      ra[it] = v;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    return mx;
  };
  auto store_a = [&](const f32x4 (&ra)[C_AIT], float use) {
#pragma unroll
    for (int it = 0; it < C_AIT; ++it) {
      if (C_AN % 256 == 0 || tid + it * 256 < C_AN) {
        const int idx = tid + it * 256;
        const f32x4 v = ra[it];
        unsigned h0, l0, h1, l1;
        split2_pair(v.x * use, v.y * use, h0, l0);
        split2_pair(v.z * use, v.w * use, h1, l1);
        unsigned char* dst = smem + (idx >> 3) * C_PITCH + (idx & 7) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        if (!AMP) *(u32x2*)(dst + C_APLANE) = u32x2{l0, l1};
      }
    }
  };

  // ---- W fragments of (chunk, tap): rows tap*N + n of the tap-major pack, sub-chunk 2*chunk + (g >> 1)
  const long wrows = 9L * p.N;
  const long plane_bytes = wrows * p.Kp * 2;
  const float* const winv_all = (const float*)((const char*)p.Wb + 2 * plane_bytes);     // 2^-s of the output channels
  unsigned boff[3];
  float winv[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    const int col = n0 + min(wave * 48 + jt * 16 + c, nvalid - 1);
    boff[jt] = (unsigned)(((g >> 1) * wrows + col) * 32 + (g & 1) * 16);
    winv[jt] = winv_all[col];
  }
  const int niter = nkc * 9;
  // every block walks the nine taps from another start (blocks of one XCD -- same blockIdx.x mod 8 -- get different ones):
  // in lockstep the whole launch would ask the L2 for the same tap's weight lines at the same moment (k_ntw, k_mlp_f16)
  const int rot9 = p.k_rot ? (int)(((unsigned)blockIdx.x >> 3) % 9u) : 0;
  auto tr9 = [&](int tap) { const int x = tap + rot9; return x >= 9 ? x - 9 : x; };
  auto load_b = [&](int it, u32x4 (&fb)[3][2]) {          // iterations past the end re-read the last one
    const int itc = min(it, niter - 1);
    const int kc = itc / 9, tap = tr9(itc - kc * 9);
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = (i * 18 + c) * C_PITCH + 16 * g;     // image row i of the tile, pixel c, octet g

  auto mma = [&](int tap, const u32x4 (&fb)[3][2]) {
    const int tp = tr9(tap), ty3 = (tp * 11) >> 5;                            // tp / 3 for tp < 9
    const int toff = (ty3 * 18 + (tp - 3 * ty3)) * C_PITCH;                      // halo shift of the tap
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(smem + pl * C_APLANE + a_off[i] + toff);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16h(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  f32x4 ra[C_AIT];
  u32x4 fb0[3][2], fb1[3][2], fb2[3][2];
  load_a(0, ra);
  load_b(0, fb0); load_b(1, fb1); load_b(2, fb2);
  for (int kc = 0; kc < nkc; ++kc) {
    const float tmx = wave_max(clean_a(ra, kc));
    if (kc) __syncthreads();                  // every tap of the previous chunk has read the halo tile (and `red`)
    if (lane == 0) red[wave] = tmx;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float need = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 3.0e38f;
    const float old = cur;
    cur = fminf(cur, need);                   // the scale only goes down: nothing accumulated can overflow
    const float use = cur > 1.0e38f ? 1.f : cur;
    store_a(ra, use);
    __syncthreads();
    if (kc + 1 < nkc) load_a(kc + 1, ra);     // nine taps to land
    if (kc && old != cur && old < 1.0e38f) {  // the tile's scale dropped: bring the accumulators to the new one (exact)
      const float f = cur / old;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] *= f;
    }
    const int it = kc * 9;
#pragma unroll 1
    for (int t3 = 0; t3 < 9; t3 += 3) {
      mma(t3, fb0);     if (it + t3 + 3 < niter) load_b(it + t3 + 3, fb0);
      mma(t3 + 1, fb1); if (it + t3 + 4 < niter) load_b(it + t3 + 4, fb1);
      mma(t3 + 2, fb2); if (it + t3 + 5 < niter) load_b(it + t3 + 5, fb2);
    }
  }

  const float tinv = 1.0f / (cur > 1.0e38f ? 1.f : cur);
  // ---- re-layout into the 32x32 / 2 x 2 layout of nt_epi.h: tile row 16*y + x <-> its 32-row tile (2 image rows x 16)
  __syncthreads();
  float* const T = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc[i][j][e] * (tinv * winv[j]);
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31;
  f32x16 acc2[1][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[0][j][q] = T[(wm * 32 + mfma_row(q, lane)) * TP + (wn * 3 + j) * 32 + r];
  nt_epilogue<1, 3, true>(p, acc2, lane, wm, wn, n0, nvalid, 0, img, y0, x0);
}

// ---------------------------------------------------------------------------------------------------------
// ... and for the 64-column conv tiles (128 pixels x 64 columns: the 64 -> 64 body convs of EDSR / VDSR /
// MSLapSRN / MemNet, and 64-column slices of wider outputs; operands / epilogues of k_ntb<2, 1, true>).  Waves
// 2 x 2 as in nt_epi.h -- wave (wm, wn) owns image rows 4*wm .. 4*wm+3 (four 16-pixel row tiles) x columns
// 32*wn .. +31 (two column tiles): the two row halves read the same W fragments (L1 hits), nothing of W goes
// through LDS, the nine taps of a channel chunk run without a barrier, and the accumulators reach nt_epi.h's
// layout through a wave-private LDS tile (the region of a wave is the same in both layouts).
constexpr int D_TP = 36;                         // pitch of the wave's re-layout tile (floats)
constexpr int ntcw2_lds(int rw) { return 3 * (2 * rw + 2) * 18 * C_PITCH; }   // 43200 / 25920 B (> 4 waves x 16 rw x D_TP x 4)

// RW = image rows per wave: 4 (128-pixel tiles, k_ntb<2, 1>'s shapes) or 2 (64-pixel tiles, k_ntb<1, 1>'s: small images)
template <int RW, bool AMP>
__global__ void __launch_bounds__(256, 2) k_ntcw2(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 3;
  constexpr int D_AROWS = (2 * RW + 2) * 18;       // halo pixels of a 2 RW x 16 tile
  constexpr int D_APLANE = D_AROWS * C_PITCH;
  constexpr int D_AN = D_AROWS * 8;
  constexpr int D_AIT = (D_AN + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  // one-dimensional grid, column block fastest: the column blocks of a pixel tile (they stage the same halo) are
  // neighbours in time and, with the XCD-aware order, in one L2
  int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int ncol = (p.N + p.n_tile - 1) / p.n_tile;
  const int n0 = (t % ncol) * p.n_tile; t /= ncol;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = (p.K + 31) / 32;

  unsigned offA[D_AIT];
  bool inA[D_AIT];
#pragma unroll
  for (int it = 0; it < D_AIT; ++it) {
    const int idx = min(tid + it * 256, D_AN - 1);
    const int row = idx >> 3, c4 = idx & 7;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    if (p.ps == 2) offA[it] = (unsigned)(((img * 2 * p.H + 2 * yc) * 2 * p.Wd + 2 * xc) * (int)p.lda + c4 * 4) * 4u;
    else offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
  }
  auto load_a = [&](int kc, f32x4 (&ra)[D_AIT]) {
    long koff = kc * 32;
    if (p.ps == 2) {     // chunk kc = channels c0.. of sub-pixel sp of the shuffled image (K/4 is a multiple of 32)
      const int fk = p.K >> 2, sp = (kc * 32) / fk, c0 = kc * 32 - sp * fk;
      koff = ((long)(sp >> 1) * 2 * p.Wd + (sp & 1)) * p.lda + c0;
    }
    const char* base = (const char*)(p.A + koff);
#pragma unroll
    for (int it = 0; it < D_AIT; ++it) {
      const int c4 = min(tid + it * 256, D_AN - 1) & 7;
      const bool oob = kc * 32 + c4 * 4 >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - c4 * 16u : offA[it]));
    }
  };
  auto store_a = [&](const f32x4 (&ra)[D_AIT], int kc) {
#pragma unroll
    for (int it = 0; it < D_AIT; ++it) {
      if (D_AN % 256 == 0 || tid + it * 256 < D_AN) {
        const int idx = tid + it * 256;
        f32x4 v = ra[it];
        if (!inA[it] || kc * 32 + (idx & 7) * 4 >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0_, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        unsigned char* dst = smem + (idx >> 3) * C_PITCH + (idx & 7) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        if (!AMP) {
          *(u32x2*)(dst + D_APLANE) = u32x2{m0_, m1};
          *(u32x2*)(dst + 2 * D_APLANE) = u32x2{l0, l1};
        }
      }
    }
  };

  const long wrows = 9L * p.N;
  const long plane_bytes = wrows * p.Kp * 2;
  unsigned boff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
    boff[jt] = (unsigned)(((g >> 1) * wrows + n0 + min(wn * 32 + jt * 16 + c, nvalid - 1)) * 32 + (g & 1) * 16);
  const int niter = nkc * 9;
  auto load_b = [&](int it, u32x4 (&fb)[2][3]) {
    const int kc = it / 9, tap = it - kc * 9;
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 18 + c) * C_PITCH + 16 * g;

  auto mma = [&](int tap, const u32x4 (&fb)[2][3]) {
    const int toff = ((tap / 3) * 18 + (tap % 3)) * C_PITCH;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      u32x4 fa[3];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(smem + pl * D_APLANE + a_off[i] + toff);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  f32x4 ra[D_AIT];
  load_a(0, ra);
  if constexpr (AMP) {
    // single-product form: a tap is 4 / 8 MFMAs per wave, so three taps of cover are nothing -- the fragments of a whole
    // channel chunk (9 taps, 18 registers each) are requested one chunk ahead (set of a tap = the tap)
    u32x4 fb[9][2][3];
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9)
      if (s9 < niter) load_b(s9, fb[s9]);
    for (int kc = 0; kc < nkc; ++kc) {
      if (kc) __syncthreads();
      store_a(ra, kc);
      __syncthreads();
      if (kc + 1 < nkc) load_a(kc + 1, ra);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        mma(tap, fb[tap]);
        if (kc + 1 < nkc) load_b(kc * 9 + tap + 9, fb[tap]);
      }
    }
  } else {
    u32x4 fb0[2][3], fb1[2][3], fb2[2][3];
    load_b(0, fb0); load_b(1, fb1); load_b(2, fb2);
    for (int kc = 0; kc < nkc; ++kc) {
      if (kc) __syncthreads();
      store_a(ra, kc);
      __syncthreads();
      if (kc + 1 < nkc) load_a(kc + 1, ra);
      const int it = kc * 9;
#pragma unroll 1
      for (int t3 = 0; t3 < 9; t3 += 3) {
        mma(t3, fb0);     if (it + t3 + 3 < niter) load_b(it + t3 + 3, fb0);
        mma(t3 + 1, fb1); if (it + t3 + 4 < niter) load_b(it + t3 + 4, fb1);
        mma(t3 + 2, fb2); if (it + t3 + 5 < niter) load_b(it + t3 + 5, fb2);
      }
    }
  }

  // ---- re-layout inside the wave: 4 x 2 tiles of 16 x 16 -> 2 x 1 tiles of 32 x 32 (tile row 16*y + x of the wave's 4 image rows)
  __syncthreads();                                   // the halo tile is dead from here on
  float* const T = (float*)smem + wave * (16 * RW * D_TP);
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * i + 4 * g + e) * D_TP + 16 * j + c] = acc[i][j][e];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the wave's own LDS writes have landed
  const int r = lane & 31;
  f32x16 acc2[RW / 2][1];
#pragma unroll
  for (int i = 0; i < RW / 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[i][0][q] = T[(32 * i + mfma_row(q, lane)) * D_TP + r];
  nt_epilogue<RW / 2, 1, true>(p, acc2, lane, wm, wn, n0, nvalid, 0, img, y0, x0);
}

// 16-byte epilogue of the 64-column conv kernels (k_nhcw2): the 2 RW x 16 pixel x 64 column tile row-major in LDS (pitch
// W_TP), a thread owns 8 consecutive columns of a pixel per pass -- the residual / mask operand comes as two 16-byte loads,
// the result leaves as two 16-byte stores.  The generic nt_epilogue moves single floats (a lane holds one column of 16 rows):
// 4.3 us of a 23-us block on 256 x 256 images.  Same arithmetic per element, in the same order (nt_epi.h); needs N % 8 == 0
// and 16-byte aligned C / R rows (the dispatcher decides: NtArgs.wide_epi).
constexpr int W_TP = 68;
template <int RW>
__device__ __forceinline__ void nhcw2_epilogue_wide(const NtArgs& p, const float* T, int tid, int n0, int nvalid, int img, int y0,
                                                    int x0) {
  constexpr int NPX = 2 * RW * 16;
  float blk_s = p.alpha;
  if (p.rowscale) blk_s *= p.rowscale[img];
  const bool prelu = p.epi == 9 || p.epi == 10;
  const float slope = prelu ? ldg_f(p.slope) : 0.f;
  if (prelu) blk_s = 1.f;
  const bool needR = p.R != nullptr && p.epi >= 2 && p.epi != 9 && p.epi != 11;
  const bool shuf = p.ps == 1;
  const int fs = p.N >> 2;
#pragma unroll
  for (int it = 0; it < (NPX * 8) / 256; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = (idx & 7) * 8;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    if (y >= p.H || x >= p.Wd || c8 >= nvalid) continue;
    const int gn = n0 + c8;
    const int sp = shuf ? gn / fs : 0, cc = shuf ? gn - sp * fs : gn;
    const long grow = ((long)img * p.H + y) * p.Wd + x;
    f32x4 v0 = *(const f32x4*)(T + px * W_TP + c8), v1 = *(const f32x4*)(T + px * W_TP + c8 + 4);
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
    if (needR) { r0 = ldg_f4(p.R + grow * p.ldr + gn); r1 = ldg_f4(p.R + grow * p.ldr + gn + 4); }
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const float rv[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
    if (p.bias) {
      if (shuf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[(cc + e) * 4 + sp];
      } else {
        const f32x4 b0 = ldg_f4(p.bias + gn), b1 = ldg_f4(p.bias + gn + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
    }
    switch (p.epi) {
      case 1:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        break;
      case 2:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * blk_s + rv[e];
        break;
      case 4:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
        break;
      case 6:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
        break;
      case 7:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rv[e] > 0.f ? v[e] : v[e] * p.alpha;
        break;
      case 8:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * blk_s + rv[e], 0.f);
        break;
      case 9:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
        break;
      case 10:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] > 0.f ? v[e] : slope * v[e]) + p.alpha * rv[e];
        break;
      case 11:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        break;
      default:
        break;
    }
    float* dst;
    if (shuf) dst = p.C + (((long)img * 2 * p.H + 2 * y + (sp >> 1)) * (2 * p.Wd) + 2 * x + (sp & 1)) * p.ldc + cc;
    else dst = p.C + grow * p.ldc + gn;
    *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

// k_nhcw2 (experiment, SRHIP_F16X2_CONV=1; weight planes of prep kind 4): k_ntcw2 on TWO fp16 planes and three products.
// The accumulator of an output pixel mixes nine neighbouring pixels, so the activation's block exponent is ONE power
// of two per halo tile (all channels), kept as a running scale over the channel chunks exactly as k_nth2 does per row:
// a chunk's tile maximum comes out of the registers that hold the chunk, the scale only goes down, and when it does
// every accumulator of the block is multiplied by the exact power of two in between.  Emulated (tools/split_accuracy.py):
// worst output pixel relative to itself within 1.4x of an f32 conv, also on gradient-like inputs.
// phase timestamps of every wave of k_nhcw2 (tools/mb_conv64_phases.py): experiment builds only
#ifdef SRHIP_EXPERIMENTS
__device__ long long* g_nhcw2_dbg = nullptr;
#define SR_TSC(K) \
  if (g_nhcw2_dbg && lane == 0) g_nhcw2_dbg[((long)blockIdx.x * 4 + wave) * 16 + (K)] = (long long)wall_clock64();
#else
#define SR_TSC(K)
#endif
#ifndef SR_NHCW2_OCC
#define SR_NHCW2_OCC 3
#endif
template <bool DEEP_> struct Nhcw2Ring { static constexpr int SETS = DEEP_ ? 9 : 3, OCC = DEEP_ ? 2 : SR_NHCW2_OCC; };
template <int RW, bool AMP, bool DEEP = false>
// three blocks per CU (166 VGPRs at RW = 4, no spill; 43 KB of LDS each): same box, against two -- the 64 -> 64 conv at 8 x 256 x 256
// 152.4 -> 142.4 us, EDSR x8 training step 4.72 -> 4.65 ms, VDSR / DRRN evaluation 9.23 -> 8.75 / 88.8 -> 81.8 ms per batch
// DEEP (RW = 2, at most two blocks per CU: the launches of a few hundred blocks -- EDSR x8's body convs at 8 x 64 x 64 are 512 blocks, two per
// CU) runs at the latency of its own chain, not at a rate: tools/mb_conv64_phases.py, round 5 -- 16.2 us per block, of which the
// 18 taps take 6.8 us for 1.6 us of MFMA issue: every tap waits for its weight fragments, requested three taps (0.27 us of
// matrix time) ahead of an L2 that all 512 blocks ask for the same lines at the same moment (~1.1 us).  Two blocks per CU
// leave 256 registers: there the ring is NINE register sets -- a whole 32-channel chunk ahead; the fragments of chunk k + 1
// travel while chunk k's taps and the staging of chunk k + 1 run.
__global__ void __launch_bounds__(256, Nhcw2Ring<DEEP>::OCC) k_nhcw2(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 2;
  constexpr int D_AROWS = (2 * RW + 2) * 18;       // halo pixels of a 2 RW x 16 tile
  constexpr int D_APLANE = D_AROWS * C_PITCH;
  constexpr int D_AN = D_AROWS * 8;
  constexpr int D_AIT = (D_AN + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  // one-dimensional grid, column block fastest: the column blocks of a pixel tile (they stage the same halo) are
  // neighbours in time and, with the XCD-aware order, in one L2
  int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int ncol = (p.N + p.n_tile - 1) / p.n_tile;
  const int n0 = (t % ncol) * p.n_tile; t /= ncol;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = (p.K + 31) / 32;
  SR_TSC(0)

  unsigned offA[D_AIT];
  bool inA[D_AIT];
#pragma unroll
  for (int it = 0; it < D_AIT; ++it) {
    const int idx = min(tid + it * 256, D_AN - 1);
    const int row = idx >> 3, c4 = idx & 7;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    if (p.ps == 2) offA[it] = (unsigned)(((img * 2 * p.H + 2 * yc) * 2 * p.Wd + 2 * xc) * (int)p.lda + c4 * 4) * 4u;
    else offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
  }
  // input prologue (evaluation-mode BatchNorm + ReLU in front of the conv, network_memnet.py:27-34): relu((x - mean) k +
  // beta) per channel, applied to the halo tile on its way into the stage images -- the zero padding is of the
  // ACTIVATION, so it is applied before the out-of-image pixels are zeroed.  Same expression as k_bn_apply (bn.hip).
  f32x4 pro_mean = {0.f, 0.f, 0.f, 0.f}, pro_k = {1.f, 1.f, 1.f, 1.f}, pro_beta = {0.f, 0.f, 0.f, 0.f};
  auto load_a = [&](int kc, f32x4 (&ra)[D_AIT]) {
    long koff = kc * 32;
    if (p.ps == 2) {     // chunk kc = channels c0.. of sub-pixel sp of the shuffled image (K/4 is a multiple of 32)
      const int fk = p.K >> 2, sp = (kc * 32) / fk, c0 = kc * 32 - sp * fk;
      koff = ((long)(sp >> 1) * 2 * p.Wd + (sp & 1)) * p.lda + c0;
    }
    const char* base = (const char*)(p.A + koff);
#pragma unroll
    for (int it = 0; it < D_AIT; ++it) {
      const int c4 = min(tid + it * 256, D_AN - 1) & 7;
      const bool oob = kc * 32 + c4 * 4 >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - c4 * 16u : offA[it]));
    }
    if (p.pro_coef) {    // the thread's four channels of the chunk (256 % 8 == 0: the same in every iteration)
      const int ch = min(kc * 32 + (tid & 7) * 4, p.K - 4);
      pro_mean = ldg_f4(p.pro_coef + ch);
      pro_k = ldg_f4(p.pro_coef + 2 * p.K + ch);
      pro_beta = ldg_f4(p.pro_coef + 3 * p.K + ch);
    }
  };
  float* const red = (float*)(smem + 3 * D_APLANE);          // [4] wave maxima of the chunk (behind the halo planes)
  float cur = 3.0e38f;                                       // the tile's current 2^s (block-uniform)
  // zero what does not count, return the thread's maximum
  auto clean_a = [&](f32x4 (&ra)[D_AIT], int kc) -> float {
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < D_AIT; ++it) {
      const int idx = tid + it * 256;
      f32x4 v = ra[it];
      if (p.pro_coef) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf((v[e] - pro_mean[e]) * pro_k[e] + pro_beta[e], 0.f);
      }
      if (!(D_AN % 256 == 0 || idx < D_AN) || !inA[it] || kc * 32 + (idx & 7) * 4 >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
      ra[it] = v;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    return mx;
  };
  auto store_a = [&](const f32x4 (&ra)[D_AIT], float use) {
#pragma unroll
    for (int it = 0; it < D_AIT; ++it) {
      if (D_AN % 256 == 0 || tid + it * 256 < D_AN) {
        const int idx = tid + it * 256;
        const f32x4 v = ra[it];
        unsigned h0, l0, h1, l1;
        split2_pair(v.x * use, v.y * use, h0, l0);
        split2_pair(v.z * use, v.w * use, h1, l1);
        unsigned char* dst = smem + (idx >> 3) * C_PITCH + (idx & 7) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        if (!AMP) *(u32x2*)(dst + D_APLANE) = u32x2{l0, l1};
      }
    }
  };

  const long wrows = 9L * p.N;
  const long plane_bytes = wrows * p.Kp * 2;
  const float* const winv_all = (const float*)((const char*)p.Wb + 2 * plane_bytes);
  unsigned boff[2];
  float winv[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = n0 + min(wn * 32 + jt * 16 + c, nvalid - 1);
    boff[jt] = (unsigned)(((g >> 1) * wrows + col) * 32 + (g & 1) * 16);
    winv[jt] = winv_all[col];
  }
  const int niter = nkc * 9;
  // every block walks the nine taps from another start (blocks of one XCD -- same blockIdx.x mod 8 -- get different ones):
  // in lockstep the whole launch would ask the L2 for the same tap's weight lines at the same moment (k_ntw, k_mlp_f16)
  const int rot9 = p.k_rot ? (int)(((unsigned)blockIdx.x >> 3) % 9u) : 0;
  auto tr9 = [&](int tap) { const int x = tap + rot9; return x >= 9 ? x - 9 : x; };
  auto load_b = [&](int it, u32x4 (&fb)[2][2]) {
    const int kc = it / 9, tap = tr9(it - kc * 9);
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 18 + c) * C_PITCH + 16 * g;

  auto mma = [&](int tap, const u32x4 (&fb)[2][2]) {
    const int tp = tr9(tap), ty3 = (tp * 11) >> 5;                            // tp / 3 for tp < 9
    const int toff = (ty3 * 18 + (tp - 3 * ty3)) * C_PITCH;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(smem + pl * D_APLANE + a_off[i] + toff);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma16h(fa[PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };

  f32x4 ra[D_AIT];
  load_a(0, ra);
  constexpr int NSETS = Nhcw2Ring<DEEP>::SETS;
  u32x4 fbr[NSETS][2][2];
#pragma unroll
  for (int q = 0; q < NSETS; ++q)
    if (q < niter) load_b(q, fbr[q]);
  SR_TSC(1)
  for (int kc = 0; kc < nkc; ++kc) {
    const float tmx = wave_max(clean_a(ra, kc));
    if (kc) __syncthreads();                  // every tap of the previous chunk has read the halo tile (and `red`)
    if (lane == 0) red[wave] = tmx;
    __syncthreads();
    if (kc < 2) { SR_TSC(2 + 4 * kc) }
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float need = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 3.0e38f;
    const float old = cur;
    cur = fminf(cur, need);
    const float use = cur > 1.0e38f ? 1.f : cur;
    store_a(ra, use);
    __syncthreads();
    if (kc < 2) { SR_TSC(3 + 4 * kc) }
    if (kc + 1 < nkc) load_a(kc + 1, ra);
    if (kc && old != cur && old < 1.0e38f) {  // the tile's scale dropped: bring the accumulators to the new one (exact)
      const float f = cur / old;
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] *= f;
    }
    const int it = kc * 9;
    if constexpr (NSETS == 9) {                // the ring holds a chunk: tap t in set t, refilled with the next chunk's tap t
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        mma(t, fbr[t]);
        if (it + t + 9 < niter) load_b(it + t + 9, fbr[t]);
      }
    } else {
#pragma unroll 1
      for (int t3 = 0; t3 < 9; t3 += 3) {
        mma(t3, fbr[0]);     if (it + t3 + 3 < niter) load_b(it + t3 + 3, fbr[0]);
        mma(t3 + 1, fbr[1]); if (it + t3 + 4 < niter) load_b(it + t3 + 4, fbr[1]);
        mma(t3 + 2, fbr[2]); if (it + t3 + 5 < niter) load_b(it + t3 + 5, fbr[2]);
      }
    }
    if (kc < 2) { SR_TSC(4 + 4 * kc) }
  }
  const float tinv = 1.0f / (cur > 1.0e38f ? 1.f : cur);

  if (p.wide_epi) {                                  // block-uniform (set by the dispatcher): 16-byte epilogue accesses
    __syncthreads();                                 // the halo tile is dead from here on
    SR_TSC(10)
    float* const Tw = (float*)smem;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          Tw[(16 * (RW * wm + i) + 4 * g + e) * W_TP + 32 * wn + 16 * j + c] = acc[i][j][e] * (tinv * winv[j]);
    __syncthreads();
    SR_TSC(11)
    nhcw2_epilogue_wide<RW>(p, Tw, tid, n0, nvalid, img, y0, x0);
    SR_TSC(12)
    return;
  }
  // ---- re-layout inside the wave: 4 x 2 tiles of 16 x 16 -> 2 x 1 tiles of 32 x 32 (tile row 16*y + x of the wave's 4 image rows)
  __syncthreads();                                   // the halo tile is dead from here on
  SR_TSC(10)
  float* const T = (float*)smem + wave * (16 * RW * D_TP);
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * i + 4 * g + e) * D_TP + 16 * j + c] = acc[i][j][e] * (tinv * winv[j]);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the wave's own LDS writes have landed
  const int r = lane & 31;
  f32x16 acc2[RW / 2][1];
#pragma unroll
  for (int i = 0; i < RW / 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[i][0][q] = T[(32 * i + mfma_row(q, lane)) * D_TP + r];
  SR_TSC(11)
  nt_epilogue<RW / 2, 1, true>(p, acc2, lane, wm, wn, n0, nvalid, 0, img, y0, x0);
  SR_TSC(12)
}

}  // namespace

#ifdef SRHIP_EXPERIMENTS
// [blocks][4 waves][16] stamps of the 100 MHz wall clock
SR_DEBUG_EXPORT int srhip_nhcw2_debug_buffer(long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_nhcw2_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#endif

// rows_per_wave 4: tiles_y counts 8-row tiles; 2: 4-row tiles (the caller's wm = 2 / 1)
int sr_conv3x3_ntcw2(NtArgs& p, int rows_per_wave, hipStream_t st) {
  static_assert(ntcw2_lds(4) >= 4 * 64 * D_TP * 4 && ntcw2_lds(2) >= 4 * 32 * D_TP * 4, "LDS regions");
  dim3 grid(p.tiles_x * p.tiles_y * p.batch * sr_cdiv(p.N, p.n_tile));
  if (rows_per_wave == 4) {
    if (p.amp) hipLaunchKernelGGL((k_ntcw2<4, true>), grid, dim3(256), ntcw2_lds(4), st, p);
    else hipLaunchKernelGGL((k_ntcw2<4, false>), grid, dim3(256), ntcw2_lds(4), st, p);
  } else {
    if (p.amp) hipLaunchKernelGGL((k_ntcw2<2, true>), grid, dim3(256), ntcw2_lds(2), st, p);
    else hipLaunchKernelGGL((k_ntcw2<2, false>), grid, dim3(256), ntcw2_lds(2), st, p);
  }
  SR_LAUNCH_CHECK("k_ntcw2");
  return 0;
}

int sr_conv3x3_nhcw2(NtArgs& p, int rows_per_wave, hipStream_t st) {
  static_assert(ntcw2_lds(4) >= 128 * W_TP * 4 && ntcw2_lds(2) >= 64 * W_TP * 4, "LDS: the row-major output tile");
  {
    const auto al4 = [](const void* q, long ld) { return !q || (((size_t)q & 15) == 0 && ld % 4 == 0); };
    p.wide_epi = p.N % 8 == 0 && al4(p.C, p.ldc) && al4(p.R, p.ldr) && al4(p.bias, 4) && (p.ps != 1 || (p.N >> 2) % 8 == 0) &&
                 sr_getenv("SRHIP_NHCW2_WIDE_OFF") == nullptr;
  }
  SR_REQUIRE(p.epi != 11 || p.wide_epi, "conv3x3: the GELU epilogue needs Cout %% 8 == 0 and 16-byte aligned rows (Cout=%d)", p.N);
  dim3 grid(p.tiles_x * p.tiles_y * p.batch * sr_cdiv(p.N, p.n_tile));
  { static const int crot = [] { const char* e = sr_getenv("SRHIP_CONV_ROT"); return e ? atoi(e) : 1; }(); p.k_rot = crot; }
  const bool amp = p.amp != 0;
  const int lds = ntcw2_lds(rows_per_wave) + 64;       // + the four wave maxima behind the (three-plane sized) halo region
  if (rows_per_wave == 4) {
    if (amp) hipLaunchKernelGGL((k_nhcw2<4, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((k_nhcw2<4, false>), grid, dim3(256), lds, st, p);
  } else if (!amp && grid.x <= 512 && !sr_getenv("SRHIP_NHCW2_DEEP_OFF")) {   // at most two blocks per CU: the whole-chunk weight ring
    hipLaunchKernelGGL((k_nhcw2<2, false, true>), grid, dim3(256), lds, st, p);
  } else {
    if (amp) hipLaunchKernelGGL((k_nhcw2<2, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((k_nhcw2<2, false>), grid, dim3(256), lds, st, p);
  }
  SR_LAUNCH_CHECK("k_nhcw2");
  return 0;
}

int sr_conv3x3_nhcw(NtArgs& p, hipStream_t st) {        // as sr_conv3x3_ntcw, weight planes of preparation job kind 4
  dim3 grid(p.tiles_x * p.tiles_y * p.batch, sr_cdiv(p.N, p.n_tile));
  { static const int crot = [] { const char* e = sr_getenv("SRHIP_CONV_ROT"); return e ? atoi(e) : 1; }(); p.k_rot = crot; }
  if (p.amp) hipLaunchKernelGGL(k_nhcw<true>, grid, dim3(256), NTCW_LDS, st, p);
  else hipLaunchKernelGGL(k_nhcw<false>, grid, dim3(256), NTCW_LDS, st, p);
  SR_LAUNCH_CHECK("k_nhcw");
  return 0;
}

// 64-pixel x 192-column conv tiles of the f32-accurate path (gemm_ntb.hip decides; tiles_x / tiles_y / n_tile set by the caller)
int sr_conv3x3_ntcw(NtArgs& p, hipStream_t st) {
  dim3 grid(p.tiles_x * p.tiles_y * p.batch, sr_cdiv(p.N, p.n_tile));
  if (p.amp) hipLaunchKernelGGL(k_ntcw<true>, grid, dim3(256), NTCW_LDS, st, p);
  else hipLaunchKernelGGL(k_ntcw<false>, grid, dim3(256), NTCW_LDS, st, p);
  SR_LAUNCH_CHECK("k_ntcw");
  return 0;
}

// 192-column tiles of the f32-accurate path (gemm_ntp.hip decides): n_tile is set by the caller.
int sr_gemm_ntw(NtArgs& p, hipStream_t st) {
  static_assert(NTW_LDS >= 2 * A_STAGE && NTW_LDS >= BM * TP * 4, "LDS regions");
  dim3 grid(sr_cdiv(p.M, BM) * sr_cdiv(p.N, p.n_tile));
  static const int gridorder = [] { const char* e = sr_getenv("SRHIP_NTW_GRID"); return e ? atoi(e) : 1; }();
  p.xcd_order = gridorder;
#ifdef SRHIP_EXPERIMENTS
  static const int dbg = [] { const char* e = sr_getenv("SRHIP_NTW_DBG"); return e ? atoi(e) : 0; }();
#endif
  static const int rot = [] { const char* e = sr_getenv("SRHIP_NTW_ROT"); return e ? atoi(e) : 1; }();
  p.k_rot = rot;
  if (p.wfmt == 1) {              // the caller's planes are prep kind 3 (two fp16 planes + row scales): also under --amp
    if (p.amp && p.epi != 5) hipLaunchKernelGGL(k_nth2<true>, grid, dim3(256), NTH2_LDS, st, p);
    else hipLaunchKernelGGL(k_nth2<false>, grid, dim3(256), NTH2_LDS, st, p);
    SR_LAUNCH_CHECK("k_nth2");
    return 0;
  }
  if (p.amp) {
    hipLaunchKernelGGL((k_ntw<false, true>), grid, dim3(256), NTW_LDS, st, p);
#ifdef SRHIP_EXPERIMENTS
  } else if (dbg) {               // timing ablations of the K loop (results are wrong on purpose)
    p.dbg = dbg;
    hipLaunchKernelGGL(k_ntw<true>, grid, dim3(256), NTW_LDS, st, p);
#endif
  } else {
    hipLaunchKernelGGL(k_ntw<false>, grid, dim3(256), NTW_LDS, st, p);
  }
  SR_LAUNCH_CHECK("k_ntw");
  return 0;
}
