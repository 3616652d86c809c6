// The MLP half of a Swin block as ONE kernel per direction on the two-plane fp16 split MFMA (three products,
// block exponents: the arithmetic of k_nth2, gemm_ntw.hip) -- reference Mlp.forward + the residual around it,
// dlib/models/network_swinir.py:28-45,335-337, and their autograd:
//
//   forward : h = LN(x) W1f^T + b1f ,  out = x + s (gelu(h) W2^T + b2) ,  stats_out = {mean, rstd} of the out rows
//   backward: dh = (s dy W2) * gelu'(h) ,  gh = gelu(h) ,  dx = dy + LayerNorm_backward(dh W1f; x, stats)
//
// Why one kernel: a launch of the Linear GEMM is launch + A fetch + K loop + epilogue IN SERIES (all 512 blocks of
// a launch run in lockstep, one round), and the K loop is the smallest of the four.  Chained here, the hidden
// activation never leaves the CU: one A fetch and one epilogue per direction instead of two, no read-back of h.
//
// How the chain works without a transposition through LDS.  GEMM 1 runs TRANSPOSED: the weight fragments take the
// MFMA's row side and the activation fragments its column side (both operands of v_mfma_f32_16x16x32_f16 have the
// same lane layout: index = lane & 15, k octet = lane >> 4, so swapping them is free), which leaves lane (c, g)
// with FOUR CONSECUTIVE hidden units 16j + 4g .. + 3 of ONE token 16i + c per accumulator -- half a 16-byte
// [token][8 k] unit of GEMM 2's A stage image (one ds_write_b64 per plane), and one 16-byte global store of h.
// A wave owns 48 hidden units per 192-unit half (two halves: hidden <= 384) for all 64 tokens; the whole hidden
// row of a token is in registers (of four waves) before GEMM 2 starts, so its block exponent is known up front
// (token maximum: in-lane, two shuffles, one 1-KB LDS exchange) and needs no running rescale.  GEMM 2 is k_nth2's
// loop: W fragments straight from global memory (planes [Kp/16][N][16], three register sets), A from the stage
// images, six barrier-free stages per 192-k pass.  The epilogues run on the row-major re-laid tile: 16-byte
// accesses, row statistics by four lanes per row.
//
// LDS: one 49-KB region serves, in turn, the x stage images, the two passes of hidden images and the output tile;
// two blocks per CU.
#include "common.h"
#include "kernels.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SK = 32;                 // k per stage
constexpr int BM = 64;                 // tokens per block
constexpr int APL = BM * 64;           // bytes of one plane of a stage image (64 rows x 32 fp16)
constexpr int AST = 2 * APL;           // a stage image: two planes
constexpr int TP = 196;                // pitch of the output tile (floats): rows 4 apart land 16 banks apart
constexpr int R0 = BM * TP * 4;        // the shared region (>= 6 stage images)
constexpr int MLP_LDS = R0 + (4 * 64 + 64 + 64 + 64) * 4;
static_assert(R0 >= 6 * AST, "LDS region");

__device__ __forceinline__ f32x4 mfma16h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
typedef _Float16 sr_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
  const sr_f16x2 hv = __builtin_convertvector(sr_f32x2{x0, x1}, sr_f16x2);
  const float r0 = x0 - (float)hv.x, r1 = x1 - (float)hv.y;
  const sr_f16x2 lv = __builtin_convertvector(sr_f32x2{r0, r1}, sr_f16x2);
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}
// 16-byte unit (row, k octet u) of a stage plane (gemm_ntw.hip): conflict free for the 16-lane phases of the
// fragment reads and of the staging writes
__device__ __forceinline__ int a_slot(int row, int u) { return row * 4 + (u ^ ((row >> 2) & 3)); }

// Phi(x), phi(x) with one exp (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7): the backward's gate, as epilogue 3
// of nt_epi.h
__device__ __forceinline__ void gelu_gate(float x, float& gate, float& gl) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float e1 = __expf(-0.5f * x * x);
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e1, x));
  gate = cdf + x * 0.39894228040143267794f * e1;
  gl = x * cdf;
}

// phase timestamps of every wave (tools/mb_mlp_phases.py): experiment builds only
#ifdef SRHIP_EXPERIMENTS
long long* g_mlp_dbg = nullptr;
int* g_mlp_cu_count = nullptr;
#define SR_TS(K) \
  if (p.dbg && lane == 0) p.dbg[((long)blockIdx.x * 4 + wave) * 32 + (K)] = (long long)wall_clock64();
#else
#define SR_TS(K)
#endif

// x Phi(x) with the same Phi: the forward's activation.  erff() costs ~50 VALU instructions per element and a lane holds
// 96 hidden units: the exact form was 18 us of a 50-us block (two waves per SIMD), this one is a third of it.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float e1 = __expf(-0.5f * x * x);
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return x * (0.5f * (1.0f + copysignf(1.0f - poly * e1, x)));
}

// gelu_poly2 / gelu_fast2 (the same two functions on PAIRS of f32: v_pk_mul / v_pk_fma / v_pk_add_f32) live in common.h --
// the grouped weight-gradient launch applies the SAME x Phi(x) to the saved pre-activations in its operand prologue, so the
// backward no longer stores gelu(h).
__device__ __forceinline__ void gelu_gate2(sr_f32x2 x, sr_f32x2& gate, sr_f32x2& gl) {
  sr_f32x2 e1;
  const sr_f32x2 cdf = gelu_poly2(x, e1);
  gate = cdf + (x * 0.39894228040143267794f) * e1;
  gl = x * cdf;
}

// AMP (forward only; srhip_set_matmul_mode(1): inference under --amp): ONE product of the leading fp16 planes -- no lo
// planes are staged, loaded or multiplied: a third of the matrix-core work, half of the stage-image and weight traffic
template <bool BWD, bool AMP = false>
__global__ void __launch_bounds__(256, 2) k_mlp_f16(MlpF16Args p) {
  constexpr int NPL = AMP ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const smax = (float*)(smem + R0);           // [4 waves][64 tokens] maxima of |A2|
  float* const rinvx = smax + 256;                   // [64] 2^-s of the X rows (backward; forward: a constant)
  float* const rinv2 = rinvx + 64;                   // [64] 2^-s of the hidden rows
  float* const rinv3 = rinv2 + 64;                   // [64] 2^-s of the dx rows (backward with the chained product)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int m0 = sr_xcd_block((int)blockIdx.x, gridDim.x) * BM;
  const int NH = (p.hid + 191) / 192;                // 192-unit halves of the hidden layer (1 or 2)
  const int nst1 = p.Kp1 / SK, nst2 = p.Kp2 / SK;
  // Every block walks the SAME weight planes; in lockstep all 64 blocks of an XCD ask its L2 for the same few lines at the
  // same moment.  As k_ntw (gemm_ntw.hip), block b starts every six-stage K walk at stage rot(b) and wraps around (the blocks
  // of one XCD -- same blockIdx.x mod 8 -- get different rotations): the stage images of a pass are all in LDS before its
  // products start, so any order serves; sums are f32 either way, only their order differs per row block (deterministic).
  const int rot = p.k_rot ? (int)(((unsigned)blockIdx.x >> 3) % 6u) : 0;
  auto rs6 = [&](int s6) { const int x = s6 + rot; return x >= 6 ? x - 6 : x; };      // s6 in 0 .. 5

  // ---------------- weight fragment addressing
  const long plane1 = (long)p.N1 * p.Kp1 * 2, plane2 = (long)p.N2 * p.Kp2 * 2;
  const float* const winv1 = (const float*)((const char*)p.W1 + 2 * plane1);
  const float* const winv2 = (const float*)((const char*)p.W2 + 2 * plane2);
  unsigned boff1[2][3], boff2[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = min(192 * hh + wave * 48 + jt * 16 + c, p.N1 - 1);
      boff1[hh][jt] = (unsigned)(((g >> 1) * p.N1 + row) * 32 + (g & 1) * 16);
    }
    const int col = min(wave * 48 + jt * 16 + c, p.N2 - 1);
    boff2[jt] = (unsigned)(((g >> 1) * p.N2 + col) * 32 + (g & 1) * 16);
  }
  // stage u of GEMM 1 = (half u / 6, k stage u % 6); stages past the weight's K read its last stage (the A planes are
  // zero there, the planes finite: they add exact zeros)
  auto load_b1 = [&](int u, u32x4 (&fb)[3][2]) {
    const int hh = u / 6, s = min(rs6(u - 6 * hh), nst1 - 1);
    const char* base = (const char*)p.W1 + (long)(2 * s) * p.N1 * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane1 + (hh ? boff1[1][jt] : boff1[0][jt]));
  };
  auto load_b2 = [&](int cs, u32x4 (&fb)[3][2]) {
    const int pass6 = (cs / 6) * 6;
    const char* base = (const char*)p.W2 + (long)(2 * min(pass6 + rs6(cs - pass6), nst2 - 1)) * p.N2 * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane2 + boff2[jt]);
  };
#ifdef SRHIP_EXPERIMENTS
  {
    // which CU runs this block: {xcc, se, sh, cu} of HW_REG_XCC_ID / HW_REG_HW_ID
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
    const unsigned cuid = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15);
    if (p.dbg && lane == 0) p.dbg[((long)blockIdx.x * 4 + wave) * 32 + 24] = (long long)cuid;
    bool late = blockIdx.x & 1;                      // mode 0: odd blocks
    if (p.stagger_mode == 2) late = blockIdx.x >= gridDim.x / 2;     // mode 2: the upper half of the grid (blocks b and b + 256 share a CU at 512 blocks)
    if (p.stagger_mode == 1 && p.cu_count) {         // mode 1: the block that arrives second on its CU
      __shared__ int s_slot;
      if (tid == 0) s_slot = atomicAdd(p.cu_count + cuid, 1);
      __syncthreads();
      late = s_slot & 1;
      if (p.dbg && lane == 0) p.dbg[((long)blockIdx.x * 4 + wave) * 32 + 25] = (long long)s_slot;
    }
    if (p.stagger > 0 && late) {                     // experiment: every second block starts late (phases of the two halves interleave)
      const long long t0 = (long long)wall_clock64();
      while ((long long)wall_clock64() - t0 < p.stagger) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif
  SR_TS(0)
  u32x4 fb0[3][2], fb1[3][2], fb2[3][2];
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = a_slot(16 * i + c, g) * 16;
  float* const T = (float*)smem;

  // LayerNorm backward on the row-major tile in LDS (affine folded into the weight in front of it), four lanes per row:
  //   o = res + rstd (d - mean(d) - xhat mean(d xhat)),  xhat = (x - mean) rstd.
  // A thread owns the k octets (stage s6, octet q) of its row -- columns 32 s6 + 8 q .. + 7, the x staging's own mapping:
  // a wave instruction covers 16 rows x 128 contiguous bytes (each cache line used whole by the instruction pair e = 0, 1),
  // and an octet is one 16-byte unit of a stage image afterwards.  (Round 4 gave a thread 48 CONSECUTIVE columns: every
  // lane of an instruction in a line of its own, each line touched by twelve instructions with the block's 92 KB of x /
  // residual rows passing through the 32-KB L1 in between -- 11 us per call, the same alone on the CU or with a partner.)
  // The rows are stored and stay in the registers of the threads that computed them; returns the largest |o| of the
  // thread's part.
  auto ln_bwd_rows = [&](const float* xp, long ldxp, const float* stp, const float* rp, long ldrp, float* op, long ldop,
                         f32x4 (&dv)[12]) -> float {
    const int row = tid >> 2, q = tid & 3;
    const int gm = min(m0 + row, p.M - 1);
    const float2 st = *(const float2*)(stp + 2 * (long)gm);
    f32x4 xh[12], rr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int col = (k >> 1) * 32 + q * 8 + (k & 1) * 4;
      const int cc = col < p.C ? col : 0;              // past the end: the row's first piece, zeroed below
      xh[k] = *(const f32x4*)(xp + (long)gm * ldxp + cc);
      rr[k] = *(const f32x4*)(rp + (long)gm * ldrp + cc);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int col = (k >> 1) * 32 + q * 8 + (k & 1) * 4;
      dv[k] = *(const f32x4*)(T + row * TP + col);
      const bool ok = col < p.C;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[k][e] = ok ? (xh[k][e] - st.x) * st.y : 0.f;
        dv[k][e] = ok ? dv[k][e] : 0.f;
        s1 += dv[k][e];
        s2 += dv[k][e] * xh[k][e];
      }
    }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    const float m1 = s1 * (1.0f / (float)p.C), m2 = s2 * (1.0f / (float)p.C);
    float omx = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int col = (k >> 1) * 32 + q * 8 + (k & 1) * 4;
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col < p.C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rr[k][e] + st.y * (dv[k][e] - m1 - xh[k][e] * m2);
        if (m0 + row < p.M) *(f32x4*)(op + (long)gm * ldop + col) = o;
      }
      dv[k] = o;
      omx = fmaxf(fmaxf(omx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    return omx;
  };
  // a thread's six octets (stage s6, octet q) of its row -> the stage images under the row's block exponent (row maximum
  // over the row's four lanes); the 2^-s of the row goes to rinv_dst
  auto rows_to_images = [&](const f32x4 (&dv)[12], float omx, float* rinv_dst) {
    const int row = tid >> 2, q = tid & 3;
    omx = fmaxf(omx, __shfl_xor(omx, 1, 64));
    omx = fmaxf(omx, __shfl_xor(omx, 2, 64));
    const float sc = omx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / omx)), 100.f)) : 1.f;
    if (q == 0) rinv_dst[row] = 1.0f / sc;
    const int a_dst = a_slot(row, q) * 16;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      const f32x4 a = dv[2 * s6], b = dv[2 * s6 + 1];
      unsigned hh[4], ll[4];
      split2_pair(a[0] * sc, a[1] * sc, hh[0], ll[0]);
      split2_pair(a[2] * sc, a[3] * sc, hh[1], ll[1]);
      split2_pair(b[0] * sc, b[1] * sc, hh[2], ll[2]);
      split2_pair(b[2] * sc, b[3] * sc, hh[3], ll[3]);
      unsigned char* sa = smem + s6 * AST + a_dst;
      *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
      *(u32x4*)(sa + APL) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };

  if (BWD && p.W0) {
    // ---------------- front product (backward, optional): the dy rows of this MLP are themselves a Linear's data
    // gradient + LayerNorm backward -- dy = res0 + LN_bwd(X0 . W0^T; x0, stats0), the qkv Linear of the Swin block behind
    // this one -- computed here instead of read: k_nth2's pass loop (192 k per pass, running row exponent), the
    // LayerNorm backward on the row-major tile, and the rows go from registers into GEMM 1's stage images.
    float* const rscale0 = smax;                     // [64] rescale factor of the pass
    float* const rinv0 = smax + 64;                  // [64] 2^-s of the X0 rows so far
    const int nst0 = p.Kp0 / SK;
    const long plane0 = (long)p.C * p.Kp0 * 2;
    const float* const winv0 = (const float*)((const char*)p.W0 + 2 * plane0);
    auto load_b0 = [&](int cs, u32x4 (&fb)[3][2]) {
      const int pass6 = (cs / 6) * 6, np = min(6, nst0 - pass6);        // a pass of the front product: 6 stages, the last one fewer
      int x = cs - pass6 + (np > 0 ? rot % np : 0);
      if (x >= np) x -= np;
      const char* base = (const char*)p.W0 + (long)(2 * min(pass6 + x, nst0 - 1)) * p.C * 32;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane0 + boff2[jt]);
    };
    load_b0(0, fb0);
    load_b0(1, fb1);
    load_b0(2, fb2);
    f32x4 acc0[4][3];
    const int arow = tid >> 2, akq = tid & 3;
    const char* const abase = (const char*)p.X0 + (long)min(m0 + arow, p.M - 1) * p.ld0 * 4;
    const int a_dst = a_slot(arow, akq) * 16;
    float asc = 3.0e38f;                             // this row's current 2^s (the same in the row's four threads)
    const int npass = (nst0 + 5) / 6;
    for (int pass = 0; pass < npass; ++pass) {
      const int cs0 = pass * 6;
      if (pass) __syncthreads();                     // every wave is done with the previous pass's images
      {
        f32x4 ra[6][2];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          const int k = (cs0 + s6) * SK + akq * 8;
          ra[s6][0] = *(const f32x4*)(abase + (k < p.K0 ? k * 4 : 0));
          ra[s6][1] = *(const f32x4*)(abase + (k + 4 < p.K0 ? (k + 4) * 4 : 0));
        }
        float mx = 0.f;
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          const int k = (cs0 + s6) * SK + akq * 8;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x4 x = ra[s6][e];
            if (k + 4 * e >= p.K0) x = f32x4{0.f, 0.f, 0.f, 0.f};
            ra[s6][e] = x;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
          }
        }
        const float old = asc;
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        const float need = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 3.0e38f;
        asc = fminf(asc, need);
        const float use = asc > 1.0e38f ? 1.f : asc;
        if (akq == 0) {
          rscale0[arow] = (old > 1.0e38f || old == asc) ? 1.f : asc / old;
          rinv0[arow] = 1.0f / use;
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
          unsigned char* sa = smem + s6 * AST + a_dst;
          *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
          *(u32x4*)(sa + APL) = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
      }
      __syncthreads();
      if (pass == 0) { SR_TS(22) }
      if (pass == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {                                       // rows whose scale dropped: bring what is accumulated to the new scale
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = rscale0[16 * i + 4 * g + e];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc0[i][j][e] *= f;
          }
      }
      const int np0 = min(6, nst0 - cs0), rot0 = rot % np0;
      auto mma0 = [&](int s6, const u32x4 (&fb)[3][2]) {
        int x6 = s6 + rot0;
        if (x6 >= np0) x6 -= np0;
        const unsigned char* sa = smem + x6 * AST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 fa[2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc0[i][j] = mfma16h(fa[PA], fb[j][PB], acc0[i][j]);
          SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
        }
      };
#define SR_STAGE(S, FB)                                              \
      if (cs0 + (S) < nst0) {                                        \
        mma0((S), FB);                                               \
        if (cs0 + (S) + 3 < nst0) load_b0(cs0 + (S) + 3, FB);        \
        __builtin_amdgcn_sched_barrier(0);                           \
      }
      SR_STAGE(0, fb0) SR_STAGE(1, fb1) SR_STAGE(2, fb2) SR_STAGE(3, fb0) SR_STAGE(4, fb1) SR_STAGE(5, fb2)
#undef SR_STAGE
      if (pass == 0) { SR_TS(23) }
    }
    SR_TS(15)
    __syncthreads();
    {
      float wv[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) wv[j] = winv0[min(wave * 48 + 16 * j + c, p.C - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ri = rinv0[16 * i + 4 * g + e];
#pragma unroll
          for (int j = 0; j < 3; ++j) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc0[i][j][e] * (ri * wv[j]);
        }
    }
    __syncthreads();
    SR_TS(16)
    f32x4 dv[12];
    const float omx = ln_bwd_rows(p.x0, p.ldx0, p.stats0, p.res0, p.ldres0, p.out0, p.ldo0, dv);
    SR_TS(17)
    __syncthreads();                                 // every thread has read its part of the tile
    rows_to_images(dv, omx, rinvx);
    load_b1(0, fb0);
    load_b1(1, fb1);
    load_b1(2, fb2);
  } else {
  load_b1(0, fb0);
  load_b1(1, fb1);
  load_b1(2, fb2);

  // ---------------- the X rows: one pass of 192 k (K1 <= 192), split into six stage images
  {
    const int arow = tid >> 2, akq = tid & 3;
    const int agm = min(m0 + arow, p.M - 1);
    const char* const abase = (const char*)p.X + (long)agm * p.ldx * 4;
    const float2 rst = ldg_f2(BWD ? k_sr_neutral : p.ln_stats + 2 * agm);
    f32x4 ra[6][2];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {                 // past the end: k = 0 of the row, zeroed below
      const int k = s6 * SK + akq * 8;
      ra[s6][0] = *(const f32x4*)(abase + (k < p.K1 ? k * 4 : 0));
      ra[s6][1] = *(const f32x4*)(abase + (k + 4 < p.K1 ? (k + 4) * 4 : 0));
    }
    float mx = 0.f;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      const int k = s6 * SK + akq * 8;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4 x = ra[s6][e];
        if (!BWD) { x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y; }
        if (k + 4 * e >= p.K1) x = f32x4{0.f, 0.f, 0.f, 0.f};
        ra[s6][e] = x;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
      }
    }
    float asc;
    if (BWD) {                                       // gradient rows: the block exponent comes from the row maximum
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      asc = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 1.f;
    } else {                                         // behind the LayerNorm prologue |xhat| <= sqrt(K): a priori
      asc = exp2f(floorf(log2f(16384.f * rsqrtf((float)p.K1))));
    }
    if (akq == 0) rinvx[arow] = 1.0f / asc;
    const int a_dst = a_slot(arow, akq) * 16;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      unsigned hh[4], ll[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x4 x = ra[s6][e];
        split2_pair(x.x * asc, x.y * asc, hh[2 * e], ll[2 * e]);
        split2_pair(x.z * asc, x.w * asc, hh[2 * e + 1], ll[2 * e + 1]);
      }
      unsigned char* sa = smem + s6 * AST + a_dst;
      *(u32x4*)(sa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
      if (!AMP) *(u32x4*)(sa + APL) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  }
  }
  SR_TS(1)
  __syncthreads();
  SR_TS(2)

  // ---------------- GEMM 1, transposed: acc1[hh][i][j] = hidden units 192 hh + 48 wave + 16 j + 4 g + e of token 16 i + c
  f32x4 acc1[2][4][3];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc1[hh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma1 = [&](f32x4 (&acc)[4][3], int s6, const u32x4 (&fb)[3][2]) {
    const unsigned char* sa = smem + rs6(s6) * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16h(fb[j][PB], fa[PA], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)          // small terms first
      }
#undef SR_TERM
    }
  };
#define SR_G1(U, FB)                                                   \
  if ((U) / 6 < NH) {                                                  \
    mma1(acc1[(U) / 6], (U) % 6, FB);                                  \
    if ((U) + 3 < 6 * NH) load_b1((U) + 3, FB);                        \
  }
  SR_G1(0, fb0) SR_G1(1, fb1) SR_G1(2, fb2) SR_G1(3, fb0) SR_G1(4, fb1) SR_G1(5, fb2)
  SR_G1(6, fb0) SR_G1(7, fb1) SR_G1(8, fb2) SR_G1(9, fb0) SR_G1(10, fb1) SR_G1(11, fb2)
#undef SR_G1
  SR_TS(3)
  // the first stages of GEMM 2's weights travel while the activation math runs (the backward's gate needs the
  // registers: it requests them behind the math, in front of the token-maximum exchange)
  if (!BWD) {
    load_b2(0, fb0);
    load_b2(1, fb1);
    load_b2(2, fb2);
  }

  // ---------------- between the GEMMs: bias + GELU (forward) / the GELU gate (backward), in registers
  float tmax[4] = {0.f, 0.f, 0.f, 0.f};
  {
    float rix[4], dps[4];
    int gm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rix[i] = rinvx[16 * i + c];
      gm[i] = m0 + 16 * i + c;
      dps[i] = 1.f;
      if (BWD && p.rowscale) dps[i] = p.rowscale[min(gm[i], p.M - 1) / p.rows_per_scale];
    }
    // backward: the pre-activations of group (half, j) travel while the two groups in front of it are processed: eight
    // 16-byte loads in flight per lane (round 4 kept four: the phase ran at the latency of six dependent HBM round trips)
    f32x4 hbuf[3][4];
    auto load_h = [&](int grp, f32x4 (&hv)[4]) {
      const int unit0 = min(192 * (grp / 3) + wave * 48 + 16 * (grp % 3) + 4 * g, p.hid - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[i] = *(const f32x4*)(p.H + (long)min(gm[i], p.M - 1) * p.ldh + unit0);
    };
    if (BWD) {
      load_h(0, hbuf[0]);
      load_h(1, hbuf[1]);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh < NH) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (BWD) {
            if (3 * hh + j + 2 < 3 * NH) load_h(3 * hh + j + 2, hbuf[(3 * hh + j + 2) % 3]);
          }
          f32x4 (&hcur)[4] = hbuf[(3 * hh + j) % 3];
          const int unit0 = 192 * hh + wave * 48 + 16 * j + 4 * g;
          const bool uok = unit0 < p.hid;            // hid % 4 == 0: a lane's four units are valid or not together
          const int uc = min(unit0, p.hid - 4);
          const f32x4 wi = *(const f32x4*)(winv1 + uc);
          f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (!BWD) bv = *(const f32x4*)(p.b1 + uc);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 v = acc1[hh][i][j];
            const bool ok = uok && gm[i] < p.M;
            if (!BWD) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] * (rix[i] * wi[e]) + bv[e];
              if (p.H && ok) SR_ST_STREAM((f32x4*)(p.H + (long)gm[i] * p.ldh + unit0), v);
              {
                const sr_f32x2 g0 = gelu_fast2(sr_f32x2{v[0], v[1]}), g1 = gelu_fast2(sr_f32x2{v[2], v[3]});
                v = uok ? f32x4{g0.x, g0.y, g1.x, g1.y} : f32x4{0.f, 0.f, 0.f, 0.f};
              }
            } else {
              const f32x4 hx = hcur[i];
              f32x4 gl;
              {
                sr_f32x2 ga, gb, la, lb;
                gelu_gate2(sr_f32x2{hx[0], hx[1]}, ga, la);
                gelu_gate2(sr_f32x2{hx[2], hx[3]}, gb, lb);
                gl = f32x4{la.x, la.y, lb.x, lb.y};
                const float gt[4] = {ga.x, ga.y, gb.x, gb.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = uok ? v[e] * (rix[i] * wi[e] * dps[i]) * gt[e] : 0.f;
              }
              if (ok) {
                SR_ST_STREAM((f32x4*)(p.dH + (long)gm[i] * p.ldh + unit0), v);
                if (p.GH) SR_ST_STREAM((f32x4*)(p.GH + (long)gm[i] * p.ldh + unit0), gl);     // NULL: the weight gradient recomputes it from h
              }
            }
            acc1[hh][i][j] = v;
            tmax[i] = fmaxf(fmaxf(tmax[i], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          }
        }
      }
    }
  }
  if (BWD) {
    load_b2(0, fb0);
    load_b2(1, fb1);
    load_b2(2, fb2);
  }
  SR_TS(4)
  // token maxima over the whole hidden row: lanes g = 0..3 of a wave, then the four waves
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    tmax[i] = fmaxf(tmax[i], __shfl_xor(tmax[i], 16, 64));
    tmax[i] = fmaxf(tmax[i], __shfl_xor(tmax[i], 32, 64));
    if (g == 0) smax[wave * 64 + 16 * i + c] = tmax[i];
  }
  __syncthreads();                                   // also: every wave is done with the x stage images
  SR_TS(5)
  float use[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 16 * i + c;
    const float mx = fmaxf(fmaxf(smax[t], smax[64 + t]), fmaxf(smax[128 + t], smax[192 + t]));
    use[i] = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 1.f;
    if (wave == 0 && g == 0) rinv2[t] = 1.0f / use[i];
  }

  // ---------------- GEMM 2 (k_nth2's loop): acc2[i][j] = rows 16 i + 4 g + e, columns 48 wave + 16 j + c
  f32x4 acc2[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage_a2 = [&](const f32x4 (&acc)[4][3]) {   // a lane's 4 units = half an octet: one 8-byte store per plane
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kk = wave * 48 + 16 * j + 4 * g;
      const int s6 = kk >> 5, u = (kk & 31) >> 3, pos = kk & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v = acc[i][j];
        unsigned h0, l0, h1, l1;
        split2_pair(v[0] * use[i], v[1] * use[i], h0, l0);
        split2_pair(v[2] * use[i], v[3] * use[i], h1, l1);
        unsigned char* sa = smem + s6 * AST + a_slot(16 * i + c, u) * 16 + pos * 2;
        *(u32x2*)(sa) = u32x2{h0, h1};
        if (!AMP) *(u32x2*)(sa + APL) = u32x2{l0, l1};
      }
    }
  };
  auto mma2 = [&](int s6, const u32x4 (&fb)[3][2]) {
    const unsigned char* sa = smem + rs6(s6) * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc2[i][j] = mfma16h(fa[PA], fb[j][PB], acc2[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };
#define SR_G2(U, FB)                                                   \
  {                                                                    \
    mma2((U) % 6, FB);                                                 \
    if ((U) + 3 < 6 * NH) load_b2((U) + 3, FB);                        \
  }
  stage_a2(acc1[0]);
  SR_TS(6)
  __syncthreads();
  SR_TS(7)
  SR_G2(0, fb0) SR_G2(1, fb1) SR_G2(2, fb2) SR_G2(3, fb0) SR_G2(4, fb1) SR_G2(5, fb2)
  SR_TS(8)
  if (NH > 1) {
    __syncthreads();                                 // every wave is done with the first pass's images
    SR_TS(9)
    stage_a2(acc1[1]);
    __syncthreads();
    SR_TS(10)
    SR_G2(6, fb0) SR_G2(7, fb1) SR_G2(8, fb2) SR_G2(9, fb0) SR_G2(10, fb1) SR_G2(11, fb2)
  }
#undef SR_G2
  SR_TS(11)

  // ---------------- the output tile, row-major in LDS (block exponents undone: exact powers of two)
  __syncthreads();
  SR_TS(12)
  {
    float wv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wv[j] = winv2[min(wave * 48 + 16 * j + c, p.N2 - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ri = rinv2[16 * i + 4 * g + e];
#pragma unroll
        for (int j = 0; j < 3; ++j) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc2[i][j][e] * (ri * wv[j]);
      }
  }
  __syncthreads();
  SR_TS(13)
  const int C = p.C;
  if (!BWD) {
    // out = x + s (acc + b2): 16-byte pieces in row-major order (a wave instruction covers 1 KB of consecutive tile bytes)
    f32x4 rv[12];
    int prow[12], pcol[12];
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const int idx = it * 256 + tid;
      prow[it] = idx / 48;
      pcol[it] = (idx - prow[it] * 48) * 4;
      const int gm = m0 + prow[it];
      const bool ok = gm < p.M && pcol[it] < C;
      rv[it] = ok ? *(const f32x4*)(p.R + (long)gm * p.ldr + pcol[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (!ok) prow[it] = -1;
    }
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      if (prow[it] >= 0) {
        const int gm = m0 + prow[it];
        float* tp = T + prow[it] * TP + pcol[it];
        f32x4 v = *(const f32x4*)tp;
        const f32x4 bv = p.b2 ? *(const f32x4*)(p.b2 + pcol[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float s = p.rowscale ? p.rowscale[gm / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + bv[e]) * s + rv[it][e];
        *(f32x4*)(p.out + (long)gm * p.ldo + pcol[it]) = v;
        if (p.stats_out) *(f32x4*)tp = v;
      }
    }
    if (p.stats_out) {
      // {mean, rstd} of the out rows for the next LayerNorm (two-pass, eps 1e-5, biased variance): four lanes per row
      __syncthreads();
      const int row = tid >> 2, q = tid & 3;
      f32x4 xv[12];
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        xv[k] = *(const f32x4*)(T + row * TP + q * 48 + 4 * k);
        if (q * 48 + 4 * k < C) s1 += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
      }
      s1 += __shfl_xor(s1, 1, 64);
      s1 += __shfl_xor(s1, 2, 64);
      const float mean = s1 * (1.0f / (float)C);
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k)
        if (q * 48 + 4 * k < C) {
          const float d0 = xv[k].x - mean, d1 = xv[k].y - mean, d2 = xv[k].z - mean, d3 = xv[k].w - mean;
          s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      s2 += __shfl_xor(s2, 1, 64);
      s2 += __shfl_xor(s2, 2, 64);
      if (q == 0 && m0 + row < p.M)
        *(float2*)(p.stats_out + 2 * (long)(m0 + row)) = float2{mean, rsqrtf(s2 * (1.0f / (float)C) + 1e-5f)};
    }
  } else {
    // dx = dy + LayerNorm_backward(dxh; x, stats)  (the affine folded into W1)
    f32x4 dv[12];
    float omx = ln_bwd_rows(p.R, p.ldr, p.ep_stats, p.R2, p.ldr2, p.out, p.ldo, dv);
    SR_TS(18)
    if (p.W3) {
      // ---------------- chained third product: out3 = s3 (dx . W3^T), the data gradient of the Linear in front of this
      // block's residual (the attention's proj): dx goes from the registers that hold it into the stage images
      u32x4 fc0[3][2], fc1[3][2], fc2[3][2];
      const long plane3 = (long)p.C * p.Kp1 * 2;
      const float* const winv3 = (const float*)((const char*)p.W3 + 2 * plane3);
      auto load_b3 = [&](int cs, u32x4 (&fb)[3][2]) {
        const char* base = (const char*)p.W3 + (long)(2 * min(rs6(cs), nst1 - 1)) * p.C * 32;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane3 + boff2[jt]);
      };
      load_b3(0, fc0);
      load_b3(1, fc1);
      load_b3(2, fc2);
      __syncthreads();                               // every thread has read its part of the tile
      rows_to_images(dv, omx, rinv3);
      __syncthreads();
      SR_TS(19)
      f32x4 acc3[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto mma3 = [&](int s6, const u32x4 (&fb)[3][2]) {
        const unsigned char* sa = smem + rs6(s6) * AST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 fa[2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc3[i][j] = mfma16h(fa[PA], fb[j][PB], acc3[i][j]);
          SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
        }
      };
      mma3(0, fc0); load_b3(3, fc0); __builtin_amdgcn_sched_barrier(0);
      mma3(1, fc1); load_b3(4, fc1); __builtin_amdgcn_sched_barrier(0);
      mma3(2, fc2); load_b3(5, fc2); __builtin_amdgcn_sched_barrier(0);
      mma3(3, fc0); __builtin_amdgcn_sched_barrier(0);
      mma3(4, fc1); __builtin_amdgcn_sched_barrier(0);
      mma3(5, fc2);
      SR_TS(20)
      __syncthreads();                               // every wave is done with the images
      {
        float wv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) wv[j] = winv3[min(wave * 48 + 16 * j + c, p.C - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ri = rinv3[16 * i + 4 * g + e];
#pragma unroll
            for (int j = 0; j < 3; ++j) T[(16 * i + 4 * g + e) * TP + wave * 48 + 16 * j + c] = acc3[i][j][e] * (ri * wv[j]);
          }
      }
      __syncthreads();
      SR_TS(21)
#pragma unroll
      for (int it = 0; it < 12; ++it) {
        const int idx = it * 256 + tid, prow = idx / 48, pcol = (idx - prow * 48) * 4;
        const int gm3 = m0 + prow;
        if (gm3 < p.M && pcol < C) {
          f32x4 v = *(const f32x4*)(T + prow * TP + pcol);
          const float s3 = p.rowscale3 ? p.rowscale3[gm3 / p.rows_per_scale] : 1.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= s3;
          *(f32x4*)(p.out3 + (long)gm3 * p.ld3 + pcol) = v;
        }
      }
    }
  }
  SR_TS(14)
}


// =====================================================================================================================
// Round 5: the same forward with the memory side and the matrix side in DIFFERENT waves.
//
// What round 4's stamps could not show and a one-block-per-CU run did (tools/mb_mlp_stagger.py): a 64-token block ALONE on
// its CU takes 32 us of the 42 us two co-resident blocks take -- the kernel is a chain of dependent phases, each waiting
// for its own loads, and its partner block only shares that wait.  Putting the partner in another phase by a start delay
// gains 3 % (the chain is as long as before).  What shortens the chain: all eight matrix waves of the CU on ONE tile (each
// phase half as long), and the row traffic -- x rows in, LayerNorm, the split into stage images; the residual, out rows
// and row statistics out -- in four waves of their own that work on the OTHER tile of the CU meanwhile.  Their loads and
// stores never sit in front of a weight fragment in a matrix wave's in-order memory counter.
//
// One block per CU: 12 waves (three per SIMD, 168 registers), two 64-token tiles.  G = waves 0..7, R = waves 8..11.  A
// static schedule of slots, one LDS-only barrier between slots, the same barrier count on every path (no flags, no spins):
//
//   slot   G (matrix waves)                                       R (row waves)
//    0     W1 fragments requested                                 x rows (t0) -> LN -> images A   (raw rows stay in registers)
//    1     GEMM 1 (t0, images A), bias + GELU, h rows out, maxima x rows (t1) -> images B
//    2     hidden rows -> images: half 0 -> C, half 1 -> A
//    3     GEMM 2 (t0): waves 0..3 over half 0 (C), 4..7 over half 1 (A)
//    4     partial output tiles -> C, A
//    5     GEMM 1 (t1, images B), ...                             out rows (t0) = x + s (C + A + b2), row statistics
//    6..8  as 2..4 for t1
//    9                                                            out rows (t1)
//
// GEMM 1: wave (hh, w) owns hidden units 192 hh + 48 w .. + 47 of all 64 tokens (what wave w of k_mlp_f16 held in acc1[hh]:
// the same accumulators, the same h bits).  GEMM 2 is split over K between the two wave quartets (a wave's fragments are
// used for four row tiles: no weight byte is requested twice by the CU); the two partial tiles meet in the row waves.
constexpr int GR_NT = 768;
constexpr int GR_SMALL = (8 * 64 + 2 * 64 + 64) * 4;          // smax [8][64], rinvx [2][64], rinv2 [64]
constexpr int GR_LDS = 3 * R0 + GR_SMALL;
static_assert(GR_LDS <= 160 * 1024, "LDS of one CU");

#ifdef SRHIP_EXPERIMENTS
#define GR_TS(K) \
  if (p.dbg && lane == 0) p.dbg[((long)blockIdx.x * 12 + wave) * 32 + (K)] = (long long)wall_clock64();
#else
#define GR_TS(K)
#endif

__global__ void __launch_bounds__(GR_NT, 3) k_mlp_gr_fwd(MlpF16Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const RA = smem;
  unsigned char* const RB = smem + R0;
  unsigned char* const RC = smem + 2 * R0;
  float* const smax = (float*)(smem + 3 * R0);       // [8 waves][64 tokens] maxima of |gelu(h)|
  float* const rinvx = smax + 8 * 64;                // [2 tiles][64] 2^-s of the LN(x) rows
  float* const rinv2 = rinvx + 2 * 64;               // [64] 2^-s of the hidden rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool isG = wave < 8;
  const int c = lane & 15, g = lane >> 4;
  const int ntiles = (p.M + BM - 1) / BM;
  const int tile0 = 2 * sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int nt = min(2, ntiles - tile0);             // tiles of this block (>= 1)
  const int nst1 = p.Kp1 / SK, nst2 = p.Kp2 / SK;
  const int C = p.C;
  GR_TS(0)

  // ------------------------------------------------------------------ matrix-wave state
  const int hh = (wave >> 2) & 1, w4 = wave & 3;     // G: hidden half and 48-wide slice
  const long plane1 = (long)p.N1 * p.Kp1 * 2, plane2 = (long)p.N2 * p.Kp2 * 2;
  const float* const winv1 = (const float*)((const char*)p.W1 + 2 * plane1);
  const float* const winv2 = (const float*)((const char*)p.W2 + 2 * plane2);
  unsigned boff1[3], boff2[3];
  int a_off[4];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    const int row = min(192 * hh + w4 * 48 + jt * 16 + c, p.N1 - 1);
    boff1[jt] = (unsigned)(((g >> 1) * p.N1 + row) * 32 + (g & 1) * 16);
    const int col = min(w4 * 48 + jt * 16 + c, p.N2 - 1);
    boff2[jt] = (unsigned)(((g >> 1) * p.N2 + col) * 32 + (g & 1) * 16);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = a_slot(16 * i + c, g) * 16;
  auto load_b1 = [&](int s6, u32x4 (&fb)[3][2]) {
    const char* base = (const char*)p.W1 + (long)(2 * min(s6, nst1 - 1)) * p.N1 * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane1 + boff1[jt]);
  };
  auto load_b2 = [&](int s6, u32x4 (&fb)[3][2]) {     // stage 6 hh + s6 of the hidden contraction
    const char* base = (const char*)p.W2 + (long)(2 * min(6 * hh + s6, nst2 - 1)) * p.N2 * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane2 + boff2[jt]);
  };
  u32x4 fb0[3][2], fb1[3][2];                         // two weight register sets (168 registers per wave): stage s in set s & 1
  f32x4 acc[4][3];                                   // GEMM 1: units 192 hh + 48 w4 + 16 j + 4 g + e of token 16 i + c;
                                                     // GEMM 2: rows 16 i + 4 g + e, columns 48 w4 + 16 j + c
  float use[4];
  // A fragments from `img`, weights on the ROW side (transposed product)
  auto mma_t = [&](const unsigned char* img, int s6, const u32x4 (&fb)[3][2]) {
    const unsigned char* sa = img + s6 * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16h(fb[j][PB], fa[PA], acc[i][j]);
      SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
    }
  };
  auto mma_n = [&](const unsigned char* img, int s6, const u32x4 (&fb)[3][2]) {
    const unsigned char* sa = img + s6 * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[i][j] = mfma16h(fa[PA], fb[j][PB], acc[i][j]);
      SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
    }
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // slot 1 / 5: GEMM 1 of tile t over the x images, bias + GELU in registers, h rows out, token maxima
  auto g_gemm1_act = [&](int t, const unsigned char* img) {
    const int m0 = (tile0 + t) * BM;
    zero_acc();
    mma_t(img, 0, fb0); load_b1(2, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_t(img, 1, fb1); load_b1(3, fb1); __builtin_amdgcn_sched_barrier(0);
    mma_t(img, 2, fb0); load_b1(4, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_t(img, 3, fb1); load_b1(5, fb1); __builtin_amdgcn_sched_barrier(0);
    mma_t(img, 4, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_t(img, 5, fb1);
    float tmax[4] = {0.f, 0.f, 0.f, 0.f};
    float rix[4];
    int gm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rix[i] = rinvx[64 * t + 16 * i + c];
      gm[i] = m0 + 16 * i + c;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int unit0 = 192 * hh + w4 * 48 + 16 * j + 4 * g;
      const bool uok = unit0 < p.hid;                  // hid % 4 == 0: a lane's four units are valid or not together
      const int uc = min(unit0, p.hid - 4);
      const f32x4 wi = *(const f32x4*)(winv1 + uc);
      const f32x4 bv = *(const f32x4*)(p.b1 + uc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * (rix[i] * wi[e]) + bv[e];
        if (p.H && uok && gm[i] < p.M) *(f32x4*)(p.H + (long)gm[i] * p.ldh + unit0) = v;
        const sr_f32x2 g0 = gelu_fast2(sr_f32x2{v[0], v[1]}), g1 = gelu_fast2(sr_f32x2{v[2], v[3]});
        v = uok ? f32x4{g0.x, g0.y, g1.x, g1.y} : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][j] = v;
        tmax[i] = fmaxf(fmaxf(tmax[i], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
    // GEMM 2's first weight stages travel through the barrier and the staging slot behind it (requested here, not in front
    // of the activation math: 72 registers the math needs)
    __builtin_amdgcn_sched_barrier(0);
    load_b2(0, fb0);
    load_b2(1, fb1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tmax[i] = fmaxf(tmax[i], __shfl_xor(tmax[i], 16, 64));
      tmax[i] = fmaxf(tmax[i], __shfl_xor(tmax[i], 32, 64));
      if (g == 0) smax[wave * 64 + 16 * i + c] = tmax[i];
    }
  };
  // slot 2 / 6: the hidden rows under their block exponent (token maximum over the eight waves) -> images of half hh
  auto g_stage_a2 = [&](unsigned char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 16 * i + c;
      float mx = fmaxf(fmaxf(smax[t], smax[64 + t]), fmaxf(smax[128 + t], smax[192 + t]));
      mx = fmaxf(mx, fmaxf(fmaxf(smax[256 + t], smax[320 + t]), fmaxf(smax[384 + t], smax[448 + t])));
      use[i] = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 1.f;
      if (wave == 0 && g == 0) rinv2[t] = 1.0f / use[i];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kk = w4 * 48 + 16 * j + 4 * g;
      const int s6 = kk >> 5, u = (kk & 31) >> 3, pos = kk & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v = acc[i][j];
        unsigned h0, l0, h1, l1;
        split2_pair(v[0] * use[i], v[1] * use[i], h0, l0);
        split2_pair(v[2] * use[i], v[3] * use[i], h1, l1);
        unsigned char* sa = img + s6 * AST + a_slot(16 * i + c, u) * 16 + pos * 2;
        *(u32x2*)(sa) = u32x2{h0, h1};
        *(u32x2*)(sa + APL) = u32x2{l0, l1};
      }
    }
  };
  // slot 3 / 7: this quartet's half of GEMM 2 (six stages of its own images); then the next tile's W1 stages are requested
  auto g_gemm2 = [&](const unsigned char* img) {
    zero_acc();
    mma_n(img, 0, fb0); load_b2(2, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_n(img, 1, fb1); load_b2(3, fb1); __builtin_amdgcn_sched_barrier(0);
    mma_n(img, 2, fb0); load_b2(4, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_n(img, 3, fb1); load_b2(5, fb1); __builtin_amdgcn_sched_barrier(0);
    mma_n(img, 4, fb0); __builtin_amdgcn_sched_barrier(0);
    mma_n(img, 5, fb1);
  };
  // slot 4 / 8: the partial output tile, row-major (block exponents undone: exact powers of two)
  auto g_tile_out = [&](unsigned char* img, bool more) {
    float* const T = (float*)img;
    float wv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wv[j] = winv2[min(w4 * 48 + 16 * j + c, p.N2 - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ri = rinv2[16 * i + 4 * g + e];
#pragma unroll
        for (int j = 0; j < 3; ++j) T[(16 * i + 4 * g + e) * TP + w4 * 48 + 16 * j + c] = acc[i][j][e] * (ri * wv[j]);
      }
    if (more) {                                      // the next tile's first W1 stages travel through the barrier
      __builtin_amdgcn_sched_barrier(0);
      load_b1(0, fb0);
      load_b1(1, fb1);
    }
  };

  // ------------------------------------------------------------------ row-wave state
  const int rtid = tid - 512, rrow = (rtid >> 2) & 63, rq = rtid & 3;
  f32x4 xr0[6][2], xr1[6][2];                        // the raw x rows of the two tiles: the residual of the epilogue
  // slot 0 / 1: x rows of tile t -> LayerNorm -> split -> stage images (one pass of 192 k)
  auto r_stage = [&](int t, unsigned char* img, f32x4 (&xr)[6][2]) {
    const int m0 = (tile0 + t) * BM;
    const int agm = min(m0 + rrow, p.M - 1);
    const char* const abase = (const char*)p.X + (long)agm * p.ldx * 4;
    const float2 rst = ldg_f2(p.ln_stats + 2 * agm);
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {                 // past the end: k = 0 of the row, zeroed below
      const int k = s6 * SK + rq * 8;
      xr[s6][0] = *(const f32x4*)(abase + (k < p.K1 ? k * 4 : 0));
      xr[s6][1] = *(const f32x4*)(abase + (k + 4 < p.K1 ? (k + 4) * 4 : 0));
    }
    const float asc = exp2f(floorf(log2f(16384.f * rsqrtf((float)p.K1))));      // |xhat| <= sqrt(K): a priori
    if (rq == 0) rinvx[64 * t + rrow] = 1.0f / asc;
    const int a_dst = a_slot(rrow, rq) * 16;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      const int k = s6 * SK + rq * 8;
      unsigned hv[4], lv[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4 x = xr[s6][e];
        x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y;
        if (k + 4 * e >= p.K1) x = f32x4{0.f, 0.f, 0.f, 0.f};
        split2_pair(x.x * asc, x.y * asc, hv[2 * e], lv[2 * e]);
        split2_pair(x.z * asc, x.w * asc, hv[2 * e + 1], lv[2 * e + 1]);
      }
      unsigned char* sa = img + s6 * AST + a_dst;
      *(u32x4*)(sa) = u32x4{hv[0], hv[1], hv[2], hv[3]};
      *(u32x4*)(sa + APL) = u32x4{lv[0], lv[1], lv[2], lv[3]};
    }
  };
  // slot 5 / 9: out = x + s (T0 + T1 + b2) and the {mean, rstd} of the out rows (two-pass, eps 1e-5, biased variance): a
  // thread owns the octets (stage s6, octet rq) of its row, as it loaded them
  auto r_epilogue = [&](int t, const unsigned char* i0, const unsigned char* i1, f32x4 (&xr)[6][2]) {
    const int m0 = (tile0 + t) * BM;
    const int gm = m0 + rrow;
    const bool rok = gm < p.M;
    const float* const T0 = (const float*)i0;
    const float* const T1 = (const float*)i1;
    const float s = p.rowscale ? p.rowscale[min(gm, p.M - 1) / p.rows_per_scale] : 1.f;
    float s1 = 0.f;                                  // (the out row replaces the x row in the thread's registers)
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = s6 * SK + rq * 8 + 4 * e;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < C) {
          const f32x4 a = *(const f32x4*)(T0 + rrow * TP + col), b = *(const f32x4*)(T1 + rrow * TP + col);
          const f32x4 bv = p.b2 ? *(const f32x4*)(p.b2 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ((a[q] + b[q]) + bv[q]) * s + xr[s6][e][q];
          if (rok) *(f32x4*)(p.out + (long)gm * p.ldo + col) = v;
          s1 += (v[0] + v[1]) + (v[2] + v[3]);
        }
        xr[s6][e] = v;
      }
    if (p.stats_out) {
      s1 += __shfl_xor(s1, 1, 64);
      s1 += __shfl_xor(s1, 2, 64);
      const float mean = s1 * (1.0f / (float)C);
      float s2 = 0.f;
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (s6 * SK + rq * 8 + 4 * e < C) {
            const f32x4 v = xr[s6][e];
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
      s2 += __shfl_xor(s2, 1, 64);
      s2 += __shfl_xor(s2, 2, 64);
      if (rq == 0 && rok) *(float2*)(p.stats_out + 2 * (long)gm) = float2{mean, rsqrtf(s2 * (1.0f / (float)C) + 1e-5f)};
    }
  };

  // ------------------------------------------------------------------ the schedule: one straight-line program per role
  // (a role's registers are live only inside its own branch), the same number of barriers on both
  const bool two = nt > 1;
  if (isG) {
    load_b1(0, fb0);                                 // slot 0
    load_b1(1, fb1);
    GR_TS(1)
    sr_lds_barrier();
    GR_TS(2)
    g_gemm1_act(0, RA);                              // slot 1
    GR_TS(3)
    sr_lds_barrier();
    GR_TS(4)
    g_stage_a2(hh ? RA : RC);                        // slot 2
    GR_TS(5)
    sr_lds_barrier();
    GR_TS(6)
    g_gemm2(hh ? RA : RC);                           // slot 3
    GR_TS(7)
    sr_lds_barrier();
    GR_TS(8)
    g_tile_out(hh ? RA : RC, two);                   // slot 4
    GR_TS(9)
    sr_lds_barrier();
    GR_TS(10)
    if (!two) return;
    g_gemm1_act(1, RB);                              // slot 5
    GR_TS(11)
    sr_lds_barrier();
    GR_TS(12)
    g_stage_a2(hh ? RA : RC);                        // slot 6
    GR_TS(13)
    sr_lds_barrier();
    GR_TS(14)
    g_gemm2(hh ? RA : RC);                           // slot 7
    GR_TS(15)
    sr_lds_barrier();
    GR_TS(16)
    g_tile_out(hh ? RA : RC, false);                 // slot 8
    GR_TS(17)
    sr_lds_barrier();
  } else {
    r_stage(0, RA, xr0);                             // slot 0
    GR_TS(1)
    sr_lds_barrier();
    GR_TS(2)
    if (two) r_stage(1, RB, xr1);                    // slot 1
    GR_TS(3)
    sr_lds_barrier();
    sr_lds_barrier();                                // slots 2, 3, 4: nothing for the row waves
    sr_lds_barrier();
    sr_lds_barrier();
    GR_TS(10)
    r_epilogue(0, RC, RA, xr0);                      // slot 5
    GR_TS(11)
    if (!two) return;
    sr_lds_barrier();
    sr_lds_barrier();                                // slots 6, 7, 8
    sr_lds_barrier();
    sr_lds_barrier();
    GR_TS(18)
    r_epilogue(1, RC, RA, xr1);                      // slot 9
    GR_TS(19)
  }
}

}  // namespace

#ifdef SRHIP_EXPERIMENTS
SR_DEBUG_EXPORT int srhip_mlp_debug_buffer(long long* buf) { g_mlp_dbg = buf; return 0; }    // [blocks][4][32] wall-clock stamps
SR_DEBUG_EXPORT int srhip_mlp_cu_counters(int* buf) { g_mlp_cu_count = buf; return 0; }      // [4096] zeroed ints: arrival order per CU
#endif

int sr_mlp_f16(MlpF16Args& p, int bwd, hipStream_t st) {
#ifdef SRHIP_EXPERIMENTS
  p.dbg = g_mlp_dbg;
  { const char* e = sr_getenv("SRHIP_MLP_STAGGER"); p.stagger = e ? atoi(e) : 0; }      // 10-ns units
  { const char* e = sr_getenv("SRHIP_MLP_STAGGER_MODE"); p.stagger_mode = e ? atoi(e) : 0; }
  p.cu_count = g_mlp_cu_count;
  if (p.cu_count && p.stagger_mode == 1) (void)hipMemsetAsync(p.cu_count, 0, 4096 * sizeof(int), st);
#endif
  SR_REQUIRE(p.C % 4 == 0 && p.C >= 4 && p.C <= 192, "mlp_f16x2: C = %d (multiple of 4, <= 192)", p.C);
  SR_REQUIRE(p.hid % 4 == 0 && p.hid >= 4 && p.hid <= 384, "mlp_f16x2: hidden = %d (multiple of 4, <= 384)", p.hid);
  SR_REQUIRE(p.M > 0, "mlp_f16x2: M = %d", p.M);
  SR_REQUIRE(p.ldx % 4 == 0 && p.ldo % 4 == 0 && p.ldh % 4 == 0 && p.ldr % 4 == 0 && (!bwd || p.ldr2 % 4 == 0),
             "mlp_f16x2: row pitches must be multiples of 4 floats");
  static const int krot = [] { const char* e = sr_getenv("SRHIP_MLP_ROT"); return e ? atoi(e) : 1; }();
  p.k_rot = krot;
  dim3 grid(sr_cdiv(p.M, BM));
  // two hidden halves, f32-grade: the producer / consumer form (one block per CU, two tiles each)
  static int gr_on = -1;
  if (gr_on < 0) {
    const char* e = sr_getenv("SRHIP_MLP_GR");
    gr_on = e ? atoi(e) : 0;        // measured (profiles/r05_mlp_structure_experiments.txt): 76.7 us against 57.4 -- off
    if (gr_on && hipFuncSetAttribute((const void*)k_mlp_gr_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, GR_LDS) != hipSuccess)
      return sr_fail(-5, "mlp_f16x2: cannot reserve %d bytes of LDS", GR_LDS);
  }
  if (!bwd && gr_on && sr_matmul_mode() != 1 && p.hid > 192) {
    hipLaunchKernelGGL(k_mlp_gr_fwd, dim3(sr_cdiv(sr_cdiv(p.M, BM), 2)), dim3(GR_NT), GR_LDS, st, p);
    SR_LAUNCH_CHECK("k_mlp_gr_fwd");
    return 0;
  }
  if (bwd) hipLaunchKernelGGL(k_mlp_f16<true>, grid, dim3(256), MLP_LDS, st, p);
  else if (sr_matmul_mode() == 1) hipLaunchKernelGGL((k_mlp_f16<false, true>), grid, dim3(256), MLP_LDS, st, p);     // inference under --amp
  else hipLaunchKernelGGL(k_mlp_f16<false>, grid, dim3(256), MLP_LDS, st, p);
  SR_LAUNCH_CHECK("k_mlp_f16");
  return 0;
}
