// The MLP half of a Swin block as ONE kernel (forward) and ONE kernel (data gradient):
//
//   forward : x2 = x1 + s * ( gelu( LN(x1) . W1^T + b1 ) . W2^T + b2 )        (+ row statistics of x2)
//   backward: g1 = g + LNbwd( ( (s * g . W2) * gelu'(h) ) . W1 )             (+ dh and gelu(h) for the weight gradients)
//
// (Mlp.forward + the residual of SwinTransformerBlock.forward, dlib/models/network_swinir.py:28-45,
// 335-337, and the autograd graph behind them.)  The separate Linear launches of gemm_ntp.hip write
// the hidden activation to HBM and read it back (forward: h; backward: dh) and each pays a launch,
// a first-touch prologue and an epilogue in which MFMA idles.  Here a block keeps its 64 tokens
// from the first product to the second:
//
//  phase 1  H^T[hidden, token] = W1 . X^T with the WEIGHT rows on the MFMA row side: the
//           accumulator of lane (token r, half h) holds hidden units 8*(q>>2) + 4h + (q&3) of each
//           32-row tile -- per 16 hidden units, 8 values of ONE token in 8 consecutive registers.
//           That is exactly one 16-byte unit [token][8 k] of the A-operand stage of the next product
//           (k order inside a 16-group permuted; the second weight is stored with the same
//           permutation: job_planes perm 2 in prep.hip), so
//  phase 2  the activation (bias, exact-erf GELU or the gelu' gate applied in registers, h / dh /
//           gelu(h) stored straight from them) goes registers -> bf16x3 planes -> LDS stage ->
//           A operand of out[token, channel] += G . W2^T in the usual orientation, and ends in
//           the epilogues of nt_epi.h (residual + DropPath scale + row statistics; LayerNorm
//           backward).
//
// Work split: 4 waves = 2 token groups x 2 hidden halves; a wave walks its half of the hidden
// layer in NCR rounds of 96 units (3 MFMA tiles).  Both phases stream their weight through the
// same 16-k stage pipeline as k_ntp (two LDS stage buffers, one barrier per stage, loads two
// stages ahead -- across phase boundaries too); phase 2's "A stage" is written from registers by
// the two waves that own the stage's hidden half.  Two blocks per CU.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "nt_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SK = 16, BM = 64, WN = 3, BN = 64 * WN;
constexpr int B_N = 3 * BN * 2;                  // 16-byte W units per stage
constexpr int B_IT = (B_N + 255) / 256;
constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
constexpr int A_STAGE = 3 * A_PLANE, B_STAGE = 3 * B_PLANE;
constexpr int CW = 32 * WN;                      // hidden units per wave and round
constexpr int P2S = 2 * CW / SK;                 // phase-2 stages per round (both halves)

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int unit_slot(int row, int u) { return 2 * row + (u ^ ((row >> 3) & 1)); }

constexpr int mlp_lds() {
  const int stages = 2 * (A_STAGE + B_STAGE);
  const int wide = 4 * 32 * (32 * WN + 8) * 4 + 2 * 2 * 64 * 4;
  return stages > wide ? stages : wide;
}

template <int NCR, bool BWD>
__global__ void __launch_bounds__(256, 2) k_mlp(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const As = smem;                 // [2][A_STAGE]
  unsigned char* const Bs = smem + 2 * A_STAGE;   // [2][B_STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int M = p.o.M, N = p.o.N;
  const int nst1 = p.Kp1 / SK;                    // phase-1 stages (even: Kp1 is a multiple of 32)
  constexpr int RS = 12 + P2S;                    // stages per round when nst1 == 12 (only used for prefetch indexing)
  const int NST = NCR * (nst1 + P2S);

  // ---- staging invariants (k_ntp): thread = one float4 of the 64 x 16 A stage + up to B_IT W units
  const int arow = tid >> 2, ac4 = tid & 3;
  const int agm = min(m0 + arow, M - 1);
  const unsigned offA = (unsigned)(agm * (int)p.ldx + ac4 * 4) * 4u;
  const float2 rst = ldg_f2(p.ln_stats ? p.ln_stats + 2 * agm : k_sr_neutral);
  const int a_dst = unit_slot(arow, ac4 >> 1) * 16 + (ac4 & 1) * 8;
  unsigned offB1[B_IT], offB2[B_IT];
  int b_dst[B_IT];
  const int N1 = NCR * BN;                        // rows of the phase-1 planes
  const long pb1 = (long)N1 * p.Kp1 * 2, pb2 = (long)N * (NCR * BN) * 2;
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = min(tid + it * 256, B_N - 1);
    const int pl = idx / (BN * 2), rem = idx - pl * (BN * 2);
    const int row = rem >> 1, u = rem & 1;
    offB1[it] = (unsigned)(pl * pb1 + (long)row * 32 + u * 16);
    offB2[it] = (unsigned)(pl * pb2 + (long)min(row, N - 1) * 32 + u * 16);
    b_dst[it] = pl * B_PLANE + unit_slot(row, u) * 16;
  }
  (void)RS;

  // global stage g -> (round, local); local < nst1: phase 1, else phase 2
  auto issue = [&](int g, f32x4& ra, u32x4 (&rb)[B_IT]) {
    if (g >= NST) return;                          // block-uniform
    const int per = nst1 + P2S;
    const int cr = g / per, l = g - cr * per;
    if (l < nst1) {
      const int k = l * SK + ac4 * 4;
      const bool oob = k >= p.K1;
      ra = *(const f32x4*)((const char*)p.X + (oob ? offA - ac4 * 16u : offA + (unsigned)l * (SK * 4)));
      const char* base = (const char*)p.W1b + ((long)l * N1 + cr * BN) * 32;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) rb[it] = *(const u32x4*)(base + offB1[it]);
    } else {
      const char* base = (const char*)p.W2b + (long)(cr * P2S + (l - nst1)) * N * 32;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) rb[it] = *(const u32x4*)(base + offB2[it]);
    }
  };
  auto store_w = [&](int buf, const u32x4 (&rb)[B_IT]) {
    unsigned char* sb = Bs + buf * B_STAGE;
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (B_N % 256 == 0 || tid + it * 256 < B_N) *(u32x4*)(sb + b_dst[it]) = rb[it];
  };
  auto store1 = [&](int c, f32x4 v, const u32x4 (&rb)[B_IT]) {     // phase-1 stage c (buffer c & 1)
    unsigned char* sa = As + (c & 1) * A_STAGE;
    if (!BWD) {                                    // LayerNorm prologue (neutral statistics without ln_stats); scalar on purpose
      v.x = (v.x - rst.x) * rst.y; v.y = (v.y - rst.x) * rst.y; v.z = (v.z - rst.x) * rst.y; v.w = (v.w - rst.x) * rst.y;
    }
    if (c * SK + ac4 * 4 >= p.K1) v = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned h0, m0_, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0_, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *(u32x2*)(sa + a_dst) = u32x2{h0, h1};
    *(u32x2*)(sa + A_PLANE + a_dst) = u32x2{m0_, m1};
    *(u32x2*)(sa + 2 * A_PLANE + a_dst) = u32x2{l0, l1};
    store_w(c & 1, rb);
  };

  f32x16 acc1[WN];        // phase 1: [hidden tile][token]   (rows = weight rows)
  f32x16 acc2[1][WN];     // phase 2: [token][channel tile]
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[0][j][q] = 0.f;
  const int a_off = unit_slot(wm * 32 + r, h) * 16;
  int b_off[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) b_off[j] = unit_slot((wn * WN + j) * 32 + r, h) * 16;

  // phase-2 stage t of a round: the waves of hidden half (t & 1) write 8 accumulator registers
  // (16 hidden units x their token) as one 16-byte unit per plane
  auto store2 = [&](int t, const u32x4 (&rb)[B_IT]) {
    const int buf = t & 1;
    if (wn == (t & 1)) {
      const int s2 = t >> 1;                       // 16-group inside the wave's 96 units
      const f32x16& tl = acc1[s2 >> 1];
      const int q0 = (s2 & 1) * 8;
      u32x4 hv, mv, lv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        split3_pair(tl[q0 + 2 * i], tl[q0 + 2 * i + 1], a, b, c);
        hv[i] = a; mv[i] = b; lv[i] = c;
      }
      unsigned char* sa = As + buf * A_STAGE + a_off;
      *(u32x4*)(sa) = hv;
      *(u32x4*)(sa + A_PLANE) = mv;
      *(u32x4*)(sa + 2 * A_PLANE) = lv;
    }
    store_w(buf, rb);
  };

#define SR_FRAGS(buf)                                                                         \
  const unsigned char* sa_ = As + (buf) * A_STAGE;                                            \
  const unsigned char* sb_ = Bs + (buf) * B_STAGE;                                            \
  u32x4 fa[3], fb[WN][3];                                                                     \
  _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) fa[pl] = *(const u32x4*)(sa_ + pl * A_PLANE + a_off); \
  _Pragma("unroll") for (int j = 0; j < WN; ++j)                                              \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) fb[j][pl] = *(const u32x4*)(sb_ + pl * B_PLANE + b_off[j]);
  // small terms first; term-outer so that consecutive MFMAs hit different tiles
  auto mma1 = [&](int buf) {       // weight rows on the MFMA row side
    SR_FRAGS(buf)
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < WN; ++j) acc1[j] = mfma_bf(fb[j][PB], fa[PA], acc1[j]);
    SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
  };
  auto mma2 = [&](int buf) {
    SR_FRAGS(buf)
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < WN; ++j) acc2[0][j] = mfma_bf(fa[PA], fb[j][PB], acc2[0][j]);
    SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
  };
#undef SR_FRAGS

  // between the phases (the vector work of a whole round at once: spread over the phase-2 stages it
  // sits in front of every stage's barrier and the block waits for the two owner waves each time --
  // measured 115 us against 95 us): the lane's token, 4 consecutive hidden units per register quad
  const int tok = m0 + wm * 32 + r, tokc = min(tok, M - 1);
  const bool tok_ok = tok < M;
  float s1 = 1.f;
  if (BWD && p.rowscale1) s1 = ldg_f(p.rowscale1 + tokc / p.rows_per_scale1);
  auto transform = [&](int cr) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      f32x4 ld[4];
      int hidx[4];
      bool ok[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int off = cr * CW + j * 32 + 8 * qq + 4 * h;      // inside the hidden half
        ok[qq] = off < p.hs;                                       // hs % 4 == 0: a quad is valid as a whole
        hidx[qq] = wn * p.hs + (ok[qq] ? off : 0);
        if (BWD) ld[qq] = ldg_f4(p.H + (long)tokc * p.ldh + hidx[qq]);
        else ld[qq] = ldg_f4(p.b1 + hidx[qq]);
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        f32x4 v = {acc1[j][4 * qq], acc1[j][4 * qq + 1], acc1[j][4 * qq + 2], acc1[j][4 * qq + 3]};
        if (!BWD) {
          v = v + ld[qq];
          if (p.H && ok[qq] && tok_ok) *(f32x4*)(p.H + (long)tok * p.ldh + hidx[qq]) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok[qq] ? gelu_f(v[e]) : 0.f;
        } else {
          f32x4 gl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {            // Phi / phi as in nt_epilogue's epilogue 3 (backward-grade erf)
            const float x = ld[qq][e];
            const float z = fabsf(x) * 0.70710678118654752440f;
            const float e1 = __expf(-0.5f * x * x);
            const float t = __frcp_rn(1.0f + 0.3275911f * z);
            const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f +
                               t * (-1.453152027f + t * 1.061405429f))));
            const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e1, x));
            v[e] = ok[qq] ? v[e] * s1 * (cdf + x * 0.39894228040143267794f * e1) : 0.f;
            gl[e] = x * cdf;
          }
          if (ok[qq] && tok_ok) {
            *(f32x4*)(p.dH + (long)tok * p.ldh + hidx[qq]) = v;
            *(f32x4*)(p.GH + (long)tok * p.ldh + hidx[qq]) = gl;
          }
        }
        acc1[j][4 * qq] = v.x; acc1[j][4 * qq + 1] = v.y; acc1[j][4 * qq + 2] = v.z; acc1[j][4 * qq + 3] = v.w;
      }
    }
  };

  // ---- pipeline: register set P holds the stage of parity P (every phase has an even stage count)
  f32x4 ra0, ra1;
  u32x4 rb0[B_IT], rb1[B_IT];
  issue(0, ra0, rb0);
  issue(1, ra1, rb1);
  store1(0, ra0, rb0);
  issue(2, ra0, rb0);
  __syncthreads();
  int g0 = 0;                                      // global index of the round's first stage
#pragma unroll 1
  for (int cr = 0; cr < NCR; ++cr) {
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[j][q] = 0.f;
    // ---- phase 1: nst1 stages
#pragma unroll 1
    for (int c = 0; c < nst1; c += 2) {
      store1(c + 1, ra1, rb1);
      issue(g0 + c + 3, ra1, rb1);
      mma1(0);
      __syncthreads();
      if (c + 2 < nst1) {
        store1(c + 2, ra0, rb0);
        issue(g0 + c + 4, ra0, rb0);
        mma1(1);
        __syncthreads();
      } else {
        mma1(1);
      }
    }
    // the first phase-2 stage needs the finished accumulators
    transform(cr);
    store2(0, rb0);
    issue(g0 + nst1 + 2, ra0, rb0);
    __syncthreads();
    // ---- phase 2: P2S stages (compile-time register indices)
#pragma unroll
    for (int t = 0; t < P2S; t += 2) {
      store2(t + 1, rb1);
      issue(g0 + nst1 + t + 3, ra1, rb1);
      mma2(0);
      __syncthreads();
      if (t + 2 < P2S) {
        store2(t + 2, rb0);
        issue(g0 + nst1 + t + 4, ra0, rb0);
      } else if (cr + 1 < NCR) {
        store1(0, ra0, rb0);                       // stage 0 of the next round
        issue(g0 + nst1 + t + 4, ra0, rb0);
      }
      mma2(1);
      if (t + 2 < P2S || cr + 1 < NCR) __syncthreads();
    }
    g0 += nst1 + P2S;
  }

  if (BWD) nt_epilogue_lnbwd<WN>(p.o, acc2, lane, wm, wn, m0, N, (float*)smem);
  else nt_epilogue_wide<WN>(p.o, acc2, lane, wave, wm, wn, 0, N, m0, (float*)smem);
}

}  // namespace

int sr_mlp_fused(MlpArgs& p, hipStream_t st) {
  const int hs = p.hs;
  SR_REQUIRE(p.o.M > 0 && p.o.N > 0 && p.o.N <= BN && p.o.N % 4 == 0, "mlp_fused: channels %d (<= 192, multiple of 4)", p.o.N);
  SR_REQUIRE(p.K1 == p.o.N, "mlp_fused: the MLP maps channels -> hidden -> channels");
  SR_REQUIRE(hs % 4 == 0 && hs > CW && hs <= 2 * CW, "mlp_fused: hidden %d (needs 192 < hidden <= 384, multiple of 8)", 2 * hs);
  SR_REQUIRE(p.ldx % 4 == 0 && p.ldh % 4 == 0 && p.o.ldc % 4 == 0 && p.o.ldr % 4 == 0, "mlp_fused: row strides must be multiples of 4");
  const auto al = [](const void* q) { return ((size_t)q & 15) == 0; };
  SR_REQUIRE(al(p.X) && al(p.H) && al(p.dH) && al(p.GH) && al(p.o.C) && al(p.o.R) && al(p.b1), "mlp_fused: operands must be 16-byte aligned");
  SR_REQUIRE((long)p.o.M * p.ldx < (1L << 29), "mlp_fused: activation larger than 2 GiB (32-bit staging offsets)");
  p.Kp1 = sr_kp(p.K1);
  SR_REQUIRE(p.Kp1 / 16 == 12 || p.Kp1 / 16 % 2 == 0, "mlp_fused: stage count");
  dim3 grid(sr_cdiv(p.o.M, BM));
  if (p.bwd) hipLaunchKernelGGL((k_mlp<2, true>), grid, dim3(256), mlp_lds(), st, p);
  else hipLaunchKernelGGL((k_mlp<2, false>), grid, dim3(256), mlp_lds(), st, p);
  SR_LAUNCH_CHECK("k_mlp");
  return 0;
}
