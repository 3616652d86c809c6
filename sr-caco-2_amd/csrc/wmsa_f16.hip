// The W-MSA half of a Swin block, forward, as ONE kernel on the two-plane fp16 split MFMA (three products, block
// exponents) -- reference SwinTransformerBlock.forward up to the first residual, dlib/models/network_swinir.py:288-334
// with WindowAttention.forward (:153-176) inside:
//
//   qkv = LN(x) Wqf^T + bqf ;  a = softmax(scale q k^T + bias + mask) v  per (window, head) ;
//   out = x + s (a Wp^T + bp) ;  stats_out = {mean, rstd} of the out rows
//
// Why one kernel: the three launches it replaces (qkv GEMM, attention core, proj GEMM) each pay launch + operand fetch +
// epilogue in series, and the attention core and the proj GEMM re-read from HBM what the launch before them just wrote.
// A block here owns ONE 8x8 window (64 tokens, gathered under the cyclic shift): every token of the window's q, k, v is
// produced by this block, so the attention of the window's heads can start as soon as the block's own qkv stores are
// visible (one __syncthreads) and reads them back from L2, and so can the proj GEMM behind the attention.  qkv and a
// still go out to HBM once (the backward reads them) but are not read back from it.
//
//   phase 1  x rows of the window -> LayerNorm -> six stage images (the A operand of k_nth2 / k_mlp_f16)
//   phase 2  qkv GEMM, transposed as GEMM 1 of k_mlp_f16: a lane ends with four consecutive output channels of one
//            token = one 16-byte store; three column tiles (q, k, v) of C channels, 18 barrier-free stages
//   phase 3  attention: wave w runs heads w, w + 4 (w2_fwd_body of wattn2_dev.h, no LDS)
//   phase 4  proj GEMM (k_nth2's loop) on the a rows, row-major epilogue with bias, DropPath scale, residual, statistics
//
// LDS: one 49-KB region (stage images, then the output tile); two blocks per CU; 512 windows of a B = 8, 64x64 batch
// are exactly one round.
#include "common.h"
#include "kernels.h"
#include "wattn2_dev.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SK = 32;                 // k per stage
constexpr int BM = 64;                 // tokens per block = one window
constexpr int APL = BM * 64;           // bytes of one plane of a stage image (64 rows x 32 fp16)
constexpr int AST = 2 * APL;           // a stage image: two planes
constexpr int TP = 196;                // pitch of the output tile (floats)
constexpr int R0 = BM * TP * 4;        // the shared region (>= 6 stage images)
constexpr int WMSA_LDS = R0 + (64 + 64 + 2 * 3 * 192) * 4;
static_assert(R0 >= 6 * AST, "LDS region");

__device__ __forceinline__ int a_slot(int row, int u) { return row * 4 + (u ^ ((row >> 2) & 3)); }

// phase timestamps of every wave (tools/mb_wmsa_f16.py --phases): experiment builds only
#ifdef SRHIP_EXPERIMENTS
long long* g_wmsa_dbg = nullptr;
#define SR_TS(K) \
  if (p.dbg && lane == 0) p.dbg[(((long)blockIdx.x * WPB + grp) * NW + wave) * 16 + (K)] = (long long)wall_clock64();
#else
#define SR_TS(K)
#endif

// NW waves per block, each CW = 192 / NW output columns wide in both GEMMs (NJ 16-column fragments): 4 waves, or 6 when
// the window has 5 or 6 heads (one head per wave in phase 3; three waves per SIMD)
// WPB windows per block, each run by its own group of NW waves on its own LDS region (the groups share nothing but the
// barriers).  Why not two 6-wave blocks per CU: a block's six waves sit 2, 2, 1, 1 on the four SIMDs and the second
// block's are not placed 1, 1, 2, 2 -- at 168 registers only one such block is resident per CU (measured: the second
// half of the grid started when the first half ended).  One 12-wave block is three waves on every SIMD.
template <int D, int NW, int WPB>
__global__ void __launch_bounds__(64 * NW * WPB, NW / 2) k_wmsa_f16(WmsaF16Args p) {
  constexpr int NT = 64 * NW, CW = 192 / NW, NJ = CW / 16, NIT = 3072 / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int grp = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  unsigned char* const smem = smem_all + grp * WMSA_LDS;
  int* const tokl = (int*)(smem + R0);               // [64] token index of window position
  float* const rinva = (float*)(tokl + 64);          // [64] 2^-s of the a rows
  float* const colq = rinva + 64;                    // [2][3C] column scales (2^-s of x and of the W rows) and biases of qkv
  const int tid = (int)threadIdx.x - grp * NT, lane = tid & 63;      // thread within the window's group
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int C = p.C, N3 = 3 * C;
  const int nst = p.Kp / SK;
  const int nWx = p.W / 8, nWy = p.H / 8;
  // (a block past the last window repeats it: same values to the same addresses)
  const int win = min(sr_xcd_block((int)blockIdx.x, gridDim.x) * WPB + grp, p.B * nWx * nWy - 1);
  W2Geom geo = w2_decode((long)win, 1, nWx, nWy, p.shift);

  // ---------------- weight fragment addressing
  const long planeq = (long)N3 * p.Kp * 2, planep = (long)C * p.Kp * 2;
  const float* const winvq = (const float*)((const char*)p.Wqkv + 2 * planeq);
  const float* const winvp = (const float*)((const char*)p.Wproj + 2 * planep);
  unsigned boffq[NJ];
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    const int col = min(wave * CW + jt * 16 + c, C - 1);
    boffq[jt] = (unsigned)(((g >> 1) * N3 + col) * 32 + (g & 1) * 16);
  }
  // stage u of the qkv GEMM = (column tile u / 6, k stage u % 6)
  auto load_bq = [&](int u, u32x4 (&fb)[NJ][2]) {
    const int ct = u / 6, s = min(u - 6 * ct, nst - 1);
    const char* base = (const char*)p.Wqkv + (long)(2 * s) * N3 * 32 + (long)ct * C * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * planeq + boffq[jt]);
  };
  SR_TS(0)
  u32x4 fb0[NJ][2], fb1[NJ][2], fb2[NJ][2];
  load_bq(0, fb0);
  load_bq(1, fb1);
  load_bq(2, fb2);

  // ---------------- a 64 x C row block (rows gathered by token) -> six stage images; LN: LayerNorm prologue with the
  // a-priori block exponent, else the row maximum's.  Returns nothing; the 2^-s of the rows go to rinva.
  // (the first four waves: four lanes per row: thread = (row arow, k quarter akq); atok = the row's token)
  auto stage_rows = [&](const float* __restrict__ src, const bool ln, const int arow, const int akq, const int atok) {
    const char* const abase = (const char*)src + (long)atok * C * 4;
    const float2 rst = ldg_f2(ln ? p.ln_stats + 2 * (long)atok : k_sr_neutral);
    f32x4 ra[6][2];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {                 // past the end: k = 0 of the row, zeroed below
      const int k = s6 * SK + akq * 8;
      ra[s6][0] = *(const f32x4*)(abase + (k < C ? k * 4 : 0));
      ra[s6][1] = *(const f32x4*)(abase + (k + 4 < C ? (k + 4) * 4 : 0));
    }
    float mx = 0.f;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      const int k = s6 * SK + akq * 8;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4 x = ra[s6][e];
        x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y;
        if (k + 4 * e >= C) x = f32x4{0.f, 0.f, 0.f, 0.f};
        ra[s6][e] = x;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
      }
    }
    float asc;
    if (ln) {                                        // |xhat| <= sqrt(K): a priori
      asc = exp2f(floorf(log2f(16384.f * rsqrtf((float)C))));
    } else {
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      asc = pow2_scale(mx);
    }
    if (akq == 0) rinva[arow] = pow2_inv(asc);
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
      *(u32x4*)(sa + APL) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  {
    const float rix = 1.0f / exp2f(floorf(log2f(16384.f * rsqrtf((float)C))));
    for (int n = tid; n < N3; n += NT) {
      colq[n] = rix * winvq[n];
      colq[N3 + n] = p.bqkv[n];
    }
  }
  {
    const int arow = (tid >> 2) & 63, akq = tid & 3;
    const int atok = w2_token(geo, arow, p.H, p.W, p.shift);
    if (akq == 0 && tid < 256) tokl[arow] = atok;
    if (NW == 4 || wave < 4) stage_rows(p.X, true, arow, akq, atok);
  }
  SR_TS(1)
  __syncthreads();
  SR_TS(2)

  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = a_slot(16 * i + c, g) * 16;

  // ---------------- phase 2: qkv, transposed: acc[i][j] = channels ct C + CW wave + 16 j + 4 g + e of token 16 i + c
  {
    f32x4 acc[4][NJ];
    auto zero = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma1 = [&](int s6, const u32x4 (&fb)[NJ][2]) {
      const unsigned char* sa = smem + s6 * AST;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 fa[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16h(fb[j][PB], fa[PA], acc[i][j]);
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)          // small terms first
#undef SR_TERM
      }
    };
    int tk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tk[i] = tokl[16 * i + c];
    auto store_tile = [&](int ct) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch0 = wave * CW + 16 * j + 4 * g;
        const bool ok = ch0 < C;                     // C % 4 == 0: a lane's four channels are valid or not together
        const int cc = ct * C + min(ch0, C - 4);
        const f32x4 wi = *(const f32x4*)(colq + cc);
        const f32x4 bv = *(const f32x4*)(colq + N3 + cc);
        if (ok) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * wi[e] + bv[e];
            *(f32x4*)(p.qkv + (long)tk[i] * N3 + cc) = v;
          }
        }
      }
    };
#define SR_G1(U, FB)                                \
  {                                                 \
    mma1((U) % 6, FB);                              \
    if ((U) + 3 < 18) load_bq((U) + 3, FB);         \
    __builtin_amdgcn_sched_barrier(0);              \
  }
    zero();
    SR_G1(0, fb0) SR_G1(1, fb1) SR_G1(2, fb2) SR_G1(3, fb0) SR_G1(4, fb1) SR_G1(5, fb2)
    SR_TS(10)
    store_tile(0);
    SR_TS(11)
    zero();
    SR_G1(6, fb0) SR_G1(7, fb1) SR_G1(8, fb2) SR_G1(9, fb0) SR_G1(10, fb1) SR_G1(11, fb2)
    SR_TS(12)
    store_tile(1);
    SR_TS(13)
    zero();
    SR_G1(12, fb0) SR_G1(13, fb1) SR_G1(14, fb2) SR_G1(15, fb0) SR_G1(16, fb1) SR_G1(17, fb2)
    SR_TS(14)
    store_tile(2);
#undef SR_G1
  }
  SR_TS(3)
  __syncthreads();                                   // the window's qkv rows are visible to the whole block
  SR_TS(4)

  // ---------------- phase 3: attention of the window's heads
  if (NW == 6) {                                     // heads <= 6: one head per wave
    if (wave < p.heads) {
      geo.head = wave;
      w2_fwd_body<D>(p.qkv, p.att, p.biasF, geo, p.H, p.W, C, p.shift, p.scale, lane);
    }
  } else {
    for (int hd = wave; hd < p.heads; hd += NW) {
      geo.head = hd;
      w2_fwd_body<D>(p.qkv, p.att, p.biasF, geo, p.H, p.W, C, p.shift, p.scale, lane);
    }
  }
  SR_TS(5)
  // Phase 3 needs every register it can get (three waves per SIMD when NW = 6): whatever phase 4 needs of the thread's
  // position is derived again from the thread index, so that nothing but the index lives through phase 3.
  int t4 = tid;
  asm volatile("" : "+v"(t4));
  const int lane4 = t4 & 63, c4 = lane4 & 15, g4 = lane4 >> 4;
  unsigned boffp[NJ];
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    const int col = min(wave * CW + jt * 16 + c4, C - 1);
    boffp[jt] = (unsigned)(((g4 >> 1) * C + col) * 32 + (g4 & 1) * 16);
  }
  auto load_bp = [&](int s, u32x4 (&fb)[NJ][2]) {
    const char* base = (const char*)p.Wproj + (long)(2 * min(s, nst - 1)) * C * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * planep + boffp[jt]);
  };
  load_bp(0, fb0);
  load_bp(1, fb1);
  load_bp(2, fb2);
  __syncthreads();                                   // ... and so are its a rows
  SR_TS(6)

  // ---------------- phase 4: proj (k_nth2's loop): acc2[i][j] = rows 16 i + 4 g + e, columns CW wave + 16 j + c
  const int arow4 = (t4 >> 2) & 63, akq4 = t4 & 3;
  const int atok4 = tokl[arow4];
  if (NW == 4 || wave < 4) stage_rows(p.att, false, arow4, akq4, atok4);
  int a_off4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off4[i] = a_slot(16 * i + c4, g4) * 16;
  __syncthreads();
  SR_TS(7)
  f32x4 acc2[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma2 = [&](int s6, const u32x4 (&fb)[NJ][2]) {
    const unsigned char* sa = smem + s6 * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off4[i]);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc2[i][j] = mfma16h(fa[PA], fb[j][PB], acc2[i][j]);
      SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
#undef SR_TERM
    }
  };
  // the residual pieces of the row-major epilogue travel while the MFMAs run
  f32x4 rv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * NT + t4, row = idx / 48, col = (idx - row * 48) * 4;
    rv[it] = *(const f32x4*)(p.X + (long)tokl[row] * C + min(col, C - 4));
  }
  __builtin_amdgcn_sched_barrier(0);
  mma2(0, fb0); load_bp(3, fb0); __builtin_amdgcn_sched_barrier(0);
  mma2(1, fb1); load_bp(4, fb1); __builtin_amdgcn_sched_barrier(0);
  mma2(2, fb2); load_bp(5, fb2); __builtin_amdgcn_sched_barrier(0);
  mma2(3, fb0); __builtin_amdgcn_sched_barrier(0);
  mma2(4, fb1); __builtin_amdgcn_sched_barrier(0);
  mma2(5, fb2); __builtin_amdgcn_sched_barrier(0);
  SR_TS(8)

  // ---------------- the output tile, row-major in LDS (block exponents undone: exact powers of two)
  __syncthreads();
  float* const T = (float*)smem;
  {
    float wv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wv[j] = winvp[min(wave * CW + 16 * j + c4, C - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ri = rinva[16 * i + 4 * g4 + e];
#pragma unroll
        for (int j = 0; j < NJ; ++j) T[(16 * i + 4 * g4 + e) * TP + wave * CW + 16 * j + c4] = acc2[i][j][e] * (ri * wv[j]);
      }
  }
  __syncthreads();
  // out = x + s (acc + bp): 16-byte pieces in row-major order
  const float dps = p.rowscale ? p.rowscale[geo.b] : 1.f;
  {
    int prow[NIT], pcol[NIT], ptok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * NT + t4;
      prow[it] = idx / 48;
      pcol[it] = (idx - prow[it] * 48) * 4;
      ptok[it] = tokl[prow[it]];
      if (pcol[it] >= C) prow[it] = -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (prow[it] >= 0) {
        float* tp = T + prow[it] * TP + pcol[it];
        f32x4 v = *(const f32x4*)tp;
        const f32x4 bv = p.bproj ? *(const f32x4*)(p.bproj + pcol[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + bv[e]) * dps + rv[it][e];
        *(f32x4*)(p.out + (long)ptok[it] * C + pcol[it]) = v;
        if (p.stats_out) *(f32x4*)tp = v;
      }
    }
  }
  if (p.stats_out) {
    // {mean, rstd} of the out rows for the next LayerNorm (two-pass, eps 1e-5, biased variance): four lanes per row
    __syncthreads();
    if (NW > 4 && wave >= 4) return;
    const int row = arow4, q = akq4;
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
    if (q == 0) *(float2*)(p.stats_out + 2 * (long)atok4) = float2{mean, rsqrtf(s2 * (1.0f / (float)C) + 1e-5f)};
  }
  SR_TS(9)
}


// ------------------------------------------------------------------------------------------------------------------
// 5 or 6 heads: ONE HEAD PER WAVE from the qkv GEMM to the attention output, nothing in between leaves the registers.
// k_wmsa_f16 above gives a wave 32 qkv channels of every token, stores them, and after a barrier the attention of a head
// gathers q, k rows (8-byte loads) and v columns (4-byte loads) back from L2 / the Infinity Cache: 17-22 us of its 72
// and the second half of its HBM traffic (284 MB measured for 142 MB of operands).  Here wave h owns the head's D columns
// of q, of k and of v (three 64 x 32 sub-GEMMs over the same stage images, the weight rows addressed per lane):
//   q, k  transposed product -- lane (c, g) ends with entries 16 j + 4 g + e of token 16 i + c: after the epilogue these
//         eight values ARE the row-form fragment of S^T = K . Q^T (the contraction runs over the head dim in any order,
//         the same for q and k);
//   v     plain product -- lane (c, g) ends with entry 16 j + c of tokens 16 i + 4 g + e = the V^T operand of
//         O^T = V^T . P^T in the key order of the P registers;
// qkv is still written (the backward reads it) but never read.  The attention output goes to global memory (backward) and,
// split under the row's exponent (row maximum over the six waves: one LDS atomic per row and wave), straight into the
// stage images of the proj GEMM: no re-staging pass, one barrier less.  The relative-position bias comes from the heads'
// 225-entry tables in LDS (index = lane constant + 30 (I - J) - e).
constexpr int TABP = 228;              // pitch of a head's bias table
constexpr int WMSA_LDSH = R0 + (64 + 64 + 2 * 3 * 192 + 64 + 6 * TABP + 6 * 96) * 4;

// AMP (srhip_set_matmul_mode(1): inference under --amp): one product of the leading fp16 planes everywhere -- no lo plane is
// staged, loaded or multiplied
template <int D, int WPB, bool AMP = false>
__global__ void __launch_bounds__(64 * 6 * WPB, 3) k_wmsa_f16h(WmsaF16Args p) {
  constexpr int NPL = AMP ? 1 : 2;
  constexpr int NW = 6, NT = 64 * NW, CW = 32, NJ = 2, NIT = 3072 / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int grp = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  unsigned char* const smem = smem_all + grp * WMSA_LDSH;
  int* const tokl = (int*)(smem + R0);               // [64] token index of window position
  float* const rinva = (float*)(tokl + 64);          // [64] 2^-s of the a rows
  float* const colq = rinva + 64;                    // [2][3C] column scales (2^-s of x and of the W rows) and biases of qkv
  unsigned* const amax = (unsigned*)(colq + 2 * 3 * 192);   // [64] bits of the a rows' maxima
  float* const tab = (float*)(amax + 64);            // [heads][TABP] relative-position bias tables
  float* const wsc = tab + 6 * TABP;                 // [wave][96]: 2^-s of the head's 64 key rows, of its 32 V columns (x 2^-14)
  const int tid = (int)threadIdx.x - grp * NT, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int C = p.C, N3 = 3 * C;
  const int nst = p.Kp / SK;
  // as k_ntw / k_mlp_f16: every block starts its K walks at another stage (blocks of one XCD -- same blockIdx.x mod 8 -- get
  // different rotations), so that the launch does not ask an L2 for the same weight lines at the same moment; the stage
  // images are complete before a product starts; f32 sums in another order per block (deterministic)
  const int rot = p.k_rot ? (int)(((unsigned)blockIdx.x >> 3) % (unsigned)nst) : 0;
  auto rst = [&](int s) { const int x = s + rot; return s >= nst ? s : (x >= nst ? x - nst : x); };   // stages past the weight's K (zero images) stay
  const int nWx = p.W / 8, nWy = p.H / 8;
  const int win = min(sr_xcd_block((int)blockIdx.x, gridDim.x) * WPB + grp, p.B * nWx * nWy - 1);
  const W2Geom geo = w2_decode((long)win, 1, nWx, nWy, p.shift);
  const long planeq = (long)N3 * p.Kp * 2, planep = (long)C * p.Kp * 2;
  const float* const winvq = (const float*)((const char*)p.Wqkv + 2 * planeq);
  const float* const winvp = (const float*)((const char*)p.Wproj + 2 * planep);
#ifdef SRHIP_EXPERIMENTS
  if (p.stagger > 0 && grp == 1) {                   // experiment: the block's second window starts late
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < p.stagger) __builtin_amdgcn_s_sleep(16);
  }
#endif
  SR_TS(0)
  // ---------------- phase 1: x rows of the window -> LayerNorm -> stage images (the first four waves: four lanes per row)
  {
    const float rix = 1.0f / exp2f(floorf(log2f(16384.f * rsqrtf((float)C))));
    for (int n = tid; n < N3; n += NT) {
      colq[n] = rix * winvq[n];
      colq[N3 + n] = p.bqkv[n];
    }
    for (int n = tid; n < p.heads * 225; n += NT) {     // table entry (dy + 7) * 15 + dx + 7 read back from the S^T image
      const int hd = n / 225, r = n - 225 * hd;
      const int dy = r / 15 - 7, dx = r % 15 - 7;
      const int qy = max(dy, 0), ky = qy - dy, qx = max(dx, 0), kx = qx - dx;
      const int query = 8 * qy + qx, key = 8 * ky + kx;
      tab[hd * TABP + r] = ldg_f(p.biasF + (long)hd * 4096 + w2_img_index(query >> 4, key >> 4, 16 * ((key & 15) >> 2) + (query & 15)) + (key & 3));
    }
    if (tid < 64) amax[tid] = 0u;
    const int arow = (tid >> 2) & 63, akq = tid & 3;
    const int atok = w2_token(geo, arow, p.H, p.W, p.shift);
    if (akq == 0 && tid < 256) tokl[arow] = atok;
    if (wave < 4) {
      const char* const abase = (const char*)p.X + (long)atok * C * 4;
      const float2 rst = ldg_f2(p.ln_stats + 2 * (long)atok);
      f32x4 ra[6][2];
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {                 // past the end: k = 0 of the row, zeroed below
        const int k = s6 * SK + akq * 8;
        ra[s6][0] = *(const f32x4*)(abase + (k < C ? k * 4 : 0));
        ra[s6][1] = *(const f32x4*)(abase + (k + 4 < C ? (k + 4) * 4 : 0));
      }
      const float asc = exp2f(floorf(log2f(16384.f * rsqrtf((float)C))));       // |xhat| <= sqrt(K): a priori
      const int a_dst = a_slot(arow, akq) * 16;
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
        const int k = s6 * SK + akq * 8;
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f32x4 x = ra[s6][e];
          x.x = (x.x - rst.x) * rst.y; x.y = (x.y - rst.x) * rst.y; x.z = (x.z - rst.x) * rst.y; x.w = (x.w - rst.x) * rst.y;
          if (k + 4 * e >= C) x = f32x4{0.f, 0.f, 0.f, 0.f};
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

  // ---------------- phases 2 + 3: the head's q, k, v and its attention, in registers
  if (wave < p.heads) {
    const int hd = wave;
    const int a_off0 = a_slot(c, g) * 16;            // a_slot(16 i + c, g) * 16 = a_off0 + 1024 i
    unsigned wofs[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) wofs[jt] = (unsigned)(((g >> 1) * N3 + D * hd + min(16 * jt + c, D - 1)) * 32 + (g & 1) * 16);
    auto load_w = [&](int ct, int s, u32x4 (&fb)[NJ][2]) {
      const char* base = (const char*)p.Wqkv + (long)(2 * rst(s)) * N3 * 32 + (long)ct * C * 32;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * planeq + wofs[jt]);
    };
    f32x4 acc[4][NJ];
    // TR: acc[i][j] = entries 16 j + 4 g + e of token 16 i + c;  else: entry 16 j + c of tokens 16 i + 4 g + e
    auto gemm = [&](int ct, const bool TR) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 fA[NJ][2], fB[NJ][2];
      auto mma = [&](int s, const u32x4 (&fb)[NJ][2]) {
        const unsigned char* sa = smem + rst(s) * AST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 fa[2];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off0 + 1024 * i);
#define SR_TERM(PA, PB)                                                                      \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = TR ? mfma16h(fb[j][PB], fa[PA], acc[i][j]) \
                                                                 : mfma16h(fa[PA], fb[j][PB], acc[i][j]);
          if constexpr (AMP) {
            SR_TERM(0, 0)
          } else {
            SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)          // small terms first
          }
#undef SR_TERM
        }
      };
      load_w(ct, 0, fA);
      load_w(ct, min(1, nst - 1), fB);
      for (int s = 0; s < nst; s += 2) {
        mma(s, fA);
        if (s + 2 < nst) load_w(ct, s + 2, fA);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {
          mma(s + 1, fB);
          if (s + 3 < nst) load_w(ct, s + 3, fB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // epilogue of a transposed product: scale, bias, store, and the row-form fragments (entries past D are zero)
    auto rows_out = [&](int ct, u32x4 (&fh)[4], u32x4 (&fl)[4], float (&rinv)[4]) {
      float2 wi[NJ][2], bv[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int cc = ct * C + D * hd + min(16 * j + 4 * g + 2 * h2, D - 2);
          wi[j][h2] = *(const float2*)(colq + cc);
          bv[j][h2] = *(const float2*)(colq + N3 + cc);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int d0 = 16 * j + 4 * g;
          f32x4 o;
          o[0] = acc[i][j][0] * wi[j][0].x + bv[j][0].x; o[1] = acc[i][j][1] * wi[j][0].y + bv[j][0].y;
          o[2] = acc[i][j][2] * wi[j][1].x + bv[j][1].x; o[3] = acc[i][j][3] * wi[j][1].y + bv[j][1].y;
          if (d0 >= D) { o[0] = 0.f; o[1] = 0.f; }
          if (d0 + 2 >= D) { o[2] = 0.f; o[3] = 0.f; }
          if (p.qkv) w3_store<D>(p.qkv + (long)tokl[16 * i + c] * N3 + ct * C + D * hd + d0, d0, o, 1.0f);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * j + e] = o[e];
        }
        rinv[i] = w2_split_row(v, fh[i], fl[i]);
      }
    };
    u32x4 qh[4], ql[4], kh[4], kl[4];
    float rq[4], rk[4];
    gemm(0, true);
    rows_out(0, qh, ql, rq);
    SR_TS(10)
    gemm(1, true);
    rows_out(1, kh, kl, rk);
    SR_TS(11)
    gemm(2, false);
    // V: scale, bias, store (4-byte: a lane holds ONE entry of four tokens), and the V^T operand under one power-of-two
    // scale per head-dim column
    u32x4 vh[2][2], vl[2][2];
    float* const rkl = wsc + 96 * hd;                // the wave's own 96 floats: written and read by this wave only
    {
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        const bool ok = 16 * jd + c < D;
        const int cv = 2 * C + D * hd + min(16 * jd + c, D - 1);
        const float wv = colq[cv], bs = colq[N3 + cv];
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = ok ? acc[i][jd][e] * wv + bs : 0.f;
            acc[i][jd][e] = x;
            mx = fmaxf(mx, fabsf(x));
            if (ok && p.qkv) p.qkv[(long)tokl[16 * i + 4 * g + e] * N3 + cv] = x;
          }
        }
        mx = w3_max4(mx);
        const float sc = pow2_scale(mx);
        if (g == 0) rkl[64 + 16 * jd + c] = pow2_inv(sc) * (1.0f / 16384.f);      // column 16 jd + c, times the 2^-14 of P
#pragma unroll
        for (int JJ = 0; JJ < 2; ++JJ) {
          unsigned h[4], l[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {          // slots (2 t, 2 t + 1) = tokens w2_kpos(JJ, g, 2 t), + 1
            const f32x4 a = acc[2 * JJ + (t >> 1)][jd];
            const int e0 = 2 * (t & 1);
            split2_pair(a[e0] * sc, a[e0 + 1] * sc, h[t], l[t]);
          }
          vh[JJ][jd] = u32x4{h[0], h[1], h[2], h[3]};
          vl[JJ][jd] = u32x4{l[0], l[1], l[2], l[3]};
        }
      }
    }
    if (g == 0) {
#pragma unroll
      for (int J = 0; J < 4; ++J) rkl[16 * J + c] = rk[J];                        // key 16 J + c
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    SR_TS(12)
    const bool lane_masked = geo.last_col && (((c >> 2) & 1) != (g & 1));
    const float* const tb0 = tab + hd * TABP + ((c >> 3) - (g >> 1) + 7) * 15 + (c & 7) - 4 * (g & 1) + 7;
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      f32x4 S[4];
      const float rqs = rq[I] * p.scale;
      float mx = -3.0e38f;
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!AMP) {
          a = mfma16h(kh[J], ql[I], a);
          a = mfma16h(kl[J], qh[I], a);
        }
        a = mfma16h(kh[J], qh[I], a);
        const float* tb = tb0 + 30 * (I - J);
        const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
        const f32x4 rkk = *(const f32x4*)(rkl + 16 * J + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sv = a[e] * (rqs * rkk[e]) + tb[-e];
          sv += masked ? -100.f : 0.f;
          a[e] = sv;
          mx = fmaxf(mx, sv);
        }
        S[J] = a;
      }
      mx = w3_max4(mx);
      float sum = 0.f;
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __expf(S[J][e] - mx);
          S[J][e] = pe;
          sum += pe;
        }
      sum = w3_sum4(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
      f32x4 O[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int JJ = 0; JJ < 2; ++JJ) {
        unsigned h[4], l[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 pv = S[2 * JJ + (t >> 1)];
          const int e0 = 2 * (t & 1);
          split2_pair(pv[e0] * 16384.f, pv[e0 + 1] * 16384.f, h[t], l[t]);
        }
        const u32x4 ph = u32x4{h[0], h[1], h[2], h[3]}, pl = u32x4{l[0], l[1], l[2], l[3]};
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          if (!AMP) {
            O[jd] = mfma16h(vh[JJ][jd], pl, O[jd]);
            O[jd] = mfma16h(vl[JJ][jd], ph, O[jd]);
          }
          O[jd] = mfma16h(vh[JJ][jd], ph, O[jd]);
        }
      }
      float am = 0.f;
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        const int d0 = 16 * jd + 4 * g;
        const f32x4 rvo = *(const f32x4*)(rkl + 64 + d0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (d0 + e < D) ? O[jd][e] * (rvo[e] * inv) : 0.f;
        w3_store<D>(p.att + (long)tokl[16 * I + c] * C + D * hd + d0, d0, o, 1.0f);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
      }
      am = w3_max4(am);
      if (g == 0) atomicMax(amax + 16 * I + c, __builtin_bit_cast(unsigned, am));
    }
  }
  SR_TS(3)
  __syncthreads();                                   // every head is through its GEMMs: the stage images are free; row maxima complete
  SR_TS(4)
  // ---------------- the a rows, split under their exponents, into the stage images of the proj GEMM.  Each wave reads
  // back what it stored itself (its head's columns: L2 hits) -- held in registers through the attention the 32 values
  // per lane would put the kernel over the 168 registers of three waves per SIMD
  if (wave < p.heads) {
    const int hd = wave;
    f32x4 ov[4][2];
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      const float* src = p.att + (long)tokl[16 * I + c] * C + D * hd + 4 * g;
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        const int d0 = 16 * jd + 4 * g;            // a piece past D reads the lane's first piece (zeroed below)
        const float* q = src + (d0 + 2 <= D ? 16 * jd : -4 * g);
        ov[I][jd] = d0 + 4 <= D ? *(w3_gp4)q : f32x4{ldg_f(q), ldg_f(q + 1), 0.f, 0.f};
      }
    }
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      const int row = 16 * I + c;
      const float asc = pow2_scale(__builtin_bit_cast(float, amax[row]));
      if (hd == 0 && g == 0) rinva[row] = pow2_inv(asc);
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        const int d0 = 16 * jd + 4 * g;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          if (d0 + 2 * h2 < D) {                       // (D is even: a pair is valid or not as a whole)
            const int k = D * hd + d0 + 2 * h2, s6 = k >> 5, kk = k & 31;
            unsigned hh, ll;
            split2_pair(ov[I][jd][2 * h2] * asc, ov[I][jd][2 * h2 + 1] * asc, hh, ll);
            unsigned char* dst = smem + s6 * AST + a_slot(row, kk >> 3) * 16 + (kk & 7) * 2;
            *(unsigned*)dst = hh;
            if (!AMP) *(unsigned*)(dst + APL) = ll;
          }
        }
      }
    }
  }
  {
    const int npad = (p.Kp - C) >> 1;                // channel pairs past C in the last stages: zeros
    for (int idx = tid; idx < 64 * npad; idx += NT) {
      const int row = idx / npad, k = C + 2 * (idx - row * npad), s6 = k >> 5, kk = k & 31;
      unsigned char* dst = smem + s6 * AST + a_slot(row, kk >> 3) * 16 + (kk & 7) * 2;
      *(unsigned*)dst = 0u;
      if (!AMP) *(unsigned*)(dst + APL) = 0u;
    }
  }
  SR_TS(5)
  // Phase 4 needs the thread's position again: derived from the thread index so that little lives through the phases above
  int t4 = tid;
  asm volatile("" : "+v"(t4));
  const int lane4 = t4 & 63, c4 = lane4 & 15, g4 = lane4 >> 4;
  unsigned boffp[NJ];
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    const int col = min(wave * CW + jt * 16 + c4, C - 1);
    boffp[jt] = (unsigned)(((g4 >> 1) * C + col) * 32 + (g4 & 1) * 16);
  }
  u32x4 fb0[NJ][2], fb1[NJ][2], fb2[NJ][2];
  auto load_bp = [&](int s, u32x4 (&fb)[NJ][2]) {
    const char* base = (const char*)p.Wproj + (long)(2 * rst(min(s, nst - 1))) * C * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * planep + boffp[jt]);
  };
  load_bp(0, fb0);
  load_bp(1, fb1);
  load_bp(2, fb2);
  const int arow4 = (t4 >> 2) & 63, akq4 = t4 & 3;
  const int atok4 = tokl[arow4];
  const int a_off40 = a_slot(c4, g4) * 16;
  __syncthreads();
  SR_TS(7)

  // ---------------- phase 4: proj (k_nth2's loop): acc2[i][j] = rows 16 i + 4 g + e, columns CW wave + 16 j + c
  f32x4 acc2[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma2 = [&](int s6, const u32x4 (&fb)[NJ][2]) {
    const unsigned char* sa = smem + rst(s6) * AST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 fa[2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[pl] = *(const u32x4*)(sa + pl * APL + a_off40 + 1024 * i);
#define SR_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc2[i][j] = mfma16h(fa[PA], fb[j][PB], acc2[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
  };
  // the residual pieces of the row-major epilogue travel while the MFMAs run
  f32x4 rv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * NT + t4, row = idx / 48, col = (idx - row * 48) * 4;
    rv[it] = *(const f32x4*)(p.X + (long)tokl[row] * C + min(col, C - 4));
  }
  __builtin_amdgcn_sched_barrier(0);
  mma2(0, fb0); load_bp(3, fb0); __builtin_amdgcn_sched_barrier(0);
  mma2(1, fb1); load_bp(4, fb1); __builtin_amdgcn_sched_barrier(0);
  mma2(2, fb2); load_bp(5, fb2); __builtin_amdgcn_sched_barrier(0);
  mma2(3, fb0); __builtin_amdgcn_sched_barrier(0);
  mma2(4, fb1); __builtin_amdgcn_sched_barrier(0);
  mma2(5, fb2); __builtin_amdgcn_sched_barrier(0);
  SR_TS(8)

  // ---------------- the output tile, row-major in LDS (block exponents undone: exact powers of two)
  __syncthreads();
  float* const T = (float*)smem;
  {
    float wv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wv[j] = winvp[min(wave * CW + 16 * j + c4, C - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ri = rinva[16 * i + 4 * g4 + e];
#pragma unroll
        for (int j = 0; j < NJ; ++j) T[(16 * i + 4 * g4 + e) * TP + wave * CW + 16 * j + c4] = acc2[i][j][e] * (ri * wv[j]);
      }
  }
  __syncthreads();
  // out = x + s (acc + bp): 16-byte pieces in row-major order
  const float dps = p.rowscale ? p.rowscale[geo.b] : 1.f;
  {
    int prow[NIT], pcol[NIT], ptok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * NT + t4;
      prow[it] = idx / 48;
      pcol[it] = (idx - prow[it] * 48) * 4;
      ptok[it] = tokl[prow[it]];
      if (pcol[it] >= C) prow[it] = -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (prow[it] >= 0) {
        float* tp = T + prow[it] * TP + pcol[it];
        f32x4 v = *(const f32x4*)tp;
        const f32x4 bv = p.bproj ? *(const f32x4*)(p.bproj + pcol[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + bv[e]) * dps + rv[it][e];
        *(f32x4*)(p.out + (long)ptok[it] * C + pcol[it]) = v;
        if (p.stats_out) *(f32x4*)tp = v;
      }
    }
  }
  if (p.stats_out) {
    // {mean, rstd} of the out rows for the next LayerNorm (two-pass, eps 1e-5, biased variance): four lanes per row
    __syncthreads();
    if (wave >= 4) return;
    const int row = arow4, q = akq4;
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
    if (q == 0) *(float2*)(p.stats_out + 2 * (long)atok4) = float2{mean, rsqrtf(s2 * (1.0f / (float)C) + 1e-5f)};
  }
  SR_TS(9)
}

}  // namespace

#ifdef SRHIP_EXPERIMENTS
SR_DEBUG_EXPORT int srhip_wmsa_debug_buffer(long long* buf) { g_wmsa_dbg = buf; return 0; }   // [blocks][waves][16] stamps of the 100 MHz wall clock
#endif

int sr_wmsa_f16(WmsaF16Args& p, hipStream_t st) {
  SR_REQUIRE(p.C % 4 == 0 && p.C >= 4 && p.C <= 192, "wmsa_f16x2: C = %d (multiple of 4, <= 192)", p.C);
  SR_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && p.H % 8 == 0 && p.W % 8 == 0,
             "wmsa_f16x2: H, W must be positive multiples of the 8x8 window (H=%d W=%d)", p.H, p.W);
  SR_REQUIRE(p.heads > 0 && p.heads <= 8 && p.C % p.heads == 0, "wmsa_f16x2: heads = %d (<= 8, dividing C)", p.heads);
  SR_REQUIRE(p.shift == 0 || p.shift == 4, "wmsa_f16x2: shift must be 0 or 4 (got %d)", p.shift);
  SR_REQUIRE(p.shift == 0 || (p.H > 8 && p.W > 8), "wmsa_f16x2: shifted windows need H, W > 8");
  SR_REQUIRE((long)p.B * p.H * p.W < (1L << 31), "wmsa_f16x2: more than 2^31 tokens");
  const int D = p.C / p.heads;
  SR_REQUIRE(D == 30 || D == 10 || D == 16 || D == 32, "wmsa_f16x2: head dim %d not built", D);
  SR_REQUIRE(p.qkv || p.heads == 5 || p.heads == 6,
             "wmsa_f16x2: qkv may be omitted (inference) only with 5 or 6 heads: the four-wave kernel reads it back");
  p.Kp = sr_kp(p.C);
#ifdef SRHIP_EXPERIMENTS
  p.dbg = g_wmsa_dbg;
  { const char* e = sr_getenv("SRHIP_WMSA_STAGGER"); p.stagger = e ? atoi(e) : 0; }
#endif
  p.scale = 1.0f / sqrtf((float)D);
  { static const int krot = [] { const char* e = sr_getenv("SRHIP_WMSA_ROT"); return e ? atoi(e) : 1; }(); p.k_rot = krot; }
  const int nwin = p.B * (p.H / 8) * (p.W / 8);
  bool six = p.heads == 5 || p.heads == 6;
  const bool amp = sr_matmul_mode() == 1;          // inference under --amp (5 / 6 heads: one product of the leading planes)
  if (const char* e = sr_getenv("SRHIP_WMSA_NW")) six = six && e[0] == '6';      // experiment builds only
#define SR_WA(D_)                                                                                              \
  if (D == D_) {                                                                                               \
    if (six) {                                                                                                 \
      static bool attr = false;                                                                                \
      if (!attr) {                                                                                             \
        if (hipFuncSetAttribute((const void*)k_wmsa_f16h<D_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                2 * WMSA_LDSH) != hipSuccess ||                                                \
            hipFuncSetAttribute((const void*)k_wmsa_f16h<D_, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                2 * WMSA_LDSH) != hipSuccess)                                                  \
          return sr_fail(-5, "wmsa_f16x2: cannot reserve %d bytes of LDS", 2 * WMSA_LDSH);                     \
        attr = true;                                                                                           \
      }                                                                                                        \
      if (amp) hipLaunchKernelGGL((k_wmsa_f16h<D_, 2, true>), dim3(sr_cdiv(nwin, 2)), dim3(768), 2 * WMSA_LDSH, st, p); \
      else hipLaunchKernelGGL((k_wmsa_f16h<D_, 2>), dim3(sr_cdiv(nwin, 2)), dim3(768), 2 * WMSA_LDSH, st, p);  \
    } else {                                                                                                   \
      hipLaunchKernelGGL((k_wmsa_f16<D_, 4, 1>), dim3(nwin), dim3(256), WMSA_LDS, st, p);                      \
    }                                                                                                          \
  }
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("k_wmsa_f16");
  return 0;
}
