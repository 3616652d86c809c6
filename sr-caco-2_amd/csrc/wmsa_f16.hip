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

}  // namespace

#ifdef SRHIP_EXPERIMENTS
extern "C" int srhip_wmsa_debug_buffer(long long* buf) { g_wmsa_dbg = buf; return 0; }   // [blocks][waves][16] stamps of the 100 MHz wall clock
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
  p.Kp = sr_kp(p.C);
#ifdef SRHIP_EXPERIMENTS
  p.dbg = g_wmsa_dbg;
#endif
  p.scale = 1.0f / sqrtf((float)D);
  const int nwin = p.B * (p.H / 8) * (p.W / 8);
  bool six = p.heads == 5 || p.heads == 6;
  if (const char* e = sr_getenv("SRHIP_WMSA_NW")) six = six && e[0] == '6';      // experiment builds only
#define SR_WA(D_)                                                                                              \
  if (D == D_) {                                                                                               \
    if (six) {                                                                                                 \
      static bool attr = false;                                                                                \
      if (!attr) {                                                                                             \
        if (hipFuncSetAttribute((const void*)k_wmsa_f16<D_, 6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                2 * WMSA_LDS) != hipSuccess)                                                   \
          return sr_fail(-5, "wmsa_f16x2: cannot reserve %d bytes of LDS", 2 * WMSA_LDS);                      \
        attr = true;                                                                                           \
      }                                                                                                        \
      hipLaunchKernelGGL((k_wmsa_f16<D_, 6, 2>), dim3(sr_cdiv(nwin, 2)), dim3(768), 2 * WMSA_LDS, st, p);      \
    } else {                                                                                                   \
      hipLaunchKernelGGL((k_wmsa_f16<D_, 4, 1>), dim3(nwin), dim3(256), WMSA_LDS, st, p);                      \
    }                                                                                                          \
  }
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("k_wmsa_f16");
  return 0;
}
