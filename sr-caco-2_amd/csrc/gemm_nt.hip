// NT contraction on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32):
//
//   GEMM : C[M,N] = epi( pro(A)[M,K] . W[N,K]^T )           (Linear fwd, and Linear
//          bwd-data through a transposed weight copy)
//   CONV : 3x3 / stride 1 / pad 1 convolution over an NHWC image as an implicit
//          GEMM: M = output pixels, N = Cout, K = 9 taps x Cin; weights are the
//          tap-major packed copy Wp[tap][Cout][Cin] (conv fwd) or the flipped /
//          transposed copy (conv bwd-data).
//
// Replaces aten linear / conv2d calls of dlib/models/network_swinir.py:40-43,
// 148,177,544,786,850,700 and dlib/models/network_nlsn.py:38-41.
//
// Tiling: 256 threads = 4 waves as 2(M) x 2(N); each wave owns WM x WN MFMA
// tiles of 32x32.  A and W chunks (BK wide in K) are staged through LDS with a
// row pitch SA = 4*odd floats, so that the ds_read_b128 fragment reads (one
// row per lane, 4 consecutive k) are bank-conflict free.  Each lane half
// (lane>>5) owns 4 of every 8 consecutive k, which it feeds to 4 successive
// MFMAs; A and W use the same k permutation, so products pair up correctly.
// Global loads for chunk i+1 are issued before the MFMAs of chunk i (register
// prefetch) and written to LDS after them.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "nt_epi.h"

namespace {

template <int WM, int WN, int BK, bool CONV>
__global__ void __launch_bounds__(256, 2) k_nt(NtArgs p) {
  constexpr int SA = (BK % 8 == 4) ? BK : BK + 4;  // pitch = 4*odd floats
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int TROWS = BM / 16;                   // conv: image rows per tile
  constexpr int AROWS = CONV ? (TROWS + 2) * 18 : BM;
  constexpr int KV = BK / 4;                       // float4 per staged row
  constexpr int A_N = AROWS * KV, B_N = BN * KV;
  constexpr int A_IT = (A_N + 255) / 256, B_IT = (B_N + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[(AROWS + BN) * SA];
  float* As = smem;
  float* Bs = smem + AROWS * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);

  // block origin
  int m0 = 0, img = 0, y0 = 0, x0 = 0;
  if (CONV) {
    int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // neighbouring tiles (shared halos) in one L2
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; img = t / p.tiles_y;
    y0 = ty * TROWS; x0 = tx * 16;
  } else {
    m0 = blockIdx.x * BM;
    if (p.zcount > 1) {                            // one of a batch of GEMMs: shift the operand bases
      const int z0 = blockIdx.z / p.zdiv, z1 = blockIdx.z - z0 * p.zdiv;
      p.A += z0 * p.zA[0] + z1 * p.zA[1];
      p.W += z0 * p.zW[0] + z1 * p.zW[1];
      p.C += z0 * p.zC[0] + z1 * p.zC[1];
    }
  }

  // Co-resident blocks start in lockstep and would all load, compute and store
  // at the same moments; delaying every other block by part of a main loop lets
  // one block's loads / epilogue overlap its neighbour's MFMA phase.
  if (p.stagger > 0 && (blockIdx.x + blockIdx.y * gridDim.x) >= (gridDim.x * gridDim.y) / 2) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- staging invariants, hoisted out of the K loop --------------------------
  // Each thread moves the same float4 slots of the A and W tiles in every chunk;
  // only the K offset changes.  Rows outside the problem are CLAMPED to a valid
  // row instead of zero-filled (an output element depends only on its own A row
  // and W row, and out-of-range rows/columns are never stored), so a chunk's
  // loads are base(SGPR, advanced per chunk) + byte offset(VGPR, fixed): no
  // per-chunk address arithmetic on the vector ALU.
  f32x4 ra[A_IT], rb[B_IT];
  float2 rst[A_IT];               // {mean, rstd} of the A row (or the neutral pair)
  unsigned offA[A_IT], offB[B_IT];   // byte offsets at k = 0
  bool inA[A_IT];                    // conv: halo pixel inside the image (else zero)
  // LDS float offset of staging slot `it`: the image is linear in the slot index
  // when the pitch equals the chunk width (BK = 36, 60), so it costs no register
  auto lds_off = [&](int it, int n_slots) -> int {
    const int idx = min(tid + it * 256, n_slots - 1);
    if (SA == BK) return idx * 4;
    const int row = idx / KV;
    return row * SA + (idx - row * KV) * 4;
  };
  auto slot_k = [&](int it, int n_slots) -> int {   // k offset of the slot inside the chunk
    const int idx = min(tid + it * 256, n_slots - 1);
    return (idx - (idx / KV) * KV) * 4;
  };
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int idx = min(tid + it * 256, A_N - 1);
    const int row = idx / KV, c4 = idx - row * KV;
    inA[it] = true;
    const float* sp = k_sr_neutral;
    if (CONV) {
      const int hy = row / 18, hx = row - hy * 18;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
      offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
    } else {
      const int gm = min(m0 + row, p.M - 1);
      offA[it] = (unsigned)(gm * (int)p.lda + c4 * 4) * 4u;
      if (p.a_mode == 1) sp = p.ln_stats + 2 * gm;
    }
    rst[it] = *(const float2*)sp;
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = min(tid + it * 256, B_N - 1);
    const int row = idx / KV, c4 = idx - row * KV;
    offB[it] = (unsigned)((n0 + min(row, nvalid - 1)) * (int)p.ldw + c4 * 4) * 4u;
  }

  // Loaded values are not touched here (a select on a fresh load forces an immediate
  // s_waitcnt and serialises the prefetch): K-tail / halo lanes read a valid address
  // and are zeroed when the chunk is staged.
  int kc_a = 0, kc_b = 0;                    // chunk held by ra / rb
  auto load_a = [&](int kc) {
    const char* base = (const char*)(p.A + kc * BK);
    const bool ktail = kc * BK + BK > p.K;           // block-uniform
    kc_a = kc;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int sk = slot_k(it, A_N);
      const bool oob = ktail && kc * BK + sk >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - sk * 4u : offA[it]));
    }
  };
  auto load_b = [&](int kc, int tap) {
    const char* base = (const char*)(p.W + (long)tap * p.wtap + kc * BK);
    const bool ktail = kc * BK + BK > p.K;
    kc_b = kc;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int sk = slot_k(it, B_N);
      const bool oob = ktail && kc * BK + sk >= p.K;
      rb[it] = *(const f32x4*)((oob ? (const char*)(p.W + (long)tap * p.wtap) : base) + (oob ? offB[it] - sk * 4u : offB[it]));
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      if (A_N % 256 == 0 || tid + it * 256 < A_N) {
        f32x4 v = ra[it];
        if (!CONV) {
          if (p.a_mode == 1) {          // LayerNorm prologue: (x-mean)*rstd
            const float mu = rst[it].x, rs = rst[it].y;
            v.x = (v.x - mu) * rs; v.y = (v.y - mu) * rs; v.z = (v.z - mu) * rs; v.w = (v.w - mu) * rs;
          } else if (p.a_mode == 2) {   // GELU prologue
            v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
          }
        }
        // halo pixels outside the image and the K tail contribute exact zeros
        if ((CONV && !inA[it]) || (kc_a * BK + BK > p.K && kc_a * BK + slot_k(it, A_N) >= p.K))
          v = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(As + lds_off(it, A_N)) = v;
      }
    }
  };
  auto store_b = [&]() {
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_N % 256 == 0 || tid + it * 256 < B_N) {
        f32x4 v = rb[it];
        if (kc_b * BK + BK > p.K && kc_b * BK + slot_k(it, B_N) >= p.K) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(Bs + lds_off(it, B_N)) = v;
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // fragment base offsets (floats)
  int a_off[WM], b_off[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mt = wm * WM + i;
    if (CONV) a_off[i] = ((2 * mt + (r >> 4)) * 18 + (r & 15)) * SA + 4 * h;
    else a_off[i] = (mt * 32 + r) * SA + 4 * h;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) b_off[j] = ((wn * WN + j) * 32 + r) * SA + 4 * h;

  const int nkc = (p.K + BK - 1) / BK;
  const int ntap = CONV ? 9 : 1;
  const int niter = nkc * ntap;

  load_a(0);
  load_b(0, 0);
  for (int it = 0; it < niter; ++it) {
    const int kc = it / ntap, tap = it - kc * ntap;
    __syncthreads();                       // everyone done reading LDS
    if (!(p.dbg & 2)) {
      if (!CONV || tap == 0) store_a();
      store_b();
    }
    __syncthreads();
    if (it + 1 < niter && !(p.dbg & 1)) {  // prefetch next chunk into registers
      const int kc1 = (it + 1) / ntap, tap1 = (it + 1) - kc1 * ntap;
      if (!CONV || tap1 == 0) load_a(kc1);
      load_b(kc1, tap1);
    }
    const int toff = CONV ? ((tap / 3) * 18 + (tap % 3)) * SA : 0;
    if (p.dbg & 4) continue;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = *(const f32x4*)(As + a_off[i] + toff + g * 8);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[j] = *(const f32x4*)(Bs + b_off[j] + g * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = mfma32(fa[i][t], fb[j][t], acc[i][j]);
    }
    if (BK % 8 == 4) {                     // 4-wide tail: each lane half owns 2 k
      float2 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i)
        fa[i] = *(const float2*)(As + a_off[i] - 4 * h + toff + (BK - 4) + 2 * h);
#pragma unroll
      for (int j = 0; j < WN; ++j)
        fb[j] = *(const float2*)(Bs + b_off[j] - 4 * h + (BK - 4) + 2 * h);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = mfma32(fa[i].x, fb[j].x, acc[i][j]);
          acc[i][j] = mfma32(fa[i].y, fb[j].y, acc[i][j]);
        }
    }
  }

  // ---- epilogue ----
  if (p.dbg & 8) return;
  nt_epilogue<WM, WN, CONV>(p, acc, lane, wm, wn, n0, nvalid, m0, img, y0, x0);
}

template <int WM, int WN, int BK, bool CONV>
int launch_nt(const NtArgs& p, hipStream_t st) {
  constexpr int BM = 64 * WM;
  dim3 grid;
  if (CONV) grid = dim3(p.tiles_x * p.tiles_y * p.batch, sr_cdiv(p.N, p.n_tile));
  else grid = dim3(sr_cdiv(p.M, BM), sr_cdiv(p.N, p.n_tile), p.zcount > 1 ? p.zcount : 1);
  hipLaunchKernelGGL((k_nt<WM, WN, BK, CONV>), grid, dim3(256), 0, st, p);
  SR_LAUNCH_CHECK("k_nt");
  return 0;
}

int nt_env(const char* name, int dflt) {
  const char* e = sr_getenv(name);
  return e ? atoi(e) : dflt;
}

template <bool CONV>
int dispatch_nt(NtArgs& p, hipStream_t st) {
  // N tile: 180-wide problems (SwinIR embed 180 and its multiples) run as
  // 192-column blocks with 180 valid; everything else in 64/128/192 columns.
  int wn;
  if (p.N % 180 == 0) { p.n_tile = 180; wn = 3; }
  else if (p.N <= 64) { p.n_tile = 64; wn = 1; }
  else if (p.N <= 128 || p.N % 128 == 0) { p.n_tile = 128; wn = 2; }
  else { p.n_tile = 192; wn = 3; }
  // K chunk: 60 when it divides K (SwinIR 180/360/540: 3/6/9 chunks instead of
  // 5/10/15 -- fewer barrier / staging rounds per block), else 36, or 32 for
  // power-of-two channel counts (EDSR)
  int bk = ((p.K % 36 == 0) || (p.K % 32 != 0)) ? 36 : 32;
  if (!CONV && p.K % 60 == 0 && nt_env("SRHIP_NT_BK60", 1)) bk = 60;   // (64-row tiles only, below)
  // M tile: prefer 128 rows, fall back to 64 when that leaves < 2 blocks per CU
  long rows = CONV ? 0 : p.M;
  long blocks128;
  if (CONV) {
    const int tx = sr_cdiv(p.Wd, 16);
    blocks128 = (long)tx * sr_cdiv(p.H, 8) * p.batch;
  } else {
    blocks128 = sr_cdiv(rows, 128);
  }
  blocks128 *= sr_cdiv(p.N, p.n_tile);
  // 128-row tiles only for the two-column-block case (N = 360): measured faster
  // there; with three column blocks (N = 540) 64-row tiles win (88 vs 104 us)
  int wm = (blocks128 >= 512 && sr_cdiv(p.N, p.n_tile) <= 2) ? 2 : 1;
  const int force_wm = nt_env("SRHIP_NT_WM", 0);
  if (force_wm == 1 || force_wm == 2) wm = force_wm;
  p.stagger = nt_env("SRHIP_NT_STAGGER", 0);
  if (wm == 2 && bk == 60) bk = 36;      // <2,3,60> spills
  if (CONV) {
    p.tiles_x = sr_cdiv(p.Wd, 16);
    p.tiles_y = sr_cdiv(p.H, wm == 2 ? 8 : 4);
    p.xcd_order = nt_env("SRHIP_CONV_XCD_F32", 0);   // measured on EDSR x8 (64 channels, up to 512x512): 623 vs 679 patches/s with it on
  }
#define SR_NT_CASE(WM_, WN_, BK_) \
  if (wm == WM_ && wn == WN_ && bk == BK_) return launch_nt<WM_, WN_, BK_, CONV>(p, st);
  SR_NT_CASE(1, 1, 36) SR_NT_CASE(1, 2, 36) SR_NT_CASE(1, 3, 36)
  SR_NT_CASE(2, 1, 36) SR_NT_CASE(2, 2, 36) SR_NT_CASE(2, 3, 36)
  SR_NT_CASE(1, 1, 32) SR_NT_CASE(1, 2, 32) SR_NT_CASE(1, 3, 32)
  SR_NT_CASE(2, 1, 32) SR_NT_CASE(2, 2, 32) SR_NT_CASE(2, 3, 32)
  if (!CONV) {
    if (wm == 1 && wn == 3 && bk == 60) return launch_nt<1, 3, 60, false>(p, st);
    if (wm == 2 && wn == 3 && bk == 60) return launch_nt<2, 3, 60, false>(p, st);
    if (wm == 1 && wn == 1 && bk == 60) return launch_nt<1, 1, 60, false>(p, st);
    if (wm == 1 && wn == 2 && bk == 60) return launch_nt<1, 2, 60, false>(p, st);
    if (wm == 2 && wn == 1 && bk == 60) return launch_nt<2, 1, 60, false>(p, st);
    if (wm == 2 && wn == 2 && bk == 60) return launch_nt<2, 2, 60, false>(p, st);
  }
#undef SR_NT_CASE
  return sr_fail(-22, "nt: no kernel for wm=%d wn=%d", wm, wn);
}

}  // namespace

static int nt_dbg() {
  const char* e = sr_getenv("SRHIP_NT_DBG");
  return e ? atoi(e) : 0;
}


int sr_gemm_nt(NtArgs& p, hipStream_t st) {
  p.dbg = nt_dbg();
  SR_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0,
             "gemm_nt: K, lda, ldw must be multiples of 4 (K=%d lda=%ld ldw=%ld)", p.K, p.lda, p.ldw);
  SR_REQUIRE(p.M > 0 && p.N > 0, "gemm_nt: empty problem");
  SR_REQUIRE((long)p.M * p.lda < (1L << 29) && (long)p.N * p.ldw < (1L << 29),
             "gemm_nt: operand larger than 2 GiB (32-bit staging offsets)");
  return dispatch_nt<false>(p, st);
}

int sr_conv3x3_nt(NtArgs& p, hipStream_t st) {
  SR_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0,
             "conv3x3: Cin, lda, ldw must be multiples of 4 (Cin=%d)", p.K);
  SR_REQUIRE(p.batch > 0 && p.H > 0 && p.Wd > 0, "conv3x3: empty image");
  p.M = p.batch * p.H * p.Wd;
  SR_REQUIRE((long)p.M * p.lda < (1L << 29) && 9L * p.N * p.ldw < (1L << 29),
             "conv3x3: operand larger than 2 GiB (32-bit staging offsets)");
  p.dbg = nt_dbg();
  return dispatch_nt<true>(p, st);
}
