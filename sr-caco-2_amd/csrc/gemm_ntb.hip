// NT contraction with f32 operands split three ways into bf16 and multiplied on
// the bf16 MFMA (v_mfma_f32_32x32x16_bf16, f32 accumulate):
//
//   x = x_h + x_m + x_l   (each part the bf16 rounding of the remaining residual:
//                          3 x 8 significant bits = the 24 bits of an f32)
//   a*b ~= a_m*b_m + a_h*b_l + a_l*b_h + a_h*b_m + a_m*b_h + a_h*b_h
//
// The three dropped cross terms (m*l, l*m, l*l) are below 2^-24 relative, i.e.
// at the rounding level of the f32 product itself; the products of bf16 pairs
// are exact in the MFMA and the sums run in f32, so the result carries f32
// accuracy at 6 bf16 MFMAs (6 x 32 cycles per 32x32x16) instead of 8 f32 MFMAs
// (8 x 64 cycles): 2.67x the matrix-core rate of gemm_nt.hip.
//
// Same two problem shapes and the same prologues / epilogues as gemm_nt.hip
// (Linear fwd / bwd-data, 3x3 conv as implicit GEMM over an NHWC halo tile).
// W arrives PRE-SPLIT because it is reused by every block: three bf16 planes in
// SUB-CHUNK-MAJOR order [3][Kp/16][rows][16] (Kp = K rounded up to 32, zero filled;
// rows = N, or 9*Cout tap-major for the conv), so the 64*WN rows x 64 B a block
// stages per plane and chunk are one contiguous run (every wave-level load
// instruction reads 1 KB of consecutive bytes instead of sixteen 64-byte pieces); the activation
// operand is split while it is staged into LDS (once per chunk; for the conv
// once per 9 taps).
//
// LDS: per plane [row][32 k] bf16 with an 80-byte row pitch (20 dwords: the 16
// lanes of a ds_read_b128 phase hit 16 distinct 4-dword bank groups).
// Fragment of lane (r = lane&31, h = lane>>5) for k-step s: 8 consecutive k at
// s*16 + 8*h -- the operand layout of the 32x32x16 MFMA for both A and B.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "nt_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BKB = 32;          // k per chunk
constexpr int PITCH = 80;        // bytes per LDS row (32 bf16 + 16 B pad)

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// AMP = true: reduced-precision inference (the reference's --amp autocast, model_plain.py:322-327): ONE bf16
// product of the leading planes instead of six -- only plane 0 of either operand is staged and read.
template <int WM, int WN, bool CONV, bool AMP = false>
__global__ void __launch_bounds__(256, 2) k_ntb(NtArgs p) {
  constexpr int NPL = AMP ? 1 : 3;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int TROWS = BM / 16;
  constexpr int AROWS = CONV ? (TROWS + 2) * 18 : BM;
  constexpr int A_N = AROWS * 8;                 // float4 slots per chunk
  constexpr int A_IT = (A_N + 255) / 256;
  constexpr int B_N = NPL * BN * 4;              // 16-byte slots per chunk (3 planes; AMP: the leading one)
  constexpr int B_IT = (B_N + 255) / 256;
  constexpr int A_PLANE = AROWS * PITCH, B_PLANE = BN * PITCH;
  // 64-column conv tiles: W double-buffered in LDS, ONE barrier per (chunk, tap) iteration (k_ntb_dbw below)
  constexpr bool DBW = CONV && WN == 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 3 * A_PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);

  int m0 = 0, img = 0, y0 = 0, x0 = 0;
  if (CONV) {
    int t = p.xcd_order ? sr_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // neighbouring tiles (shared halos) in one L2
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; img = t / p.tiles_y;
    y0 = ty * TROWS; x0 = tx * 16;
  } else {
    m0 = blockIdx.x * BM;
  }

  // Co-resident blocks start in lockstep and would load, compute and store at the
  // same moments; delaying the second half of the grid lets one block's memory
  // phases overlap its neighbour's MFMA phase (see gemm_nt.hip).
  if (p.stagger > 0 && (blockIdx.x + blockIdx.y * gridDim.x) >= (gridDim.x * gridDim.y) / 2) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(8);      // x 512 cycles
  }

  // ---- staging invariants (see gemm_nt.hip: clamped rows, fixed byte offsets) ----
  f32x4 ra0[A_IT], ra1[A_IT];     // GEMM: A chunks it and it+1 in flight (two register sets)
  u32x4 rb[B_IT];
  float2 rst[A_IT];
  unsigned offA[A_IT], offB[B_IT];
  bool inA[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int idx = min(tid + it * 256, A_N - 1);
    const int row = idx >> 3, c4 = idx & 7;
    inA[it] = true;
    const float* sp = k_sr_neutral;
    if (CONV) {
      const int hy = row / 18, hx = row - hy * 18;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
      if (p.ps == 2) offA[it] = (unsigned)(((img * 2 * p.H + 2 * yc) * 2 * p.Wd + 2 * xc) * (int)p.lda + c4 * 4) * 4u;
      else offA[it] = (unsigned)(((img * p.H + yc) * p.Wd + xc) * (int)p.lda + c4 * 4) * 4u;
    } else {
      const int gm = min(m0 + row, p.M - 1);
      offA[it] = (unsigned)(gm * (int)p.lda + c4 * 4) * 4u;
      if (p.a_mode == 1) sp = p.ln_stats + 2 * gm;
    }
    rst[it] = ldg_f2(sp);
  }
  const long wrows = (long)(CONV ? 9 : 1) * p.N;                    // rows of the weight matrix
  const long plane_bytes = wrows * p.Kp * 2;                        // one bf16 plane of W
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = min(tid + it * 256, B_N - 1);
    const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4);
    const int row = rem >> 2, q = rem & 3;
    // planes are stored in 16-k sub-chunks [Kp/16][rows][16]: unit q (8 k) lives in sub-chunk q/2
    offB[it] = (unsigned)(pl * plane_bytes + (long)(q >> 1) * wrows * 32 +
                          (long)(n0 + min(row, nvalid - 1)) * 32 + (q & 1) * 16);
  }

  // Loaded values are not touched here (a select on a fresh load would force an
  // immediate s_waitcnt and serialise the prefetch): K-tail / halo lanes read a valid
  // address and are zeroed when the chunk is staged (store_a).
  auto load_a = [&](int kc, f32x4 (&ra)[A_IT]) {
    long koff = kc * BKB;
    if (CONV && p.ps == 2) {     // chunk kc = channels c0.. of sub-pixel sp of the shuffled image (K/4 is a multiple of 32)
      const int fk = p.K >> 2, sp = (kc * BKB) / fk, c0 = kc * BKB - sp * fk;
      koff = ((long)(sp >> 1) * 2 * p.Wd + (sp & 1)) * p.lda + c0;
    }
    const char* base = (const char*)(p.A + koff);
    const bool ktail = kc * BKB + BKB > p.K;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int c4 = min(tid + it * 256, A_N - 1) & 7;
      const bool oob = ktail && kc * BKB + c4 * 4 >= p.K;
      ra[it] = *(const f32x4*)((oob ? (const char*)p.A : base) + (oob ? offA[it] - c4 * 16u : offA[it]));   // oob: k = 0 of the row
    }
  };
  auto load_b = [&](int kc, int tap) {
    const char* base = (const char*)p.Wb + ((long)kc * 2 * wrows + (long)tap * p.N) * 32;   // sub-chunk 2*kc, row tap*N
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rb[it] = *(const u32x4*)(base + offB[it]);
  };
  auto store_a = [&](const f32x4 (&ra)[A_IT], int kc) {
    const bool ktail = kc * BKB + BKB > p.K;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      if (A_N % 256 == 0 || tid + it * 256 < A_N) {
        const int idx = tid + it * 256;
        f32x4 v = ra[it];
        if (!CONV) {
          if (p.a_mode == 1) {
            const float mu = rst[it].x, rs = rst[it].y;
            v.x = (v.x - mu) * rs; v.y = (v.y - mu) * rs; v.z = (v.z - mu) * rs; v.w = (v.w - mu) * rs;
          } else if (p.a_mode == 2) {
            v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
          }
        }
        // halo pixels outside the image and the K tail contribute exact zeros
        if ((CONV && !inA[it]) || (ktail && kc * BKB + (idx & 7) * 4 >= p.K)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(v.x, v.y, h0, m0_, l0);
        split3_pair(v.z, v.w, h1, m1, l1);
        unsigned char* dst = As + (idx >> 3) * PITCH + (idx & 7) * 8;
        *(u32x2*)(dst) = u32x2{h0, h1};
        if (!AMP) {
          *(u32x2*)(dst + A_PLANE) = u32x2{m0_, m1};
          *(u32x2*)(dst + 2 * A_PLANE) = u32x2{l0, l1};
        }
      }
    }
  };
  auto store_b = [&](int buf = 0) {
    unsigned char* dstb = Bs + buf * (3 * B_PLANE);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (B_N % 256 == 0 || tid + it * 256 < B_N) {
        const int idx = tid + it * 256;
        const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4);
        *(u32x4*)(dstb + pl * B_PLANE + (rem >> 2) * PITCH + (rem & 3) * 16) = rb[it];
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

  int a_off[WM], b_off[WN];      // byte offsets of the lane's fragment rows
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mt = wm * WM + i;
    if (CONV) a_off[i] = ((2 * mt + (r >> 4)) * 18 + (r & 15)) * PITCH + 16 * h;
    else a_off[i] = (mt * 32 + r) * PITCH + 16 * h;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) b_off[j] = ((wn * WN + j) * 32 + r) * PITCH + 16 * h;

  const int nkc = (p.K + BKB - 1) / BKB;
  const int ntap = CONV ? 9 : 1;
  const int niter = nkc * ntap;

  // The A operand comes from HBM and the loop is latency bound with one chunk in
  // flight: the GEMM keeps TWO A chunks in flight (W chunks are L2 hits, one is enough)
  // timing build (SRHIP_NT_DBG bit 64): s_memtime stamps of the loop phases of one wave
  long tk_b1 = 0, tk_store = 0, tk_b2 = 0, tk_load = 0, tk_mma = 0, tk_prev = 0;
  const bool stamp = (p.dbg & 64) != 0;
  auto tick = [&](long& acc_t) {
    if (stamp) { const long t = (long)__builtin_amdgcn_s_memtime(); acc_t += t - tk_prev; tk_prev = t; }
  };
  auto iter = [&](int it, f32x4 (&ra)[A_IT]) {
    const int kc = it / ntap, tap = it - kc * ntap;
    __syncthreads();
    tick(tk_b1);
    if (!(p.dbg & 2)) {
      if (!CONV || tap == 0) store_a(ra, kc);
      store_b();
    }
    tick(tk_store);
    __syncthreads();
    tick(tk_b2);
    // A operand of a later chunk (two ahead for the GEMM; the next channel chunk's halo
    // for the conv): few loads, their own block
    if (CONV) {
      if (it + 1 < niter && (it + 1) % ntap == 0 && !(p.dbg & 32)) load_a((it + 1) / ntap, ra);
    } else {
      if (it + 2 < niter && !(p.dbg & 32)) load_a(it + 2, ra);
    }
    tick(tk_load);
    if (p.dbg & 4) return;
    // ---- ONE basic block from here: the W loads of the next chunk and this chunk's MFMAs.
    // A wave that issues its 9 W loads back to back stalls ~1100 cycles on the full
    // memory queue (the CU's L1 fills 64 B/clk: 74 KB of W per CU and chunk) BEFORE its
    // MFMAs start -- measured with s_memtime; interleaved one load per four MFMAs the
    // queue drains while the matrix core works.  The last iteration reloads its own
    // chunk (no branch in the block).
    {
      const int itn = min(it + 1, niter - 1);
      const int kcn = itn / ntap;
      load_b(kcn, itn - kcn * ntap);
    }
    const int toff = CONV ? ((tap / 3) * 18 + (tap % 3)) * PITCH : 0;
#pragma unroll
    for (int s = 0; s < BKB / 16; ++s) {
      u32x4 fa[WM][3], fb[WN][3];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fa[i][pl] = *(const u32x4*)(As + pl * A_PLANE + a_off[i] + toff + s * 32);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fb[j][pl] = *(const u32x4*)(Bs + pl * B_PLANE + b_off[j] + s * 32);
      // small terms first; term-outer so that consecutive MFMAs hit different tiles
#define SR_TERM(PA, PB)                                                              \
  _Pragma("unroll") for (int i = 0; i < WM; ++i)                                     \
  _Pragma("unroll") for (int j = 0; j < WN; ++j)                                     \
    acc[i][j] = mfma_bf(fa[i][PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
    {
      constexpr int NMFMA = (BKB / 16) * (AMP ? 1 : 6) * WM * WN;
      constexpr int PER = NMFMA / B_IT > 0 ? NMFMA / B_IT : 1;
#pragma unroll
      for (int g = 0; g < B_IT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);     // PER MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // then one VMEM read
      }
    }
    tick(tk_mma);
  };
  // 64-column conv tiles (DBW): the W chunk of iteration it+1 is stored into the OTHER LDS buffer while iteration
  // it computes, so an iteration has ONE barrier instead of two (12-24 MFMAs per wave sat between two barriers:
  // the loop was barrier / latency bound, 28 % of the MFMA rate inside it).  The halo tile of A stays single
  // buffered: one extra barrier per channel chunk.  Same LDS budget class: two blocks per CU either way.
  auto iter_dbw = [&](int it, f32x4 (&ra)[A_IT]) {
    const int kc = it / ntap, tap = it - kc * ntap;
    if (it + 1 < niter) store_b((it + 1) & 1);                 // W(it+1): its buffer's last reader was iteration it-1
    if (tap == 0 && kc + 1 < nkc) load_a(kc + 1, ra);          // next channel chunk's halo: a whole chunk to land
    {
      const int itn = min(it + 2, niter - 1);
      const int kcn = itn / ntap;
      load_b(kcn, itn - kcn * ntap);
    }
    const unsigned char* Bc = Bs + (it & 1) * (3 * B_PLANE);
    const int toff = ((tap / 3) * 18 + (tap % 3)) * PITCH;
#pragma unroll
    for (int s = 0; s < BKB / 16; ++s) {
      u32x4 fa[WM][3], fb[WN][3];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fa[i][pl] = *(const u32x4*)(As + pl * A_PLANE + a_off[i] + toff + s * 32);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fb[j][pl] = *(const u32x4*)(Bc + pl * B_PLANE + b_off[j] + s * 32);
#define SR_TERM(PA, PB)                                                              \
  _Pragma("unroll") for (int i = 0; i < WM; ++i)                                     \
  _Pragma("unroll") for (int j = 0; j < WN; ++j)                                     \
    acc[i][j] = mfma_bf(fa[i][PA], fb[j][PB], acc[i][j]);
      if constexpr (AMP) {
        SR_TERM(0, 0)
      } else {
        SR_TERM(1, 1) SR_TERM(0, 2) SR_TERM(2, 0) SR_TERM(0, 1) SR_TERM(1, 0) SR_TERM(0, 0)
      }
#undef SR_TERM
    }
    {
      constexpr int NMFMA = (BKB / 16) * (AMP ? 1 : 6) * WM * WN;
      constexpr int PER = NMFMA / B_IT > 0 ? NMFMA / B_IT : 1;
#pragma unroll
      for (int g = 0; g < B_IT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);     // PER MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // then one VMEM read
      }
    }
    if (tap == ntap - 1 && kc + 1 < nkc) {     // chunk boundary: every tap of this chunk has read the halo tile
      __syncthreads();
      store_a(ra, kc + 1);
    }
    __syncthreads();
  };
  const long tk_begin = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
  load_a(0, ra0);
  load_b(0, 0);
  if (!CONV && niter > 1) load_a(1, ra1);
  tk_prev = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
  const long tk_prologue = tk_prev - tk_begin;
  if (DBW && !stamp && !p.dbg) {
    store_a(ra0, 0);
    store_b(0);
    if (niter > 1) load_b(0, 1);
    __syncthreads();
    for (int it = 0; it < niter; ++it) iter_dbw(it, ra0);
  } else {
    for (int it = 0; it < niter; it += 2) {
      iter(it, ra0);
      if (it + 1 < niter) iter(it + 1, CONV ? ra0 : ra1);
    }
  }
  if (stamp) {       // cycles per phase, summed over the K loop, of wave 0 of two blocks -> C[0..15] (output is lost)
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3) && blockIdx.y == 0 && tid == 0) {
      float* o = p.C + (blockIdx.x == 0 ? 0 : 8);
      o[0] = (float)tk_prologue; o[1] = (float)tk_b1; o[2] = (float)tk_store; o[3] = (float)tk_b2;
      o[4] = (float)tk_load; o[5] = (float)tk_mma; o[6] = (float)niter; o[7] = 0.f;
    }
    return;
  }

  if (p.dbg & 8) return;
  if constexpr (!CONV && WM == 1) {
    if (p.epi == 5) {
      nt_epilogue_lnbwd<WN>(p, acc, lane, wm, wn, m0, nvalid, (float*)smem);
      return;
    }
  }
  if constexpr (!CONV && WM == 1) {
    if (p.wide_epi) {                          // block-uniform (set by the dispatcher)
      nt_epilogue_wide<WN>(p, acc, lane, wave, wm, wn, n0, nvalid, m0, (float*)smem);
      return;
    }
  }
  nt_epilogue<WM, WN, CONV>(p, acc, lane, wm, wn, n0, nvalid, m0, img, y0, x0);
  if constexpr (!CONV && WM == 1) {
    if (p.stats_out) nt_row_stats<WN>(p, acc, lane, wm, wn, m0, nvalid, (float*)smem);
  }
}

template <int WM, int WN, bool CONV, bool AMP = false>
int launch_ntb(const NtArgs& p, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int AROWS = CONV ? (BM / 16 + 2) * 18 : BM;
  constexpr int LDS = 3 * (AROWS + BN * ((CONV && WN == 1) ? 2 : 1)) * PITCH;     // DBW: two W buffers
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_ntb<WM, WN, CONV, AMP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return sr_fail(-5, "k_ntb: cannot reserve %d B of LDS: %s", LDS, hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid;
  if (CONV) grid = dim3(p.tiles_x * p.tiles_y * p.batch, sr_cdiv(p.N, p.n_tile));
  else grid = dim3(sr_cdiv(p.M, BM), sr_cdiv(p.N, p.n_tile));
  hipLaunchKernelGGL((k_ntb<WM, WN, CONV, AMP>), grid, dim3(256), LDS, st, p);
  SR_LAUNCH_CHECK("k_ntb");
  return 0;
}

int ntb_env(const char* name, int dflt) {
  const char* e = sr_getenv(name);
  return e ? atoi(e) : dflt;
}

template <bool CONV>
int dispatch_ntb(NtArgs& p, hipStream_t st) {
  int wn;
  if (p.N % 180 == 0) { p.n_tile = 180; wn = 3; }
  else if (p.N <= 64) { p.n_tile = 64; wn = 1; }
  else if (p.N <= 128 || p.N % 128 == 0) { p.n_tile = 128; wn = 2; }
  else { p.n_tile = 192; wn = 3; }
  long blocks128;
  if (CONV) blocks128 = (long)sr_cdiv(p.Wd, 16) * sr_cdiv(p.H, 8) * p.batch;
  else blocks128 = sr_cdiv(p.M, 128);
  blocks128 *= sr_cdiv(p.N, p.n_tile);
  int wm = blocks128 >= 1024 ? 2 : 1;
  if (!CONV && wn == 3) wm = 1;          // <2,3> GEMM tile spills
  if (!CONV && p.stats_out) wm = 1;      // row statistics: WM = 1 epilogue
  const int force_wm = ntb_env("SRHIP_NTB_WM", 0);
  if (force_wm == 1 || force_wm == 2) wm = force_wm;
  if (CONV) {
    p.tiles_x = sr_cdiv(p.Wd, 16);
    p.tiles_y = sr_cdiv(p.H, wm == 2 ? 8 : 4);
    p.xcd_order = ntb_env("SRHIP_CONV_XCD", 1);
  }
  if constexpr (CONV) {       // weight planes of format 1 (two fp16 planes, prep kind 4): k_nhcw2 (64-column tiles / slices) or k_nhcw
    if (p.wfmt == 1) {          // 64-column slices (column block fastest) for wider outputs, as k_ntcw2
      SR_REQUIRE((p.N <= 4096 || p.N % 180 == 0) && p.K <= 4096, "conv3x3_f16x2: Cout <= 4096 or a multiple of 180, Cin <= 4096 (Cout=%d Cin=%d)", p.N, p.K);
      if (wn == 3 && p.N % 64 != 0 && p.ps == 0) {     // 180 / 192-column tiles (SwinIR): 64-pixel tiles whatever the image size
        p.tiles_y = sr_cdiv(p.H, 4);
        return sr_conv3x3_nhcw(p, st);
      }
      p.n_tile = 64;
      if (wm == 1 && wn != 1) p.tiles_y = sr_cdiv(p.H, 4);
      return sr_conv3x3_nhcw2(p, wm == 2 ? 4 : 2, st);
    }
  }
  if constexpr (CONV) {       // 64-pixel x 192-column tiles: W fragments straight from global memory (gemm_ntw.hip); SRHIP_NTCW=0: k_ntb<1, 3>
    if (wm == 1 && wn == 3 && p.ps == 0 && !p.dbg && !p.stagger && ntb_env("SRHIP_NTCW", 1)) return sr_conv3x3_ntcw(p, st);
    if (wm == 2 && (wn == 1 || (wn == 2 && ntb_env("SRHIP_NTCW2_WIDE", 1))) && !p.dbg && !p.stagger && ntb_env("SRHIP_NTCW2", 1)) {
      p.n_tile = 64;          // wider outputs (64 -> 256 of the upsampler): 64-column slices, column block fastest
      return sr_conv3x3_ntcw2(p, 4, st);
    }
    if (wm == 1 && wn == 1 && !p.dbg && !p.stagger && ntb_env("SRHIP_NTCW2_SMALL", 1)) return sr_conv3x3_ntcw2(p, 2, st);
  }
  if constexpr (CONV) {       // reduced-precision inference (srhip_set_matmul_mode(1)): conv kernels only here,
    if (p.amp) {              // GEMMs take gemm_ntp.hip's AMP instantiation
#define SR_NTB_AMP(WM_, WN_) \
  if (wm == WM_ && wn == WN_) return launch_ntb<WM_, WN_, true, true>(p, st);
      SR_NTB_AMP(1, 1) SR_NTB_AMP(1, 2) SR_NTB_AMP(1, 3)
      SR_NTB_AMP(2, 1) SR_NTB_AMP(2, 2) SR_NTB_AMP(2, 3)
#undef SR_NTB_AMP
    }
  }
#define SR_NTB_CASE(WM_, WN_) \
  if (wm == WM_ && wn == WN_) return launch_ntb<WM_, WN_, CONV>(p, st);
  SR_NTB_CASE(1, 1) SR_NTB_CASE(1, 2) SR_NTB_CASE(1, 3)
  SR_NTB_CASE(2, 1) SR_NTB_CASE(2, 2) SR_NTB_CASE(2, 3)
#undef SR_NTB_CASE
  return sr_fail(-22, "ntb: no kernel for wm=%d wn=%d", wm, wn);
}

// W[rows][ldw] f32 -> out[3][Kp/16][rows][16] bf16 (Kp = K rounded up to 32, zero filled)
__device__ __forceinline__ void split3_slot(const float* __restrict__ W, long ldw, int rows, int K, int Kp,
                                            unsigned short* __restrict__ out, long i) {
  const int kq = Kp >> 2;                                       // one float4 (4 k) per thread
  if (i >= (long)rows * kq) return;
  const int row = (int)(i / kq), k = (int)(i - (long)row * kq) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < K) v = ldg_f4(W + (long)row * ldw + k);      // K % 4 == 0
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(v.x, v.y, h0, m0, l0);
  split3_pair(v.z, v.w, h1, m1, l1);
  const long plane = (long)rows * Kp;
  unsigned short* d = out + ((long)(k >> 4) * rows + row) * 16 + (k & 15);      // 16-k sub-chunk major
  *(u32x2*)(d) = u32x2{h0, h1};
  *(u32x2*)(d + plane) = u32x2{m0, m1};
  *(u32x2*)(d + 2 * plane) = u32x2{l0, l1};
}

__global__ void k_split3(const float* __restrict__ W, long ldw, int rows, int K, int Kp,
                         unsigned short* __restrict__ out) {
  split3_slot(W, ldw, rows, K, Kp, out, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

}  // namespace

int sr_split3(const float* W, long ldw, int rows, int K, unsigned short* out, hipStream_t st) {
  SR_REQUIRE(K % 4 == 0 && ldw % 4 == 0 && rows > 0, "split_bf16x3: K, ldw multiples of 4 (K=%d)", K);
  const int Kp = sr_kp(K);
  const long n = (long)rows * (Kp / 4);
  hipLaunchKernelGGL(k_split3, dim3(sr_cdiv(n, 256)), dim3(256), 0, st, W, ldw, rows, K, Kp, out);
  SR_LAUNCH_CHECK("k_split3");
  return 0;
}

int sr_gemm_ntb(NtArgs& p, hipStream_t st) {
  SR_REQUIRE(!p.stats_out || p.N <= 192, "gemm_nt_bx3: row statistics need the row (N=%d) in one 192-column block", p.N);
  p.dbg = ntb_env("SRHIP_NT_DBG", 0);      // ablation bits, 0 in production
  p.stagger = ntb_env("SRHIP_NTB_STAGGER", 0);
  SR_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0, "gemm_nt_bx3: K, lda must be multiples of 4 (K=%d)", p.K);
  SR_REQUIRE(p.M > 0 && p.N > 0, "gemm_nt_bx3: empty problem");
  p.Kp = sr_kp(p.K);
  SR_REQUIRE((long)p.M * p.lda < (1L << 29) && 6L * p.N * p.Kp < (1L << 31),
             "gemm_nt_bx3: operand larger than 2 GiB (32-bit staging offsets)");
  // 16-byte epilogue accesses need 4-float alignment of every matrix it touches
  const auto al4 = [](const void* q, long ld) { return !q || (((size_t)q & 15) == 0 && ld % 4 == 0); };
  p.wide_epi = p.N % 4 == 0 && al4(p.C, p.ldc) && al4(p.R, p.ldr) && al4(p.aux, p.ldaux) &&
               ((size_t)p.C & 15) == 0 && ntb_env("SRHIP_NTB_WIDE", 1);
  // K <= 192 at 180-column widths (the K = 180 Linears of a Swin block): weights resident in registers,
  p.amp = sr_matmul_mode();
  if (p.wfmt == 1) return sr_gemm_ntp(p, st);          // two-plane fp16 operand: k_nth (gemm_ntw.hip), whatever the mode
  if (p.amp && p.epi != 5) return sr_gemm_ntp(p, st);
  // 64-row tiles (every case but very tall problems with narrow N, which take the 128-row
  // tiles of this file): the 16-wide-stage kernel of gemm_ntp.hip
  {
    const bool wide = p.N % 180 == 0 || p.N > 128;
    const long blocks128 = (long)sr_cdiv(p.M, 128) * sr_cdiv(p.N, wide ? 192 : (p.N <= 64 ? 64 : 128));
    if ((wide || p.stats_out || blocks128 < 1024) && ntb_env("SRHIP_NTP", 1)) return sr_gemm_ntp(p, st);
  }
  return dispatch_ntb<false>(p, st);
}

// out = res + LayerNorm-backward(A . W^T) in one kernel (epilogue 5)
int sr_gemm_ntb_lnbwd(NtArgs& p, hipStream_t st) {
  SR_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0, "gemm_nt_bx3_lnbwd: K, lda must be multiples of 4 (K=%d)", p.K);
  SR_REQUIRE(p.M > 0 && p.N > 0 && p.N <= 192, "gemm_nt_bx3_lnbwd: the row (N=%d) must fit one 192-column block", p.N);
  SR_REQUIRE(p.R && p.ep_stats, "gemm_nt_bx3_lnbwd: x and its statistics are required");
  p.Kp = sr_kp(p.K);
  p.epi = 5;
  p.dbg = 0; p.stagger = 0;
  if (ntb_env("SRHIP_NTP", 1) || p.wfmt == 1) return sr_gemm_ntp(p, st);
  if (p.N % 180 == 0 || p.N > 128) { p.n_tile = (p.N % 180 == 0) ? 180 : 192; return launch_ntb<1, 3, false>(p, st); }
  if (p.N > 64) { p.n_tile = 128; return launch_ntb<1, 2, false>(p, st); }
  p.n_tile = 64;
  return launch_ntb<1, 1, false>(p, st);
}

int sr_conv3x3_ntb(NtArgs& p, hipStream_t st) {
  SR_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0, "conv3x3_bx3: Cin, lda must be multiples of 4 (Cin=%d)", p.K);
  SR_REQUIRE(p.batch > 0 && p.H > 0 && p.Wd > 0, "conv3x3_bx3: empty image");
  p.M = p.batch * p.H * p.Wd;
  p.Kp = sr_kp(p.K);
  p.dbg = ntb_env("SRHIP_NT_DBG", 0);      // ablation / stamp bits, 0 in production
  p.amp = sr_matmul_mode();
  SR_REQUIRE((long)p.M * p.lda * (p.ps == 2 ? 4 : 1) < (1L << 29) && 54L * p.N * p.Kp < (1L << 31),
             "conv3x3_bx3: operand larger than 2 GiB (32-bit staging offsets)");
  SR_REQUIRE(p.ps == 0 || (p.ps == 1 && p.N % 4 == 0 && !p.R) || (p.ps == 2 && p.K % 128 == 0),
             "conv3x3_bx3 + PixelShuffle(2): Cout %% 4 == 0 and no residual operand (store side), Cin/4 a multiple of 32 (load side)");
  return dispatch_ntb<true>(p, st);
}
