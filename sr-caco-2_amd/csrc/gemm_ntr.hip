// NT GEMM on the 3-way bf16 split MFMA with the WEIGHT OPERAND RESIDENT IN REGISTERS
// (K <= 192: the K = 180 Linears of a Swin block -- qkv, proj, fc1 forward; proj and fc2
// data gradients):
//
//   C[M,N] = epi( pro(A)[M,K] . W[N,K]^T )      W pre-split into bf16 planes (prep.hip)
//
// Why: the K loop of gemm_ntp.hip is bound by the LDS pipe, not by the matrix core -- per
// 16-wide stage a block moves 18 KB of W through VGPR -> LDS stores (13 cycles per
// ds_write_b128) and reads it back, ~90 % LDS-busy at the MFMA bound,
// and every launch pays prologue / W stream / epilogue for 64 rows in lockstep.  Here
//   * a block is PERSISTENT (one per CU, 4 waves = one per SIMD, so each wave may use the SIMD's
//     whole 512-entry register file) and walks 64-row tiles of A;
//   * wave ns holds W[48 columns][192 k] as MFMA B-fragments in 216 registers for the whole
//     launch -- loaded once, straight from the global planes, never through LDS (an 8-wave
//     form with the k range split over wave pairs needed 256 registers per wave and spilled 99);
//   * only A passes through LDS: the tile is split ONCE into three bf16 planes (64 rows x
//     192 k: 73.7 KB), double buffered, so the global loads of tile t+1 are in flight during the
//     MFMAs of tile t; LDS traffic per MFMA drops ~4x.
// v_mfma_f32_16x16x32_bf16 (A: lane (c, g) holds row c, k = 8g..8g+7; D: col = c, rows 4g..4g+3;
// cdna_hip_programming.md section 3): a wave owns 4 row tiles x 3 column tiles = 48
// accumulator registers, 432 MFMAs per tile.
//
// LDS image of A: 16-byte units (8 bf16 of one row), unit (row, u) at slot
//   u*64 + (row & 48) + ((row & 15) ^ ((u & 3) | 12*((u >> 2) & 1)))
// -- the 16 lanes of every ds_read_b128 group (all 16 rows of a row tile, g from a pair {g, g^1})
// and the 16 lanes of every ds_write_b64 group (8 consecutive units of one row x 2 halves)
// hit 16 different 16-byte bank groups.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int RM = 64;                       // rows per tile
constexpr int KP = 192;                      // k held in registers (6 MFMA k steps of 32)
constexpr int NU = KP / 8;                   // 16-byte units per row and plane
constexpr int PLANE_B = NU * RM * 16;        // 24576
constexpr int ABUF = 3 * PLANE_B;            // 73728
constexpr int F4_ROW = KP / 4;               // 48 float4 per staged row
constexpr int NTHR = 256;
constexpr int A_IT = RM * F4_ROW / NTHR;     // 12 float4 per thread
constexpr int LDS_BYTES = 2 * ABUF + 2 * 4 * RM * 4;

__device__ __forceinline__ int a_slot(int row, int u) {
  return u * 64 + (row & 48) + ((row & 15) ^ ((u & 3) | (((u >> 2) & 1) * 12)));
}
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                 0, 0, 0);
}
// sum over the 16 lanes that share g (= one output row)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct AStage {
  f32x4 v[A_IT];
  float2 st[A_IT];
};

__global__ void __launch_bounds__(NTHR, 1) k_ntr(NtArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const red = (float*)(smem + 2 * ABUF);            // [2][4][64] row-statistics exchange
  const int tid = threadIdx.x, lane = tid & 63;
  const int ns = __builtin_amdgcn_readfirstlane(tid >> 6); // column slice (48 columns) of this wave
  const int c = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * p.n_tile;
  const int nvalid = min(p.n_tile, p.N - n0);
  const int ntiles = (p.M + RM - 1) / RM;

  // ---- W fragments: [column tile][k step][plane], once per launch.  Planes are [Kp/16][N][16]
  //      bf16: lane (c, g) of k step ks reads the 16 bytes (g & 1) of row gn in sub-chunk
  //      2*ks + (g >> 1); columns past the block's width re-read the last valid row (never
  //      stored), sub-chunks past Kp are zero.
  u32x4 wf[3][6][3];
  {
    const long plane_bytes = (long)p.N * p.Kp * 2;
    const int nsub = p.Kp / 16;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int gn = n0 + min(ns * 48 + nt * 16 + c, nvalid - 1);
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const int sub = 2 * ks + (g >> 1);
        const bool ok = sub < nsub;
        const char* base = (const char*)p.Wb + ((long)min(sub, nsub - 1) * p.N + gn) * 32 + (g & 1) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const u32x4 w = *(const u32x4*)(base + pl * plane_bytes);
          wf[nt][ks][pl] = ok ? w : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  }

  // ---- A staging: thread = A_IT float4 of the 64 x 192 tile, consecutive threads on consecutive
  //      float4 of a row (coalesced); rows / k past the problem are clamped on load, zeroed on store
  auto load_a = [&](int m0, AStage& s) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int f = tid + i * NTHR;
      const int row = f / F4_ROW, k4 = f - row * F4_ROW;
      const int gm = min(m0 + row, p.M - 1), kk = min(4 * k4, p.K - 4);
      s.v[i] = *(const f32x4*)(p.A + (long)gm * p.lda + kk);
      s.st[i] = ldg_f2(p.a_mode == 1 ? p.ln_stats + 2 * (long)gm : k_sr_neutral);
    }
  };
  auto store_a = [&](unsigned char* buf, int m0, const AStage& s) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int f = tid + i * NTHR;
      const int row = f / F4_ROW, k4 = f - row * F4_ROW;
      f32x4 v = s.v[i];
      if (p.a_mode == 1) {
        v = (v - s.st[i].x) * s.st[i].y;
      } else if (p.a_mode == 2) {
        v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
      }
      if (4 * k4 >= p.K || m0 + row >= p.M) v = f32x4{0.f, 0.f, 0.f, 0.f};     // exact zeros
      unsigned h0, m0_, l0, h1, m1, l1;
      split3_pair(v.x, v.y, h0, m0_, l0);
      split3_pair(v.z, v.w, h1, m1, l1);
      unsigned char* dst = buf + a_slot(row, k4 >> 1) * 16 + (k4 & 1) * 8;
      *(u32x2*)(dst) = u32x2{h0, h1};
      *(u32x2*)(dst + PLANE_B) = u32x2{m0_, m1};
      *(u32x2*)(dst + 2 * PLANE_B) = u32x2{l0, l1};
    }
  };

  int t = blockIdx.x;
  AStage sa;
  load_a(t * RM, sa);                                       // grid.x <= ntiles
  store_a(smem, t * RM, sa);
  __syncthreads();

  for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
    const unsigned char* const cur = smem + (it & 1) * ABUF;
    unsigned char* const nxt = smem + ((it + 1) & 1) * ABUF;
    const int m0 = t * RM;
    const int tn = t + gridDim.x;
    const bool more = tn < ntiles;                          // block-uniform
    if (more && !(p.dbg & 2)) load_a(tn * RM, sa);          // in flight during the MFMA phase

    f32x4 acc[4][3];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 24 steps (k step, row tile), fully unrolled (the W fragments are register arrays); the A
    // fragments of step s+1 are read while the 18 MFMAs of step s run.  The scheduling barriers keep
    // the compiler from hoisting ALL 72 fragment reads (288 registers) to the top of the phase.
    u32x4 fa[3];
    {
      const unsigned char* ap = cur + a_slot(c, g) * 16;
      fa[0] = *(const u32x4*)(ap); fa[1] = *(const u32x4*)(ap + PLANE_B); fa[2] = *(const u32x4*)(ap + 2 * PLANE_B);
    }
    if (!(p.dbg & 4))
#pragma unroll
    for (int sidx = 0; sidx < 24; ++sidx) {
      const int ks = sidx >> 2, rt = sidx & 3;
      u32x4 fn[3] = {fa[0], fa[1], fa[2]};
      if (sidx + 1 < 24) {
        const int ks2 = (sidx + 1) >> 2, rt2 = (sidx + 1) & 3;
        const unsigned char* ap = cur + a_slot(rt2 * 16 + c, 4 * ks2 + g) * 16;
        fn[0] = *(const u32x4*)(ap); fn[1] = *(const u32x4*)(ap + PLANE_B); fn[2] = *(const u32x4*)(ap + 2 * PLANE_B);
      }
      // the six cross products >= 2^-24, small terms first (as gemm_ntp.hip)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[1], wf[nt][ks][1], acc[rt][nt]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[0], wf[nt][ks][2], acc[rt][nt]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[2], wf[nt][ks][0], acc[rt][nt]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[0], wf[nt][ks][1], acc[rt][nt]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[1], wf[nt][ks][0], acc[rt][nt]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[rt][nt] = mfma16(fa[0], wf[nt][ks][0], acc[rt][nt]);
      __builtin_amdgcn_sched_barrier(0);
      fa[0] = fn[0]; fa[1] = fn[1]; fa[2] = fn[2];
    }
    // tile t+1 goes into the other buffer (its last readers finished before the barrier that
    // ended the previous iteration); the barrier below publishes it and retires `cur`
    if (more && !(p.dbg & 2)) store_a(nxt, tn * RM, sa);
    if (p.dbg & 1) {            // timing experiment: no epilogue traffic (results are lost)
      float sacc = 0.f;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) sacc += acc[rt][nt][0] + acc[rt][nt][1] + acc[rt][nt][2] + acc[rt][nt][3];
      if (sacc == 123456.789f) p.C[0] = sacc;
      __syncthreads();
      continue;
    }

    // ---- epilogue on the wave's 64 rows x 48 columns: value (rt, nt, i) is row 16rt + 4g + i,
    //      column ns*48 + nt*16 + c
    float blk_s = p.alpha;
    if (p.rowscale) blk_s *= p.rowscale[m0 / p.rows_per_scale];   // rows_per_scale % 64 == 0 (dispatcher)
    const bool needR = p.R != nullptr && p.epi >= 2;
    int gn[3];
    bool cok[3];
    float bv[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int col = ns * 48 + nt * 16 + c;
      cok[nt] = col < nvalid;
      gn[nt] = n0 + min(col, nvalid - 1);
      bv[nt] = (cok[nt] && p.bias) ? p.bias[gn[nt]] : 0.f;
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 rv[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = min(m0 + 16 * rt + 4 * g + i, p.M - 1);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) rv[nt][i] = needR ? p.R[(long)gr * p.ldr + gn[nt]] : 0.f;
      }
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        f32x4 v = acc[rt][nt] + bv[nt];
        switch (p.epi) {
          case 1:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            break;
          case 2:
            v = v * blk_s + rv[nt];
            break;
          case 3:
#pragma unroll
            for (int i = 0; i < 4; ++i) {                   // Phi / phi as nt_epilogue (backward-grade erf)
              const float x = rv[nt][i];
              const float z = fabsf(x) * 0.70710678118654752440f;
              const float e1 = __expf(-0.5f * x * x);
              const float tt = __frcp_rn(1.0f + 0.3275911f * z);
              const float poly = tt * (0.254829592f + tt * (-0.284496736f + tt * (1.421413741f +
                                 tt * (-1.453152027f + tt * 1.061405429f))));
              const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e1, x));
              v[i] = v[i] * blk_s * (cdf + x * 0.39894228040143267794f * e1);
              rv[nt][i] = x * cdf;                          // gelu(R): second output
            }
            break;
          case 4:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rv[nt][i] > 0.f ? v[i] : 0.f;
            break;
          default:
            break;
        }
        acc[rt][nt] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = m0 + 16 * rt + 4 * g + i;
        if (gr < p.M) {
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
            if (cok[nt]) {
              p.C[(long)gr * p.ldc + gn[nt]] = acc[rt][nt][i];
              if (p.epi == 3 && p.aux) p.aux[(long)gr * p.ldaux + gn[nt]] = rv[nt][i];
            }
        }
      }
    }

    if (p.stats_out) {
      // LayerNorm statistics of the OUTPUT rows (one N block holds the row): two-pass like the
      // reference -- mean, then squared deviations; 16-lane sums, then the 4 column slices in LDS
      const float inv = 1.0f / (float)p.N;
      float mean[4][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s1 = 0.f;
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) s1 += cok[nt] ? acc[rt][nt][i] : 0.f;
          s1 = row16_sum(s1);
          if (c == 0) red[ns * RM + 16 * rt + 4 * g + i] = s1;
        }
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lr = 16 * rt + 4 * g + i;
          mean[rt][i] = ((red[lr] + red[RM + lr]) + (red[2 * RM + lr] + red[3 * RM + lr])) * inv;
          float s2 = 0.f;
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) {
            const float d = cok[nt] ? acc[rt][nt][i] - mean[rt][i] : 0.f;
            s2 += d * d;
          }
          s2 = row16_sum(s2);
          if (c == 0) red[4 * RM + ns * RM + lr] = s2;
        }
      __syncthreads();
      if (ns == 0 && c == 0) {
        const float* r2 = red + 4 * RM;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int lr = 16 * rt + 4 * g + i;
            const float var = ((r2[lr] + r2[RM + lr]) + (r2[2 * RM + lr] + r2[3 * RM + lr])) * inv;
            if (m0 + lr < p.M) *(float2*)(p.stats_out + 2 * (long)(m0 + lr)) = float2{mean[rt][i], rsqrtf(var + 1e-5f)};
          }
      }
    }
    __syncthreads();           // `nxt` is complete, every wave is done with `cur` (and with `red`)
  }
}

}  // namespace
// Can this problem run on the register-resident-W kernel?  (checked by the dispatcher in gemm_ntb.hip)
bool sr_gemm_ntr_ok(const NtArgs& p) {
  // OFF by default: measured on MI355X (tools/mb_ntr.py, T = 32768) the launch is
  // 1.5-1.8x SLOWER than gemm_ntp.hip -- with one wave per SIMD nothing overlaps the wave's own
  // non-MFMA phases (ablation, us per 64-row tile: MFMA 4.5, A staging 4.4, epilogue 8, + 14 us of
  // prologue per launch for the 221 KB of W fragments every CU pulls at the same moment).  Read per
  // call so that tests can switch it on.
  const char* e = getenv("SRHIP_NTR");
  if (!(e && atoi(e) == 1)) return false;
  if (p.K > KP || p.K <= 96 || p.Kp > KP) return false;          // one 192-k pass; narrower K wastes the registers
  if (!(p.N % 180 == 0 || p.N <= 192)) return false;
  if (p.epi < 0 || p.epi > 4 || p.a_mode < 0 || p.a_mode > 2) return false;
  if (p.rowscale && p.rows_per_scale % RM != 0) return false;
  if (p.stats_out && p.N > 192) return false;
  if (p.M < 64 * RM) return false;                               // persistent blocks want many tiles
  return true;
}

int sr_gemm_ntr(NtArgs& p, hipStream_t st) {
  p.n_tile = (p.N % 180 == 0) ? 180 : 192;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_ntr, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return sr_fail(-5, "k_ntr: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e));
    attr = true;
  }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  // one resident block per CU in total: the CUs are shared out among the N blocks, every block keeps
  // its W slice for all the row tiles it walks (N = 540: 85 blocks x 6 tiles per column block)
  const int ntiles = sr_cdiv(p.M, RM), ny = sr_cdiv(p.N, p.n_tile);
  int gx = ncu / ny;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  dim3 grid(gx, ny);
  hipLaunchKernelGGL(k_ntr, grid, dim3(NTHR), LDS_BYTES, st, p);
  SR_LAUNCH_CHECK("k_ntr");
  return 0;
}
