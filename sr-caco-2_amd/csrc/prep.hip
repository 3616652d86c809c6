// Per-step weight preparation for a whole network in ONE launch.
//
// After every optimizer step the matmul operands derived from the parameters
// must be rebuilt: bf16x3 planes of every Linear / conv weight (plain, transposed
// for the data gradient, LayerNorm gamma folded in; conv weights tap-major, the
// data-gradient twin with flipped taps), the LayerNorm-folded biases and the
// dense relative-position bias images.  Done weight by weight that is ~500 tiny
// launches per step for SwinIR -- CPU-launch bound.  Here the host builds a job
// table once (pointers are stable), and one kernel walks it: block -> job by
// binary search over the running block count.
//
// Replaces the per-weight srhip_fold_layernorm / srhip_transpose /
// srhip_pack_conv_weight / srhip_bias_expand / srhip_split_bf16x3 sequence
// (same results; those entry points remain for single weights).
#include "common.h"
#include "kernels.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

// Index maps of the fused MLP kernel (mlp_fused.hip), hs = hidden / 2, 96 hidden units per wave and round:
//   perm 1 (rows): plane row  round*192 + half*96 + i            <- hidden unit half*hs + round*96 + i
//   perm 2 (k)   : plane k    round*192 + t*16 + 8h + 4a + e     <- hidden unit (t&1)*hs + round*96 + 16*(t>>1) + 8a + 4h + e
//                  (t = stage of the round: its parity is the hidden half; inside a 16-group the order of the
//                  accumulator registers of the 32x32 MFMA tile that produced the activation)
// -1 = padding (zero)
__device__ __forceinline__ int mlp_row_src(int r, int hs) {
  const int c = r / 192, rem = r - c * 192, half = rem / 96, i = rem - half * 96;
  const int off = c * 96 + i;
  return off < hs ? half * hs + off : -1;
}
__device__ __forceinline__ int mlp_k_src(int k, int hs) {
  const int c = k / 192, rem = k - c * 192, t = rem >> 4, pos = rem & 15;
  const int off = c * 96 + 16 * (t >> 1) + 8 * ((pos >> 2) & 1) + 4 * (pos >> 3) + (pos & 3);
  return off < hs ? (t & 1) * hs + off : -1;
}

// kind 0: out planes [3][Kp/16][ntap*rows][16] (sub-chunk major, see gemm_ntb.hip) of  v(tap, r, k) = W[off + tap*s_tap + r*s_row + k*s_k]
//         * (gamma_mode = mode & 3: 1: gamma[k] | 2: gamma[r] | 0: 1);  mode >> 2: 1 = r through mlp_row_src, 2 = k through
//         mlp_k_src (hs in s0; one tap); 3 / 4 = r / k in sub-pixel-major order of a conv + PixelShuffle(2)
__device__ __forceinline__ void job_planes(const PrepEntry& e, int lb) {
  const int Kp = sr_kp(e.n2), kq = Kp >> 2;
  const long rows = (long)e.n0 * e.n1;
  const long i = (long)lb * 256 + threadIdx.x;
  if (i >= rows * kq) return;
  const int row = (int)(i / kq), k0 = (int)(i - (long)row * kq) * 4;
  const int perm = e.mode >> 2, gm = e.mode & 3;
  int tap = row / e.n0, r = row - tap * e.n0;
  long tapoff = (long)tap * e.s0;
  if (perm == 1 || perm == 2) { tap = 0; tapoff = 0; }
  if (perm == 1) r = mlp_row_src(r, e.s0);
  // conv + PixelShuffle(2) (NtArgs.ps): plane row / k index sp*(n/4) + c holds torch channel c*4 + sp
  if (perm == 3) { const int fs = e.n0 >> 2; r = (r % fs) * 4 + r / fs; }
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int k = k0 + j;
    float x = 0.f;
    if (k < e.n2 && r >= 0) {
      if (perm == 2) k = mlp_k_src(k, e.s0);
      if (perm == 4) { const int fs = e.n2 >> 2; k = (k % fs) * 4 + k / fs; }
      if (k >= 0) {
        x = ldg_f(e.a + (long)e.off + tapoff + (long)r * e.s1 + (long)k * e.s2);
        if (gm == 1) x *= ldg_f(e.b + k);
        else if (gm == 2) x *= ldg_f(e.b + r);
      }
    }
    v[j] = x;
  }
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(v[0], v[1], h0, m0, l0);
  split3_pair(v[2], v[3], h1, m1, l1);
  const long plane = rows * Kp;
  unsigned short* d = (unsigned short*)e.out + ((long)(k0 >> 4) * rows + row) * 16 + (k0 & 15);   // 16-k sub-chunk major
  *(u32x2*)(d) = u32x2{h0, h1};
  *(u32x2*)(d + plane) = u32x2{m0, m1};
  *(u32x2*)(d + 2 * plane) = u32x2{l0, l1};
}

// kind 1: out[n] = bias[n] + sum_k W[n][k] * beta[k]   (LayerNorm shift folded into the Linear bias)
__device__ __forceinline__ void job_fold_bias(const PrepEntry& e, int lb) {
  const int n = lb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= e.n0) return;
  float a = 0.f;
  for (int k = lane; k < e.n1; k += 64) a += ldg_f(e.a + (long)n * e.n1 + k) * ldg_f(e.c + k);
  a = wave_sum(a);
  if (lane == 0) ((float*)e.out)[n] = (e.b ? ldg_f(e.b + n) : 0.f) + a;
}

// kind 2: lane-ordered relative-position bias images of one block (k_bias_expand / wa_img_index, wattn.hip)
__device__ __forceinline__ void job_bias_expand(const PrepEntry& e, int lb) {
  const int i = lb * 256 + threadIdx.x, heads = e.n0;
  if (i >= heads * 4096) return;
  const int hd = i / 4096, el = i & 4095;
  const int q = el & 15, lane = (el >> 4) & 63, b = (el >> 10) & 1, a = el >> 11;
  const int row = mfma_row(q, lane), col = lane & 31;
  auto rpi = [](int query, int key) { return ((query >> 3) - (key >> 3) + 7) * 15 + ((query & 7) - (key & 7) + 7); };
  ((float*)e.out)[i] = ldg_f(e.a + rpi(col + 32 * b, row + 32 * a) * heads + hd);    // imgT
  ((float*)e.out2)[i] = ldg_f(e.a + rpi(row + 32 * a, col + 32 * b) * heads + hd);   // imgN
  if (e.c) {     // third image: the S^T accumulator order of the fp16x2 attention kernels (w2_img_index, wattn2.hip)
    const int ee = el & 3, ln = (el >> 2) & 63, J = (el >> 8) & 3, I = el >> 10;
    ((float*)e.c)[i] = ldg_f(e.a + rpi(16 * I + (ln & 15), 16 * J + 4 * (ln >> 4) + ee) * heads + hd);
    // fourth image (`b`): the key side of the backward, img[J][I][lane][e] = bias(query 16 I + 4 g + e, key 16 J + c)
    if (e.b) ((float*)e.b)[i] = ldg_f(e.a + rpi(16 * J + 4 * (ln >> 4) + ee, 16 * I + (ln & 15)) * heads + hd);
  }
}

// kind 3 (experiment, SRHIP_F16X2=1): TWO fp16 planes with a power-of-two scale per ROW instead of three bf16 planes --
//   v' = v * 2^s(row), v' = h + l, h = fp16(v'), l = fp16(v' - h);  out planes [2][Kp/16][rows][16] fp16 (same sub-chunk
//   major layout, plane stride rows*Kp halves), then rows floats 2^-s(row) behind the two planes.  One wave per row:
//   the row maximum decides s (max * 2^s in [8192, 16384]).  Same value definition as kind 0 (gamma modes; no perms).
__device__ __forceinline__ void job_planes_f16(const PrepEntry& e, int lb) {
  const int Kp = sr_kp(e.n2);
  const int rows = e.n0;
  const int row = lb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  // a transposed source (s2 != 1: the block's four rows are four CONSECUTIVE floats of every source line): the block
  // gathers its 4 x n2 values through LDS with 16 bytes of each line per 4 lanes, instead of 64 lines per load
  // instruction and wave (the texture path, one line per clock, was the bound: 39 us for SwinIR's 96 transposed jobs)
  __shared__ float tile[4][1024 + 4];
  const bool staged = e.s2 != 1 && e.s1 == 1;        // block-uniform
  if (staged) {
    const int r = threadIdx.x & 3, row_r = lb * 4 + r;
    if (row_r < rows)
      for (int k = threadIdx.x >> 2; k < e.n2; k += 64)
        tile[r][k] = ldg_f(e.a + (long)e.off + (long)row_r + (long)k * e.s2);
    __syncthreads();
  }
  if (row >= rows) return;
  const int gm = e.mode & 3;
  constexpr int MAXJ = 4;                            // 4 x 256 k per row
  float v[MAXJ][4];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = j * 256 + lane * 4 + q;
      float x = 0.f;
      if (k < e.n2) {
        x = staged ? tile[threadIdx.x >> 6][k] : ldg_f(e.a + (long)e.off + (long)row * e.s1 + (long)k * e.s2);
        if (gm == 1) x *= ldg_f(e.b + k);
        else if (gm == 2) x *= ldg_f(e.b + row);
      }
      v[j][q] = x;
      mx = fmaxf(mx, fabsf(x));
    }
  }
  mx = wave_max(mx);
  const float sc = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 1.f;
  const long plane = (long)rows * Kp;
  unsigned short* base = (unsigned short*)e.out;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k0 = j * 256 + lane * 4;
    if (k0 < Kp) {
      unsigned short hh[4], ll[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xs = v[j][q] * sc;
        const _Float16 h = (_Float16)xs;
        const _Float16 l = (_Float16)(xs - (float)h);
        hh[q] = __builtin_bit_cast(unsigned short, h);
        ll[q] = __builtin_bit_cast(unsigned short, l);
      }
      unsigned short* d = base + ((long)(k0 >> 4) * rows + row) * 16 + (k0 & 15);
      *(u32x2*)(d) = u32x2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16)};
      *(u32x2*)(d + plane) = u32x2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16)};
    }
  }
  if (lane == 0) ((float*)(base + 2 * plane))[row] = 1.0f / sc;
}

// kind 4 (experiment, SRHIP_F16X2_CONV=1): the tap-major conv pack as TWO fp16 planes with a power-of-two scale per OUTPUT
//   channel (one scale for the channel's nine tap rows: they accumulate into the same output column); planes
//   [2][Kp/16][9*n0][16] fp16, then n0 floats 2^-s(n).  One wave per output channel, 256 k per pass; K <= 4096; perms 3 / 4 (PixelShuffle orders) as kind 0.
__device__ __forceinline__ void job_conv_planes_f16(const PrepEntry& e, int lb) {
  const int Kp = sr_kp(e.n2);
  const int n = lb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= e.n0) return;
  // conv + PixelShuffle(2) (NtArgs.ps): plane row / k index sp*(n/4) + c holds torch channel c*4 + sp (perms 3 / 4 of kind 0)
  const int perm = e.mode >> 2;
  int rs = n;
  if (perm == 3) { const int fs = e.n0 >> 2; rs = (n % fs) * 4 + n / fs; }
  // 256 k per pass of the wave (K <= 256: one pass, the values stay in registers between the maximum and the split)
  const int nch = (Kp + 255) >> 8;
  float v[9][4];
  auto load_chunk = [&](int ch) -> float {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int k = ch * 256 + lane * 4 + q;
        float x = 0.f;
        if (k < e.n2) {
          if (perm == 4) { const int fs = e.n2 >> 2; k = (k % fs) * 4 + k / fs; }
          x = ldg_f(e.a + (long)e.off + (long)t * e.s0 + (long)rs * e.s1 + (long)k * e.s2);
        }
        v[t][q] = x;
        m = fmaxf(m, fabsf(x));
      }
    return m;
  };
  float mx = 0.f;
  for (int ch = 0; ch < nch; ++ch) mx = fmaxf(mx, load_chunk(ch));
  mx = wave_max(mx);
  const float sc = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 1.f;
  const long rows = 9L * e.n0, plane = rows * Kp;
  unsigned short* base = (unsigned short*)e.out;
  for (int ch = 0; ch < nch; ++ch) {
    if (nch > 1) load_chunk(ch);
    const int k0 = ch * 256 + lane * 4;
    if (k0 < Kp) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        unsigned short hh[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xs = v[t][q] * sc;
          const _Float16 h = (_Float16)xs;
          const _Float16 l = (_Float16)(xs - (float)h);
          hh[q] = __builtin_bit_cast(unsigned short, h);
          ll[q] = __builtin_bit_cast(unsigned short, l);
        }
        unsigned short* d = base + ((long)(k0 >> 4) * rows + (long)t * e.n0 + n) * 16 + (k0 & 15);
        *(u32x2*)(d) = u32x2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16)};
        *(u32x2*)(d + plane) = u32x2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16)};
      }
    }
  }
  if (lane == 0) ((float*)(base + 2 * plane))[n] = 1.0f / sc;
}

__global__ void __launch_bounds__(256) k_prep_table(const PrepEntry* __restrict__ tab, int n) {
  // block -> job: the number of jobs that start at or before this block (blk0 ascending), counted by the whole block with
  // independent loads -- ONE memory round trip where a binary search over the table chained log2(n) of them (the search
  // was most of a block's lifetime: 33 us for the 96 forward Linear jobs of SwinIR, whose data moves in 10)
  __shared__ int s_cnt[4];
  int cnt = 0;
  for (int i = threadIdx.x; i < n + (int)threadIdx.x; i += 256) {       // trip count uniform over the wave
    const bool in = i < n && tab[i].blk0 <= (int)blockIdx.x;
    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
  }
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const int lo = __builtin_amdgcn_readfirstlane((s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3])) - 1;
  const PrepEntry e = tab[lo];
  const int lb = blockIdx.x - e.blk0;
  if (e.kind == 0) job_planes(e, lb);
  else if (e.kind == 1) job_fold_bias(e, lb);
  else if (e.kind == 3) job_planes_f16(e, lb);
  else if (e.kind == 4) job_conv_planes_f16(e, lb);
  else job_bias_expand(e, lb);
}

}  // namespace

int sr_prep_blocks(const PrepEntry& e) {
  if (e.kind == 0) return sr_cdiv((long)e.n0 * e.n1 * (sr_kp(e.n2) / 4), 256);
  if (e.kind == 1) return sr_cdiv(e.n0, 4);
  if (e.kind == 2) return sr_cdiv((long)e.n0 * 4096, 256);
  if (e.kind == 3) return (e.n1 == 1 && e.n2 <= 1024 && (e.mode >> 2) == 0) ? sr_cdiv(e.n0, 4) : -1;
  if (e.kind == 4) return (e.n1 == 9 && e.n2 <= 4096 && (e.mode == 0 || e.mode == 12 || e.mode == 16)) ? sr_cdiv(e.n0, 4) : -1;
  return -1;
}

int sr_prep_table(const PrepEntry* tab_dev, int n, int total_blocks, hipStream_t st) {
  SR_REQUIRE(n > 0 && total_blocks > 0, "prep_table: empty table");
  hipLaunchKernelGGL(k_prep_table, dim3(total_blocks), dim3(256), 0, st, tab_dev, n);
  SR_LAUNCH_CHECK("k_prep_table");
  return 0;
}
