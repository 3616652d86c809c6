// fp16 STORAGE inference path of the plain conv family (VDSR, DRRN, EDSR; --amp at evaluation time): activations live in
// HBM as fp16 (channels-last), weights are the LEADING fp16 plane of the fp16x2 conv operand (srhip_prep_table job kind 4:
// [Kp/16][9*Cout][16] fp16 under one power-of-two scale per output channel, followed by the second plane and the Cout
// inverse scales), ONE v_mfma_f32_16x16x32_f16 product, f32 accumulate, fp16 out.  Half the bytes of the f32-storage --amp
// path on both sides of every conv and no split arithmetic in the staging: the halo tile goes from global memory into the
// stage image as it is.
//
//   k_conv3x3_h16   3x3 conv, stride 1, zero padding, Cin a multiple of 32, Cout a multiple of 64: 2 RW x 16 pixel tiles x
//                   64-column slices (the shapes and wave layout of k_nhcw2, gemm_ntw.hip); the nine taps of a 32-channel
//                   chunk run without a barrier on weight fragments requested a chunk ahead; optional BatchNorm-ReLU input
//                   prologue (MemNet); epilogues bias / ReLU / LeakyReLU / residual / residual + ReLU on the row-major
//                   re-laid tile (16-byte accesses), optionally stored through PixelShuffle(2) (the EDSR upsampler,
//                   network_nlsn.py:100-118)
//   k_conv1x1_h16   a 1x1 conv held as the centre tap of a 3x3 operand (MemNet's gate units, SRCNN's layers): no halo,
//                   three 32-channel chunks per stage
//   k_cin1_h16      the 1 -> Cout conv at the head (f32 image in, fp16 features out; network_vdsr.py:57-60)
//   k_cout1_h16     the Cin -> 1 conv at the tail (fp16 features in, f32 image out, + the interpolated input)
#include "common.h"
#include "kernels.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HP = 80;                           // bytes per halo pixel of a 32-channel chunk (64 + 16 pad: conflict-free ds_read_b128)
constexpr int TPF = 68;                          // pitch of the f32 output tile (floats)

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

constexpr int h16_lds(int rw) {       // the halo tile of a chunk | three chunks of the tile's own pixels (1x1) | the f32 output tile
  const int halo = (2 * rw + 2) * 18 * HP, one = 3 * 2 * rw * 16 * HP, tile = 2 * rw * 16 * TPF * 4;
  return (halo > tile ? halo : tile) > one ? (halo > tile ? halo : tile) : one;
}

// The accumulators of a 2 RW x 16 pixel x 64 column tile (acc[i][j][e] = pixel 16 (RW wm + i) + 4 g + e, column 32 wn + 16 j + c)
// -> row-major in LDS -> bias / activation / residual on 8-column groups -> fp16 stores of 16 bytes.
template <int RW>
__device__ __forceinline__ void h16_epilogue(const ConvH16Args& p, const f32x4 (&acc)[RW][2], unsigned char* smem, int tid, int wm,
                                             int wn, int c, int g, int n0, int img, int y0, int x0) {
  constexpr int NPX = 2 * RW * 16;
  __syncthreads();                                // the halo tile is dead from here on
  float* const T = (float*)smem;
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * (RW * wm + i) + 4 * g + e) * TPF + 32 * wn + 16 * j + c] = acc[i][j][e];
  __syncthreads();
  const float* const winv = (const float*)((const char*)p.Wb + 2 * p.plane_bytes);
  // thread -> (pixel, 8-column group): NPX * 8 items
#pragma unroll
  for (int it = 0; it < (NPX * 8) / 256; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = idx & 7;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    if (y >= p.H || x >= p.Wd) continue;
    const int col = n0 + c8 * 8;
    const f32x4 t0 = *(const f32x4*)(T + px * TPF + c8 * 8), t1 = *(const f32x4*)(T + px * TPF + c8 * 8 + 4);
    const f32x4 w0 = ldg_f4(winv + col), w1 = ldg_f4(winv + col + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = t0[e] * w0[e]; v[4 + e] = t1[e] * w1[e]; }
    if (p.bias) {
      // conv + PixelShuffle(2): kernel column sp * (N / 4) + cc holds torch channel cc * 4 + sp
      if (p.ps) {
        const int fs = p.N >> 2, sp = col / fs, cc = col - sp * fs;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[(cc + e) * 4 + sp];
      } else {
        const f32x4 b0 = ldg_f4(p.bias + col), b1 = ldg_f4(p.bias + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
    }
    const long pix = ((long)img * p.H + y) * p.Wd + x;
    if (p.epi == 2 || p.epi == 8) {
      const h16x8 r = *(const h16x8*)(p.R + pix * p.ldr + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + (float)r[e];
    }
    if (p.epi == 1 || p.epi == 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.epi == 6) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
    if (p.ps) {
      const int fs = p.N >> 2, sp = col / fs, cc = col - sp * fs;
      const long opix = ((long)img * 2 * p.H + 2 * y + (sp >> 1)) * (2 * p.Wd) + 2 * x + (sp & 1);
      *(h16x8*)(p.Y + opix * p.ldy + cc) = o;
    } else {
      *(h16x8*)(p.Y + pix * p.ldy + col) = o;
    }
  }
}

template <int RW>
__global__ void __launch_bounds__(256, 3) k_conv3x3_h16(ConvH16Args p) {
  constexpr int AROWS = (2 * RW + 2) * 18;       // halo pixels of a 2 RW x 16 tile
  constexpr int AN = AROWS * 4;                  // 16-byte slots per chunk (8 channels each)
  constexpr int AIT = (AN + 255) / 256;
  constexpr int NPX = 2 * RW * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int t = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int ncol = p.N >> 6;
  const int n0 = (t % ncol) * 64; t /= ncol;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = p.K >> 5;

  unsigned offA[AIT];
  bool inA[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int idx = min(tid + it * 256, AN - 1);
    const int row = idx >> 2, c8 = idx & 3;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd && (AN % 256 == 0 || tid + it * 256 < AN);
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    offA[it] = (unsigned)((((long)img * p.H + yc) * p.Wd + xc) * p.ldx + c8 * 8) * 2u;
  }
  // input prologue (MemNet's BN-ReLU-conv, network_memnet.py:27-34): relu((x - mean) k + beta) on the thread's eight
  // channels of the chunk (256 % 4 == 0: the same group in every iteration), before the padding zeros go in
  f32x4 bm[2], bk[2], bb[2];
  auto load_a = [&](int kc, u32x4 (&ra)[AIT]) {
    const char* base = (const char*)p.X + (long)kc * 64;
#pragma unroll
    for (int it = 0; it < AIT; ++it) ra[it] = inA[it] ? *(const u32x4*)(base + offA[it]) : u32x4{0u, 0u, 0u, 0u};
  };
  auto store_a = [&](const u32x4 (&ra)[AIT], int kc) {
    if (p.in_bn) {                                // (L1 / L2 hits: every block reads the same 3 x 32 floats per chunk)
      const int ch = kc * 32 + (tid & 3) * 8;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        bm[h2] = ldg_f4(p.in_bn + ch + 4 * h2);
        bk[h2] = ldg_f4(p.in_bn + 2 * p.K + ch + 4 * h2);
        bb[h2] = ldg_f4(p.in_bn + 3 * p.K + ch + 4 * h2);
      }
    }
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      if (AN % 256 == 0 || tid + it * 256 < AN) {
        const int idx = tid + it * 256;
        u32x4 v = ra[it];
        if (p.in_bn) {
          h16x8 hv = __builtin_bit_cast(h16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = (_Float16)fmaxf(((float)hv[e] - bm[e >> 2][e & 3]) * bk[e >> 2][e & 3] + bb[e >> 2][e & 3], 0.f);
          v = __builtin_bit_cast(u32x4, hv);
        }
        *(u32x4*)(smem + (idx >> 2) * HP + (idx & 3) * 16) = inA[it] ? v : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  const long wrows = 9L * p.N;
  unsigned boff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) boff[jt] = (unsigned)(((g >> 1) * wrows + n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
  auto load_b = [&](int kc, int tap, u32x4 (&fb)[2]) {
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) fb[jt] = *(const u32x4*)(base + boff[jt]);
  };
  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 18 + c) * HP + 16 * g;

  u32x4 ra[AIT];
  load_a(0, ra);
  // weight fragments three taps ahead, in three register sets (a whole chunk ahead took 72 registers: two blocks per CU;
  // with 24 the kernel fits three)
  u32x4 fb0[2], fb1[2], fb2[2];
  const int niter = nkc * 9;
  auto load_it = [&](int it, u32x4 (&fb)[2]) { load_b(it / 9, it % 9, fb); };
  auto mma = [&](int tap, const u32x4 (&fb)[2]) {
    const int toff = ((tap / 3) * 18 + (tap % 3)) * HP;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const u32x4 fa = *(const u32x4*)(smem + a_off[i] + toff);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa, fb[j], acc[i][j]);
    }
  };
  load_it(0, fb0); load_it(1, fb1); load_it(2, fb2);
  for (int kc = 0; kc < nkc; ++kc) {
    if (kc) __syncthreads();                      // every tap of the previous chunk has read the halo tile
    store_a(ra, kc);
    __syncthreads();
    if (kc + 1 < nkc) load_a(kc + 1, ra);
    const int it = kc * 9;
#pragma unroll
    for (int t3 = 0; t3 < 9; t3 += 3) {
      mma(t3, fb0);     if (it + t3 + 3 < niter) load_it(it + t3 + 3, fb0);
      mma(t3 + 1, fb1); if (it + t3 + 4 < niter) load_it(it + t3 + 4, fb1);
      mma(t3 + 2, fb2); if (it + t3 + 5 < niter) load_it(it + t3 + 5, fb2);
    }
  }

  h16_epilogue<RW>(p, acc, smem, tid, wm, wn, c, g, n0, img, y0, x0);
}

// 1x1 conv (the weight's centre tap): no halo, so a stage is THREE 32-channel chunks of the tile's own pixels (one pair of
// barriers per 96 channels instead of per 32: at one tap a chunk is 8 MFMAs per wave, the barriers were the loop).
template <int RW>
__global__ void __launch_bounds__(256, 2) k_conv1x1_h16(ConvH16Args p) {
  constexpr int NPX = 2 * RW * 16;
  constexpr int CPS = 3;                         // chunks per stage
  constexpr int AN = NPX * 4;                    // 16-byte slots per chunk
  constexpr int AIT = AN / 256;                  // 2 (RW = 4) or 1 (RW = 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int t = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int ncol = p.N >> 6;
  const int n0 = (t % ncol) * 64; t /= ncol;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = p.K >> 5;
  unsigned offA[AIT];
  bool inA[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 2, c8 = idx & 3;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    inA[it] = y < p.H && x < p.Wd;
    offA[it] = (unsigned)((((long)img * p.H + min(y, p.H - 1)) * p.Wd + min(x, p.Wd - 1)) * p.ldx + c8 * 8) * 2u;
  }
  f32x4 bm[CPS][2], bk[CPS][2], bb[CPS][2];
  auto load_a = [&](int kc0, u32x4 (&ra)[CPS][AIT]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      const int kc = min(kc0 + q, nkc - 1);
      const char* base = (const char*)p.X + (long)kc * 64;
#pragma unroll
      for (int it = 0; it < AIT; ++it) ra[q][it] = inA[it] ? *(const u32x4*)(base + offA[it]) : u32x4{0u, 0u, 0u, 0u};
      if (p.in_bn) {
        const int ch = kc * 32 + (tid & 3) * 8;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          bm[q][h2] = ldg_f4(p.in_bn + ch + 4 * h2);
          bk[q][h2] = ldg_f4(p.in_bn + 2 * p.K + ch + 4 * h2);
          bb[q][h2] = ldg_f4(p.in_bn + 3 * p.K + ch + 4 * h2);
        }
      }
    }
  };
  auto store_a = [&](const u32x4 (&ra)[CPS][AIT]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q)
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int idx = tid + it * 256;
        u32x4 v = ra[q][it];
        if (p.in_bn) {
          h16x8 hv = __builtin_bit_cast(h16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = (_Float16)fmaxf(((float)hv[e] - bm[q][e >> 2][e & 3]) * bk[q][e >> 2][e & 3] + bb[q][e >> 2][e & 3], 0.f);
          v = inA[it] ? __builtin_bit_cast(u32x4, hv) : u32x4{0u, 0u, 0u, 0u};
        }
        *(u32x4*)(smem + q * NPX * HP + (idx >> 2) * HP + (idx & 3) * 16) = v;
      }
  };
  const long wrows = 9L * p.N;
  unsigned boff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) boff[jt] = (unsigned)(((g >> 1) * wrows + 4L * p.N + n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
  auto load_b = [&](int kc0, u32x4 (&fb)[CPS][2]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      const char* base = (const char*)p.Wb + (long)(2 * min(kc0 + q, nkc - 1)) * wrows * 32;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[q][jt] = *(const u32x4*)(base + boff[jt]);
    }
  };
  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 16 + c) * HP + 16 * g;
  u32x4 ra[CPS][AIT];
  u32x4 fb[CPS][2];
  load_a(0, ra);
  load_b(0, fb);
  for (int kc0 = 0; kc0 < nkc; kc0 += CPS) {
    if (kc0) __syncthreads();
    store_a(ra);
    __syncthreads();
    if (kc0 + CPS < nkc) load_a(kc0 + CPS, ra);
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      if (kc0 + q < nkc) {                          // block-uniform
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const u32x4 fa = *(const u32x4*)(smem + q * NPX * HP + a_off[i]);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa, fb[q][j], acc[i][j]);
        }
      }
    }
    if (kc0 + CPS < nkc) load_b(kc0 + CPS, fb);
  }
  h16_epilogue<RW>(p, acc, smem, tid, wm, wn, c, g, n0, img, y0, x0);
}

// 1 -> Co conv (f32 image in, fp16 features out), act 0 none | 1 ReLU | 2 LeakyReLU(alpha).  A thread owns 8 output channels of one pixel.
__global__ void __launch_bounds__(256) k_cin1_h16(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, _Float16* __restrict__ y, long ldy, int B,
                                                  int H, int W, int Co, int act, float alpha) {
  extern __shared__ float wl[];                  // [9][Co] + [Co]
  for (int i = threadIdx.x; i < 9 * Co; i += 256) wl[(i % 9) * Co + i / 9] = w[i];
  for (int i = threadIdx.x; i < Co; i += 256) wl[9 * Co + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int G = Co >> 3;
  const long n = (long)B * H * W * G;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int gq = (int)(i % G);
    const long pix = i / G;
    const int xx = (int)(pix % W);
    const long r = pix / W;
    const int yy = (int)(r % H);
    const long b = r / H;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wl[9 * Co + gq * 8 + e];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      const float xv = x[(b * H + sy) * W + sx];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += xv * wl[t * Co + gq * 8 + e];
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(act == 1 ? fmaxf(v[e], 0.f) : (act == 2 && v[e] < 0.f ? v[e] * alpha : v[e]));
    *(h16x8*)(y + pix * ldy + gq * 8) = o;
  }
}

// Ci -> 1 conv (fp16 features in, f32 image out) + bias + optional f32 addend image.  Eight lanes per pixel, each over an
// eighth of the channels (16-byte loads), summed with three shuffles.
__global__ void __launch_bounds__(256) k_cout1_h16(const _Float16* __restrict__ x, long ldx, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ add,
                                                   const float* __restrict__ in_bn, float* __restrict__ y, int B, int H, int W,
                                                   int Ci) {
  extern __shared__ float wl[];                  // [9][Ci] weights, [3][Ci] BatchNorm mean / k / beta of the input prologue
  float* const cf = wl + 9 * Ci;
  for (int i = threadIdx.x; i < 9 * Ci; i += 256) wl[(i % 9) * Ci + i / 9] = w[i];
  if (in_bn)
    for (int i = threadIdx.x; i < Ci; i += 256) { cf[i] = in_bn[i]; cf[Ci + i] = in_bn[2 * Ci + i]; cf[2 * Ci + i] = in_bn[3 * Ci + i]; }
  __syncthreads();
  const long n = (long)B * H * W;
  const int sub = threadIdx.x & 7;
  for (long pix = blockIdx.x * 32L + (threadIdx.x >> 3); pix < n + 31; pix += (long)gridDim.x * 32) {   // whole groups of 8 lanes stay together
    const long pc = min(pix, n - 1);
    const int xx = (int)(pc % W);
    const long r = pc / W;
    const int yy = (int)(r % H);
    const long b = r / H;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      const _Float16* row = x + ((b * H + sy) * W + sx) * ldx;
      for (int c0 = sub * 8; c0 < Ci; c0 += 64) {
        const h16x8 xv = *(const h16x8*)(row + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xe = (float)xv[e];
          if (in_bn) xe = fmaxf((xe - cf[c0 + e]) * cf[Ci + c0 + e] + cf[2 * Ci + c0 + e], 0.f);
          s += xe * wl[t * Ci + c0 + e];
        }
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0 && pix < n) y[pix] = s + (bias ? bias[0] : 0.f) + (add ? add[pix] : 0.f);
  }
}

}  // namespace

int sr_conv3x3_h16(ConvH16Args& p, hipStream_t st) {
  SR_REQUIRE(p.X && p.Wb && p.Y, "conv3x3_h16: null operand");
  SR_REQUIRE(p.K % 32 == 0 && p.K >= 32 && p.K <= 4096 && p.N % 64 == 0 && p.N <= 4096,
             "conv3x3_h16: Cin = %d (a multiple of 32), Cout = %d (a multiple of 64), both <= 4096", p.K, p.N);
  SR_REQUIRE(p.ldx % 8 == 0 && p.ldy % 8 == 0 && (!p.R || p.ldr % 8 == 0), "conv3x3_h16: pixel pitches must be multiples of 8 halves");
  SR_REQUIRE(p.epi == 0 || p.epi == 1 || p.epi == 6 || ((p.epi == 2 || p.epi == 8) && p.R),
             "conv3x3_h16: epilogue %d (0, 1, 6, 2 / 8 with R)", p.epi);
  SR_REQUIRE(!p.ps || (p.N % 256 == 0 && !p.R), "conv3x3_h16 + PixelShuffle(2): Cout %% 256 == 0 and no residual (Cout=%d)", p.N);
  SR_REQUIRE(p.B > 0 && p.H > 0 && p.Wd > 0, "conv3x3_h16: empty image");
  SR_REQUIRE((long)p.B * p.H * p.Wd * p.ldx < (1L << 31), "conv3x3_h16: input larger than 4 GiB (32-bit staging offsets)");
  p.Kp = (p.K + 31) / 32 * 32;
  p.plane_bytes = 9L * p.N * p.Kp * 2;
  const long blocks128 = (long)sr_cdiv(p.Wd, 16) * sr_cdiv(p.H, 8) * p.B * (p.N / 64);
  const int rw = blocks128 >= 1024 ? 4 : 2;
  p.tiles_x = sr_cdiv(p.Wd, 16);
  p.tiles_y = sr_cdiv(p.H, 2 * rw);
  dim3 grid((unsigned)((long)p.tiles_x * p.tiles_y * p.B * (p.N / 64)));
  if (p.center_only) {
    if (rw == 4) hipLaunchKernelGGL(k_conv1x1_h16<4>, grid, dim3(256), h16_lds(4), st, p);
    else hipLaunchKernelGGL(k_conv1x1_h16<2>, grid, dim3(256), h16_lds(2), st, p);
  } else if (rw == 4) hipLaunchKernelGGL(k_conv3x3_h16<4>, grid, dim3(256), h16_lds(4), st, p);
  else hipLaunchKernelGGL(k_conv3x3_h16<2>, grid, dim3(256), h16_lds(2), st, p);
  SR_LAUNCH_CHECK("k_conv3x3_h16");
  return 0;
}

int sr_conv_cin1_h16(const float* x, const float* w, const float* bias, void* y, long ldy, int B, int H, int W, int Co, int act,
                     float alpha, hipStream_t st) {
  SR_REQUIRE(x && w && y, "conv_cin1_h16: null operand");
  SR_REQUIRE(Co % 8 == 0 && Co <= 1024 && ldy % 8 == 0, "conv_cin1_h16: Cout = %d (a multiple of 8, <= 1024)", Co);
  const long n = (long)B * H * W * (Co / 8);
  if (n <= 0) return 0;
  const int grid = (int)(n / 256 + 1 < 8192 ? n / 256 + 1 : 8192);
  hipLaunchKernelGGL(k_cin1_h16, dim3(grid), dim3(256), (size_t)10 * Co * 4, st, x, w, bias, (_Float16*)y, ldy, B, H, W, Co, act, alpha);
  SR_LAUNCH_CHECK("k_cin1_h16");
  return 0;
}

int sr_conv_cout1_h16(const void* x, long ldx, const float* w, const float* bias, const float* add, const float* in_bn, float* y,
                      int B, int H, int W, int Ci, hipStream_t st) {
  SR_REQUIRE(x && w && y, "conv_cout1_h16: null operand");
  SR_REQUIRE(Ci % 64 == 0 && Ci <= 1024 && ldx % 8 == 0, "conv_cout1_h16: Cin = %d (a multiple of 64, <= 1024)", Ci);
  const long n = (long)B * H * W;
  if (n <= 0) return 0;
  const int grid = (int)(n / 32 + 1 < 16384 ? n / 32 + 1 : 16384);
  hipLaunchKernelGGL(k_cout1_h16, dim3(grid), dim3(256), (size_t)12 * Ci * 4, st, (const _Float16*)x, ldx, w, bias, add, in_bn, y, B, H, W, Ci);
  SR_LAUNCH_CHECK("k_cout1_h16");
  return 0;
}
