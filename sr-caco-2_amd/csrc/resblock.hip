// One EDSR ResBlock per launch: conv3x3 -> ReLU -> conv3x3 -> (x res_scale) + skip, and its data gradient, as ONE kernel
// each (reference ResBlock.forward, dlib/models/network_nlsn.py:72-93, and its autograd):
//
//   forward :  a   = relu(conv(x; W1) + b1)                    out  = x + rs * (conv(a; W2) + b2)
//   backward:  da  = rs * conv(g; W2^T) * (a > 0)              dx   = g + conv(da; W1^T)
//
// Both have the same shape -- stage 1: a 3x3 conv of the block's input over the output tile PLUS a one-pixel ring, an
// elementwise epilogue, its result written out (the weight gradients read a / da) and kept on the CU as the operand of
// stage 2: a 3x3 conv over the output tile, + the block's input.  At the x8 patch size (8 x 64 x 64 pixels, 64 channels) a
// body conv of EDSR is 512 blocks of 1.6 us of matrix work inside a 16-us launch: the time is ramp, halo fetch, weight
// fragments arriving from the L2 and drain, twice per ResBlock and direction.  Here the second conv starts from LDS the
// moment the first has finished: one ramp, one drain, one halo fetch (two pixels wide) and no read-back of `a` per block.
//
// Tile: 4 x 16 output pixels per block (512 blocks at 8 x 64 x 64, two per CU), mid region 6 x 18 = 108 pixels (seven
// 16-pixel MFMA row tiles of FLATTENED mid pixels: a lane's LDS address is its own, so the 18-wide rows cost nothing),
// input halo 8 x 20.  Arithmetic as k_nhcw2 (gemm_ntw.hip): two fp16 planes per operand, three products on
// v_mfma_f32_16x16x32_f16, f32 accumulate; the input tile under ONE running power-of-two exponent per halo tile (channel
// chunks of 32), the mid tile under one exponent of its own (its maximum is known before it is split); weight planes of
// preparation kind 4 ([plane][Kp/16][9 N][16] + per-column 2^-s), fragments straight from global memory, three register
// sets ahead.  Mid pixels outside the IMAGE are zeros (the second conv pads the ACTIVATION with zeros), not conv values.
#include "common.h"
#include "kernels.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sr_f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int RB_C = 64;                       // channels in and out
constexpr int RB_TH = 4, RB_TW = 16;           // output tile
constexpr int RB_MW = RB_TW + 2, RB_MH = RB_TH + 2, RB_MPX = RB_MW * RB_MH;      // mid region 6 x 18 = 108
constexpr int RB_MT = (RB_MPX + 15) / 16;      // 7 row tiles of 16 flattened mid pixels (112 slots)
constexpr int RB_IW = RB_TW + 4, RB_IH = RB_TH + 4, RB_IPX = RB_IW * RB_IH;      // input halo 8 x 20 = 160
constexpr int RB_PITCH = 80;                   // bytes per pixel and plane of a 32-channel chunk image (64 + 16 pad)
constexpr int RB_INPLANE = RB_IPX * RB_PITCH;  // 12800
constexpr int RB_MPLANE = RB_MT * 16 * RB_PITCH;   // 8960
constexpr int RB_TP = 68;                      // pitch (floats) of the row-major f32 tiles of the two epilogues
constexpr int RB_REGION_A = (RB_MT * 16 * RB_TP * 4 > 2 * RB_INPLANE) ? RB_MT * 16 * RB_TP * 4 : 2 * RB_INPLANE;   // 30464
constexpr int RB_REGION_M = 2 * 2 * RB_MPLANE; // two chunks x two planes = 35840
constexpr int RB_LDS = RB_REGION_A + RB_REGION_M + 64;
constexpr int RB_INIT = (RB_IPX * 8 + 255) / 256;  // float4 slots of a chunk per thread (5)
constexpr int RB_E1IT = (RB_MT * 16 * 8 + 255) / 256;   // (pixel, 8 columns) pieces of the mid tile per thread (4; 3.5 used)

__device__ __forceinline__ f32x4 mfma16h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
  const sr_f16x2 hv = __builtin_convertvector(sr_f32x2{x0, x1}, sr_f16x2);
  const float r0 = x0 - (float)hv.x, r1 = x1 - (float)hv.y;
  const sr_f16x2 lv = __builtin_convertvector(sr_f32x2{r0, r1}, sr_f16x2);
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, lv);
}

template <bool BWD>
__global__ void __launch_bounds__(256, 2) k_resblock64(ResBlockArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const regA = smem;
  unsigned char* const regM = smem + RB_REGION_A;
  float* const red = (float*)(smem + RB_REGION_A + RB_REGION_M);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int t = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * RB_TH, x0 = tx * RB_TW;
  const long imgpix = (long)img * p.H * p.Wd;

  // ---------------- weight fragment addressing (both convs: N = K = 64, Kp = 64)
  constexpr long wrows = 9L * RB_C;
  constexpr long plane_bytes = wrows * RB_C * 2;
  unsigned boff[2];
  float winv1[2], winv2[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = wn * 32 + jt * 16 + c;
    boff[jt] = (unsigned)(((g >> 1) * wrows + col) * 32 + (g & 1) * 16);
    winv1[jt] = ((const float*)((const char*)p.W1 + 2 * plane_bytes))[col];
    winv2[jt] = ((const float*)((const char*)p.W2 + 2 * plane_bytes))[col];
  }
  // every block walks the nine taps from another start (k_nhcw2: the blocks of a launch would otherwise ask the L2 for the
  // same weight lines at the same moment)
  const int rot9 = p.k_rot ? (int)(((unsigned)blockIdx.x >> 3) % 9u) : 0;
  auto tr9 = [&](int tap) { const int x = tap + rot9; return x >= 9 ? x - 9 : x; };
  auto load_b = [&](const void* Wb, int it, u32x4 (&fb)[2][2]) {
    const int kc = it / 9, tap = tr9(it - kc * 9);
    const char* base = (const char*)Wb + ((long)(2 * kc) * wrows + (long)tap * RB_C) * 32;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[jt][pl] = *(const u32x4*)(base + pl * plane_bytes + boff[jt]);
  };

  // ---------------- stage 1 input: the 8 x 20 halo tile of x (forward) / g (backward), 32 channels at a time
  unsigned offA[RB_INIT];
  bool inA[RB_INIT];
#pragma unroll
  for (int it = 0; it < RB_INIT; ++it) {
    const int idx = tid + it * 256;
    const int row = idx >> 3, c4 = idx & 7;
    const int hy = row / RB_IW, hx = row - hy * RB_IW;
    const int y = y0 + hy - 2, x = x0 + hx - 2;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    offA[it] = (unsigned)(((imgpix + (long)yc * p.Wd + xc) * p.ldx + c4 * 4) * 4);
  }
  f32x4 ra[RB_INIT];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int it = 0; it < RB_INIT; ++it) ra[it] = *(const f32x4*)((const char*)p.X + offA[it] + kc * 128);
  };
  auto store_a = [&](float use) {
#pragma unroll
    for (int it = 0; it < RB_INIT; ++it) {
      const int idx = tid + it * 256;
      const f32x4 v = ra[it];
      unsigned h0, l0, h1, l1;
      split2_pair(v.x * use, v.y * use, h0, l0);
      split2_pair(v.z * use, v.w * use, h1, l1);
      unsigned char* dst = regA + (idx >> 3) * RB_PITCH + (idx & 7) * 8;
      *(u32x2*)(dst) = u32x2{h0, h1};
      *(u32x2*)(dst + RB_INPLANE) = u32x2{l0, l1};
    }
  };

  u32x4 fbr[3][2][2];
  load_a(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) load_b(p.W1, q, fbr[q]);

  // this wave's row tiles of the mid region: mt = wm + 2 i (i = 0 .. 3; wm = 1 has three)
  f32x4 acc1[4][2];
  int a_off1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m = min(16 * (wm + 2 * i) + c, RB_MPX - 1);
    const int my = m / RB_MW, mx = m - my * RB_MW;
    a_off1[i] = (my * RB_IW + mx) * RB_PITCH + 16 * g;
  }
  const int nt1 = wm == 0 ? 4 : 3;
  auto mma1 = [&](int tap, const u32x4 (&fb)[2][2]) {
    const int tp = tr9(tap), ty3 = (tp * 11) >> 5;
    const int toff = (ty3 * RB_IW + (tp - 3 * ty3)) * RB_PITCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nt1) {
        const u32x4 fh = *(const u32x4*)(regA + a_off1[i] + toff), fl = *(const u32x4*)(regA + RB_INPLANE + a_off1[i] + toff);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16h(fh, fb[j][1], acc1[i][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16h(fl, fb[j][0], acc1[i][j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = mfma16h(fh, fb[j][0], acc1[i][j]);
      }
    }
  };

  float cur = 3.0e38f;
#pragma unroll 1
  for (int kc = 0; kc < 2; ++kc) {
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < RB_INIT; ++it) {
      f32x4 v = ra[it];
      if (!inA[it]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      ra[it] = v;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_max(mx);
    if (kc) __syncthreads();                  // every tap of the previous chunk has read the halo planes (and `red`)
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float need = mx > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mx)), 100.f)) : 3.0e38f;
    const float old = cur;
    cur = fminf(cur, need);
    store_a(cur > 1.0e38f ? 1.f : cur);
    __syncthreads();
    if (kc == 0) load_a(1);
    if (kc && old != cur && old < 1.0e38f) {  // the tile's exponent dropped: bring the sums along (exact power of two)
      const float f = cur / old;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc1[i][j][e] *= f;
    }
    const int it0 = kc * 9;
#pragma unroll 1
    for (int t3 = 0; t3 < 9; t3 += 3) {
      mma1(t3, fbr[0]);     if (it0 + t3 + 3 < 18) load_b(p.W1, it0 + t3 + 3, fbr[0]);
      mma1(t3 + 1, fbr[1]); if (it0 + t3 + 4 < 18) load_b(p.W1, it0 + t3 + 4, fbr[1]);
      mma1(t3 + 2, fbr[2]); if (it0 + t3 + 5 < 18) load_b(p.W1, it0 + t3 + 5, fbr[2]);
    }
  }
  // the second conv's first weight fragments travel through the epilogue of the first
#pragma unroll
  for (int q = 0; q < 3; ++q) load_b(p.W2, q, fbr[q]);
  const float tinv1 = 1.0f / (cur > 1.0e38f ? 1.f : cur);

  // ---------------- epilogue 1: the mid tile row-major in LDS, then (pixel, 8 columns) pieces per thread
  __syncthreads();                             // the halo planes are dead
  float* const T = (float*)regA;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nt1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          T[(16 * (wm + 2 * i) + 4 * g + e) * RB_TP + 32 * wn + 16 * j + c] = acc1[i][j][e] * (tinv1 * winv1[j]);
    }
  __syncthreads();
  float mv[RB_E1IT][8];
  float mmax = 0.f;
#pragma unroll
  for (int it = 0; it < RB_E1IT; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = (idx & 7) * 8;
    const int my = px / RB_MW, mx_ = px - my * RB_MW;
    const int y = y0 + my - 1, x = x0 + mx_ - 1;
    const bool ok = px < RB_MPX && y >= 0 && y < p.H && x >= 0 && x < p.Wd;
    const f32x4 v0 = *(const f32x4*)(T + px * RB_TP + c8), v1 = *(const f32x4*)(T + px * RB_TP + c8 + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const long gpx = imgpix + (long)min(max(y, 0), p.H - 1) * p.Wd + min(max(x, 0), p.Wd - 1);
    if (!BWD) {
      const f32x4 b0 = ldg_f4(p.b1 + c8), b1 = ldg_f4(p.b1 + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = fmaxf(v[e] + b0[e], 0.f); v[4 + e] = fmaxf(v[4 + e] + b1[e], 0.f); }
    } else {
      const f32x4 a0 = ldg_f4(p.Mask + gpx * p.ldmask + c8), a1 = ldg_f4(p.Mask + gpx * p.ldmask + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = a0[e] > 0.f ? v[e] * p.rs : 0.f;
        v[4 + e] = a1[e] > 0.f ? v[4 + e] * p.rs : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = ok ? v[e] : 0.f;                 // outside the image: the zero padding of the second conv's input
      mv[it][e] = v[e];
      mmax = fmaxf(mmax, fabsf(v[e]));
    }
    if (ok && my >= 1 && my <= RB_TH && mx_ >= 1 && mx_ <= RB_TW) {       // this block's own pixels: a / da leave for HBM
      float* dst = p.Mid + gpx * p.ldmid + c8;
      *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  mmax = wave_max(mmax);
  if (lane == 0) red[4 + wave] = mmax;
  __syncthreads();                             // the maxima; every thread has read its pieces of T
  mmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float sc2 = mmax > 0.f ? exp2f(fminf(floorf(log2f(16384.f / mmax)), 100.f)) : 1.f;
#pragma unroll
  for (int it = 0; it < RB_E1IT; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = (idx & 7) * 8;
    if (px < RB_MT * 16) {
      unsigned h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split2_pair(mv[it][2 * q] * sc2, mv[it][2 * q + 1] * sc2, h[q], l[q]);
      unsigned char* dst = regM + (c8 >> 5) * (2 * RB_MPLANE) + px * RB_PITCH + (c8 & 31) * 2;
      *(u32x4*)(dst) = u32x4{h[0], h[1], h[2], h[3]};
      *(u32x4*)(dst + RB_MPLANE) = u32x4{l[0], l[1], l[2], l[3]};
    }
  }
  __syncthreads();

  // ---------------- stage 2: 4 x 16 output pixels from the mid planes
  f32x4 acc2[2][2];
  int a_off2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    a_off2[i] = ((2 * wm + i) * RB_MW + c) * RB_PITCH + 16 * g;
  }
  auto mma2 = [&](int kc, int tap, const u32x4 (&fb)[2][2]) {
    const int tp = tr9(tap), ty3 = (tp * 11) >> 5;
    const int toff = kc * (2 * RB_MPLANE) + (ty3 * RB_MW + (tp - 3 * ty3)) * RB_PITCH;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32x4 fh = *(const u32x4*)(regM + a_off2[i] + toff), fl = *(const u32x4*)(regM + RB_MPLANE + a_off2[i] + toff);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = mfma16h(fh, fb[j][1], acc2[i][j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = mfma16h(fl, fb[j][0], acc2[i][j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = mfma16h(fh, fb[j][0], acc2[i][j]);
    }
  };
#pragma unroll 1
  for (int it0 = 0; it0 < 18; it0 += 3) {
    const int kc = it0 / 9, t3 = it0 - 9 * kc;
    mma2(kc, t3, fbr[0]);     if (it0 + 3 < 18) load_b(p.W2, it0 + 3, fbr[0]);
    mma2(kc, t3 + 1, fbr[1]); if (it0 + 4 < 18) load_b(p.W2, it0 + 4, fbr[1]);
    mma2(kc, t3 + 2, fbr[2]); if (it0 + 5 < 18) load_b(p.W2, it0 + 5, fbr[2]);
  }

  // ---------------- epilogue 2: + bias, x res_scale (forward), + the block's input
  const float tinv2 = 1.0f / sc2;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        T[(16 * (2 * wm + i) + 4 * g + e) * RB_TP + 32 * wn + 16 * j + c] = acc2[i][j][e] * (tinv2 * winv2[j]);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < (RB_TH * RB_TW * 8) / 256; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = (idx & 7) * 8;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    if (y >= p.H || x >= p.Wd) continue;
    const long gpx = imgpix + (long)y * p.Wd + x;
    const f32x4 v0 = *(const f32x4*)(T + px * RB_TP + c8), v1 = *(const f32x4*)(T + px * RB_TP + c8 + 4);
    const f32x4 r0 = ldg_f4(p.X + gpx * p.ldx + c8), r1 = ldg_f4(p.X + gpx * p.ldx + c8 + 4);
    f32x4 o0, o1;
    if (!BWD) {
      const f32x4 b0 = ldg_f4(p.b2 + c8), b1 = ldg_f4(p.b2 + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = (v0[e] + b0[e]) * p.rs + r0[e]; o1[e] = (v1[e] + b1[e]) * p.rs + r1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = v0[e] + r0[e]; o1[e] = v1[e] + r1[e]; }
    }
    float* dst = p.Out + gpx * p.ldout + c8;
    *(f32x4*)dst = o0;
    *(f32x4*)(dst + 4) = o1;
  }
}

}  // namespace

int sr_resblock64(ResBlockArgs& p, int bwd, hipStream_t st) {
  p.tiles_x = sr_cdiv(p.Wd, RB_TW);
  p.tiles_y = sr_cdiv(p.H, RB_TH);
  { static const int crot = [] { const char* e = sr_getenv("SRHIP_CONV_ROT"); return e ? atoi(e) : 1; }(); p.k_rot = crot; }
  const dim3 grid(p.tiles_x * p.tiles_y * p.batch);
  static bool reserved[2] = {false, false};
  if (!reserved[bwd ? 1 : 0]) {
    const hipError_t e = bwd ? hipFuncSetAttribute((const void*)k_resblock64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS)
                             : hipFuncSetAttribute((const void*)k_resblock64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS);
    if (e != hipSuccess) return sr_fail(-5, "resblock64: cannot reserve %d bytes of LDS", RB_LDS);
    reserved[bwd ? 1 : 0] = true;
  }
  if (bwd) hipLaunchKernelGGL(k_resblock64<true>, grid, dim3(256), RB_LDS, st, p);
  else hipLaunchKernelGGL(k_resblock64<false>, grid, dim3(256), RB_LDS, st, p);
  SR_LAUNCH_CHECK("k_resblock64");
  return 0;
}
