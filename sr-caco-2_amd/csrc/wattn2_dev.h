// Device-side pieces of the fp16x2 window attention shared by wattn2.hip (the attention core as its own launch) and
// wmsa_f16.hip (the whole W-MSA half of a Swin block as one launch): the row-form fragment helpers and the forward
// body of ONE (window, head) pair run by ONE wave.  Layouts and arithmetic: the header of wattn2.hip.
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

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
// Power-of-two block exponent of a group whose largest magnitude is mx (>= 0): 2^s with mx * 2^s in [8192, 16384),
// straight from the exponent field (s <= 100; an all-zero group gets 2^100: its products are exact zeros).  The
// exp2f(floorf(log2f(16384 / mx))) form of it was ~40 VALU instructions -- 46 of them per attention backward item, a
// third of the kernel's vector instructions.
__device__ __forceinline__ float pow2_scale(float mx) {
  const int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
  return __builtin_bit_cast(float, (unsigned)min(267 - e, 227) << 23);
}
// 2^-s of a power of two 2^s, exact
__device__ __forceinline__ float pow2_inv(float sc) {
  return __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, sc));
}

// 16-byte global accesses at 8-byte alignment (a head's slice of a row starts at 4 D bytes x head)
typedef f32x4 w3_f32x4a8 __attribute__((aligned(8)));
typedef const __attribute__((address_space(1))) w3_f32x4a8* w3_gp4;
// reductions over the four lanes {c, c + 16, c + 32, c + 48}: v_permlane16_swap leaves rows {0, 0, 2, 2} of the value in
// one register and rows {1, 1, 3, 3} in the other, v_permlane32_swap the lower half in one and the upper half in the other
// (inline asm: hipcc folds the two results of __builtin_amdgcn_permlane16_swap(u, u) into one value -- it drops the max
// and doubles the sum; s_nop 1 = the wait states it puts between a VALU write of the operands and the swap)
__device__ __forceinline__ void w3_swap16(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void w3_swap32(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float w3_max4(float v) {
  float a, b;
  w3_swap16(v, a, b); v = fmaxf(a, b);
  w3_swap32(v, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float w3_sum4(float v) {
  float a, b;
  w3_swap16(v, a, b); v = a + b;
  w3_swap32(v, a, b); return a + b;
}

// four consecutive head-dim entries d0 .. d0 + 3 of one row: 16 bytes where all four exist, 8 where two do
template <int D>
__device__ __forceinline__ void w3_store(float* p, int d0, const f32x4& v, float sc) {
  if (d0 + 4 <= D) *(w3_f32x4a8*)p = f32x4{v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
  else if (d0 + 2 <= D) *(float2*)p = float2{v[0] * sc, v[1] * sc};
}
struct W2Geom {
  int head, b, wy, wx;
  bool last_row, last_col;
};
__device__ __forceinline__ W2Geom w2_decode(long gid, int heads, int nWx, int nWy, int shift) {
  W2Geom g;
  g.head = (int)(gid % heads);
  long win = gid / heads;
  g.wx = (int)(win % nWx); win /= nWx;
  g.wy = (int)(win % nWy);
  g.b = (int)(win / nWy);
  g.last_row = shift > 0 && g.wy == nWy - 1;
  g.last_col = shift > 0 && g.wx == nWx - 1;
  return g;
}
// token index of window-local position pos (0..63) under the cyclic shift (network_swinir.py:297-301)
__device__ __forceinline__ int w2_token(const W2Geom& g, int pos, int H, int W, int shift) {
  int y = g.wy * 8 + (pos >> 3) + shift, x = g.wx * 8 + (pos & 7) + shift;
  if (y >= H) y -= H;
  if (x >= W) x -= W;
  return (g.b * H + y) * W + x;
}

// raw row-form fragment: the lane's 8 values of row `tok` (head-dim entries 8 g .. 8 g + 7; zeros past D)
template <int D>
__device__ __forceinline__ void w2_load_row(float (&v)[8], const float* __restrict__ base, long pitch, int tok, int g) {
  const float* p = base + (long)tok * pitch + 8 * g;
  // a pair past the head dim reads the lane's first pair instead and is zeroed afterwards: a load under a branch is
  // waited for at the branch's join (s_waitcnt vmcnt(0)), which serialises whatever is in flight -- the address select
  // keeps all four loads unconditional (D is even: a pair is valid or not as a whole)
  float2 x[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = ldg_f2(p + ((8 * g + 2 * t < D) ? 2 * t : 0));
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool ok = 8 * g + 2 * t < D;
    v[2 * t] = ok ? x[t].x : 0.f; v[2 * t + 1] = ok ? x[t].y : 0.f;
  }
}
// ... split under the row's block exponent (row maximum over the row's four lanes); returns 2^-s
__device__ __forceinline__ float w2_split_row(const float (&v)[8], u32x4& hi, u32x4& lo) {
  float mx = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) mx = fmaxf(mx, fabsf(v[t]));
  mx = w3_max4(mx);
  const float sc = pow2_scale(mx);
  unsigned h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) split2_pair(v[2 * t] * sc, v[2 * t + 1] * sc, h[t], l[t]);
  hi = u32x4{h[0], h[1], h[2], h[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
  return pow2_inv(sc);
}

// position (0..63) of slot t (0..7) of the k octet of lane group g in k step JJ (32 positions): the positions lane
// (c, g) holds in the S^T tiles J = 2 JJ and 2 JJ + 1
__device__ __forceinline__ int w2_kpos(int JJ, int g, int t) { return 32 * JJ + 16 * (t >> 2) + 4 * g + (t & 3); }

// bias image in the S^T accumulator order: img[head][I][J][lane][e] = table[rpi(query 16 I + c, key 16 J + 4 g + e)][head]
__device__ __forceinline__ int w2_img_index(int I, int J, int lane) { return ((I * 4 + J) * 64 + lane) * 4; }

// forward of one (window, head) pair by the calling wave: softmax(scale q k^T + bias + mask) v -> out rows (token-major,
// head-major channels).  No LDS, no block-level barrier.
template <int D>
__device__ __forceinline__ void w2_fwd_body(const float* qkv, float* out,
                                            const float* __restrict__ biasF, const W2Geom& geo, int H, int W, int C,
                                            int shift, float scale, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const long C3 = 3L * C;
  const float* qb = qkv + geo.head * D;

  // ---- row-form fragments of K and Q, their block exponents
  int tok[4];
#pragma unroll
  for (int T = 0; T < 4; ++T) tok[T] = w2_token(geo, 16 * T + c, H, W, shift);
  float raw[4][8];
  u32x4 kh[4], kl[4], qh[4], ql[4];
  float rk[4], rq[4];
#pragma unroll
  for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], qb + C, C3, tok[T], g);
#pragma unroll
  for (int T = 0; T < 4; ++T) rk[T] = w2_split_row(raw[T], kh[T], kl[T]);
#pragma unroll
  for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], qb, C3, tok[T], g);
#pragma unroll
  for (int T = 0; T < 4; ++T) rq[T] = w2_split_row(raw[T], qh[T], ql[T]) * scale;

  // ---- V^T operand, gathered in the k order of the P registers: lane (r, g) = head-dim entry 16 jd + r of the keys
  // w2_kpos(JJ, g, 0..7); one power-of-two scale per head-dim column (over all 64 keys)
  float vraw[2][2][8];
#pragma unroll
  for (int JJ = 0; JJ < 2; ++JJ)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int vt = w2_token(geo, w2_kpos(JJ, g, t), H, W, shift);
      const float* p = qb + 2 * C + (long)vt * C3 + c;
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) vraw[JJ][jd][t] = (16 * jd + c < D) ? ldg_f(p + 16 * jd) : 0.f;
    }

  // ---- S^T = K . Q^T, 16 tiles: lane (c, g) holds query 16 I + c against keys 16 J + 4 g + e
  f32x4 S[4][4];      // [I][J]
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = mfma16h(kh[J], ql[I], a);
      a = mfma16h(kl[J], qh[I], a);
      a = mfma16h(kh[J], qh[I], a);
      S[I][J] = a;
    }
  // 2^-s of key 16 J + 4 g + e lives in the lanes r = 4 g + e
  float rkk[4][4];
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int e = 0; e < 4; ++e) rkk[J][e] = __shfl(rk[J], 4 * g + e, 64);

  // V columns: scale and split
  u32x4 vh[2][2], vl[2][2];
  float rvv[2];
#pragma unroll
  for (int jd = 0; jd < 2; ++jd) {
    float mx = 0.f;
#pragma unroll
    for (int JJ = 0; JJ < 2; ++JJ)
#pragma unroll
      for (int t = 0; t < 8; ++t) mx = fmaxf(mx, fabsf(vraw[JJ][jd][t]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float sc = pow2_scale(mx);
    rvv[jd] = pow2_inv(sc);
#pragma unroll
    for (int JJ = 0; JJ < 2; ++JJ) {
      unsigned h[4], l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) split2_pair(vraw[JJ][jd][2 * t] * sc, vraw[JJ][jd][2 * t + 1] * sc, h[t], l[t]);
      vh[JJ][jd] = u32x4{h[0], h[1], h[2], h[3]};
      vl[JJ][jd] = u32x4{l[0], l[1], l[2], l[3]};
    }
  }
  // 2^-s of head-dim entry 16 jd + 4 g + e (the O^T rows of this lane), times the 2^-14 of P
  float rvo[2][4];
#pragma unroll
  for (int jd = 0; jd < 2; ++jd)
#pragma unroll
    for (int e = 0; e < 4; ++e) rvo[jd][e] = __shfl(rvv[jd], 4 * g + e, 64) * (1.0f / 16384.f);

  // ---- softmax over the keys of each query and O^T = V^T . P^T, one query tile at a time
  const bool lane_masked = geo.last_col && (((c >> 2) & 1) != (g & 1));
  const float* bimg = biasF + (long)geo.head * 4096;
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    float mx = -3.0e38f;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const f32x4 bv = *(const f32x4*)(bimg + w2_img_index(I, J, lane));
      const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = S[I][J][e] * (rq[I] * rkk[J][e]) + bv[e];
        s += masked ? -100.f : 0.f;
        S[I][J][e] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = __expf(S[I][J][e] - mx);
        S[I][J][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    f32x4 O[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int JJ = 0; JJ < 2; ++JJ) {
      unsigned h[4], l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {        // slots (2t, 2t+1): tile J = 2 JJ + (t >> 1), registers 2 (t & 1), + 1
        const f32x4 pv = S[I][2 * JJ + (t >> 1)];
        const int e0 = 2 * (t & 1);
        split2_pair(pv[e0] * 16384.f, pv[e0 + 1] * 16384.f, h[t], l[t]);
      }
      const u32x4 ph = u32x4{h[0], h[1], h[2], h[3]}, pl = u32x4{l[0], l[1], l[2], l[3]};
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        O[jd] = mfma16h(vh[JJ][jd], pl, O[jd]);
        O[jd] = mfma16h(vl[JJ][jd], ph, O[jd]);
        O[jd] = mfma16h(vh[JJ][jd], ph, O[jd]);
      }
    }
    // lane (c, g): query 16 I + c, head-dim entries 16 jd + 4 g + e
    float* op = out + (long)tok[I] * C + geo.head * D + 4 * g;
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      const int d0 = 16 * jd + 4 * g;
      if (d0 < D) *(float2*)(op + 16 * jd) = float2{O[jd][0] * (rvo[jd][0] * inv), O[jd][1] * (rvo[jd][1] * inv)};
      if (d0 + 2 < D) *(float2*)(op + 16 * jd + 2) = float2{O[jd][2] * (rvo[jd][2] * inv), O[jd][3] * (rvo[jd][3] * inv)};
    }
  }
}

}  // namespace
