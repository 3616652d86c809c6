// The Fourier channel attention of DFCAN, evaluation forward -- reference RCAB.forward, dlib/models/network_dfcan.py:39-70:
// the spectrum's magnitude |FFT2(x)|^gamma (+1e-8 inside the power, :62-63), the quadrant swap fftshift2d (:27-36), and
// behind the conv + ReLU on it the channel gate: global average, 64 -> 4 -> 64 with ReLU / sigmoid (:65-68), out = x0 +
// x1 * gate (:69-70).  Plus the two point-wise activations the net uses outside conv epilogues (GELU, sigmoid).
//
// The transform is a separable DFT on channels-last data: the channel index is the contiguous one, so a row (column)
// transform is a small dense product with the channels as the coalesced batch dimension -- per output W (H) terms, f64
// accumulation, twiddles from sincospi of the exactly reduced index.  LR feature maps are <= 256 x 256 here; an FFT
// butterfly would save flops that do not matter next to the net's convs.
#include "common.h"

namespace {

constexpr int DF_C = 64;      // channels per block pass (C is handled in groups of 64)

// pass 1: along W.  block = (b, h) row, thread = (four channels 4 cg .. 4 cg + 3, output v stride 16): the twiddle of
// (v, w) is read once for four channels and the four inputs come as one 16-byte LDS read -- the loop was three LDS reads
// per two f64 FMAs (63 us for 0.27 G FMAs), now three per eight.  Every output keeps its own ascending-w sum: same bits.
__global__ void __launch_bounds__(256) k_dft_rows(const float* __restrict__ x, float2* __restrict__ T, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double* const tw = (double*)smraw;                 // [W][2] cos, sin of 2 pi k / W
  float* const row = (float*)(tw + 2 * W);           // [W][DF_C]
  const int tid = threadIdx.x, c = (tid & 15) * 4, v0 = tid >> 4;
  const long bh = blockIdx.x;
  for (int k = tid; k < W; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)W, &s, &co);
    tw[2 * k] = co; tw[2 * k + 1] = s;
  }
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    __syncthreads();
    for (int i = tid; i < W * DF_C; i += 256) {
      const int w = i >> 6, cc = i & 63;
      row[i] = c0 + cc < C ? x[(bh * W + w) * C + c0 + cc] : 0.f;
    }
    __syncthreads();
    for (int v = v0; v < W; v += 16) {
      double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
      int k = 0;                                       // (v * w) mod W, incrementally
      for (int w = 0; w < W; ++w) {
        const f32x4 xv = *(const f32x4*)(row + w * DF_C + c);
        const double co = tw[2 * k], si = tw[2 * k + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          re[e] += (double)xv[e] * co;
          im[e] -= (double)xv[e] * si;
        }
        k += v; if (k >= W) k -= W;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c + e < C) T[(bh * W + v) * C + c0 + c + e] = float2{(float)re[e], (float)im[e]};
    }
  }
}

// pass 2: along H, then magnitude^gamma and the quadrant swap.  block = (b, v) column
__global__ void __launch_bounds__(256) k_dft_cols_mag(const float2* __restrict__ T, float* __restrict__ out, int H, int W, int C,
                                                      float gamma, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double* const tw = (double*)smraw;                 // [H][2]
  float2* const col = (float2*)(tw + 2 * H);         // [H][DF_C]
  const int tid = threadIdx.x, c = (tid & 15) * 4, u0 = tid >> 4;      // four channels per thread, as in pass 1
  const int b = blockIdx.x / W, v = blockIdx.x % W;
  for (int k = tid; k < H; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)H, &s, &co);
    tw[2 * k] = co; tw[2 * k + 1] = s;
  }
  const int jv = (v - W / 2 + W) % W;                // fftshift2d: out[i][j] = img[(i + H//2) % H][(j + W//2) % W]
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    __syncthreads();
    for (int i = tid; i < H * DF_C; i += 256) {
      const int h = i >> 6, cc = i & 63;
      col[i] = c0 + cc < C ? T[(((long)b * H + h) * W + v) * C + c0 + cc] : float2{0.f, 0.f};
    }
    __syncthreads();
    for (int u = u0; u < H; u += 16) {
      double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
      int k = 0;
      for (int h = 0; h < H; ++h) {
        const f32x4 p0 = *(const f32x4*)(col + h * DF_C + c), p1 = *(const f32x4*)(col + h * DF_C + c + 2);
        const double co = tw[2 * k], s = tw[2 * k + 1];        // e^{-i t} = co - i s
        const double a[4] = {(double)p0[0], (double)p0[2], (double)p1[0], (double)p1[2]};
        const double bb[4] = {(double)p0[1], (double)p0[3], (double)p1[1], (double)p1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          re[e] += a[e] * co + bb[e] * s;
          im[e] += bb[e] * co - a[e] * s;
        }
        k += u; if (k >= H) k -= H;
      }
      const int iu = (u - H / 2 + H) % H;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c + e < C) {
          const float mag = (float)sqrt(re[e] * re[e] + im[e] * im[e]);
          out[(((long)b * H + iu) * W + jv) * C + c0 + c + e] = powf(mag + eps, gamma);
        }
    }
  }
}

// ---- backward of the spectrum magnitude (training; round 5: replaces the torch.fft calls of Tape.fourier_gate).  With
// F = FFT2(x) (unnormalised) and g the gradient of out = fftshift2d((|F| + eps)^gamma):
//     G[u][v] = g[iu][jv] gamma (|F| + eps)^(gamma - 1) / |F|   (0 where |F| = 0),   dx = Re( sum_uv G F e^{+i theta} )
// since d|F| / dx[h][w] = Re(F e^{+i theta}) / |F|, theta = 2 pi (u h / H + v w / W): an inverse DFT (no 1 / N) of Z = G F.
// Pass 1 is the forward's k_dft_rows; pass 2 below forms Z in place of T (a block owns its (b, v) column: loaded to LDS
// whole before anything is written); pass 3 the inverse transform along H, in place again; pass 4 the inverse along W,
// real part only.  Same structure as the forward passes: four channels per thread, f64 accumulation, ascending sums.
__global__ void __launch_bounds__(256) k_dft_cols_gz(float2* __restrict__ T, const float* __restrict__ g, int H, int W, int C,
                                                     float gamma, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double* const tw = (double*)smraw;                 // [H][2]
  float2* const col = (float2*)(tw + 2 * H);         // [H][DF_C]
  const int tid = threadIdx.x, c = (tid & 15) * 4, u0 = tid >> 4;
  const int b = blockIdx.x / W, v = blockIdx.x % W;
  for (int k = tid; k < H; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)H, &s, &co);
    tw[2 * k] = co; tw[2 * k + 1] = s;
  }
  const int jv = (v - W / 2 + W) % W;
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    __syncthreads();
    for (int i = tid; i < H * DF_C; i += 256) {
      const int h = i >> 6, cc = i & 63;
      col[i] = c0 + cc < C ? T[(((long)b * H + h) * W + v) * C + c0 + cc] : float2{0.f, 0.f};
    }
    __syncthreads();
    for (int u = u0; u < H; u += 16) {
      double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
      int k = 0;
      for (int h = 0; h < H; ++h) {
        const f32x4 p0 = *(const f32x4*)(col + h * DF_C + c), p1 = *(const f32x4*)(col + h * DF_C + c + 2);
        const double co = tw[2 * k], s = tw[2 * k + 1];        // e^{-i t} = co - i s
        const double a[4] = {(double)p0[0], (double)p0[2], (double)p1[0], (double)p1[2]};
        const double bb[4] = {(double)p0[1], (double)p0[3], (double)p1[1], (double)p1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          re[e] += a[e] * co + bb[e] * s;
          im[e] += bb[e] * co - a[e] * s;
        }
        k += u; if (k >= H) k -= H;
      }
      const int iu = (u - H / 2 + H) % H;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c + e < C) {
          const float mag = (float)sqrt(re[e] * re[e] + im[e] * im[e]);
          const float g0 = g[(((long)b * H + iu) * W + jv) * C + c0 + c + e];
          const float G = mag > 0.f ? g0 * gamma * powf(mag + eps, gamma - 1.0f) / fmaxf(mag, 1e-30f) : 0.f;
          T[(((long)b * H + u) * W + v) * C + c0 + c + e] = float2{G * (float)re[e], G * (float)im[e]};
        }
    }
  }
}
// pass 3: Y[h] = sum_u Z[u] e^{+2 pi i u h / H} along H, in place (block = (b, v) column)
__global__ void __launch_bounds__(256) k_idft_cols(float2* __restrict__ T, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double* const tw = (double*)smraw;
  float2* const col = (float2*)(tw + 2 * H);
  const int tid = threadIdx.x, c = (tid & 15) * 4, h0 = tid >> 4;
  const int b = blockIdx.x / W, v = blockIdx.x % W;
  for (int k = tid; k < H; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)H, &s, &co);
    tw[2 * k] = co; tw[2 * k + 1] = s;
  }
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    __syncthreads();
    for (int i = tid; i < H * DF_C; i += 256) {
      const int u = i >> 6, cc = i & 63;
      col[i] = c0 + cc < C ? T[(((long)b * H + u) * W + v) * C + c0 + cc] : float2{0.f, 0.f};
    }
    __syncthreads();
    for (int h = h0; h < H; h += 16) {
      double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
      int k = 0;
      for (int u = 0; u < H; ++u) {
        const f32x4 p0 = *(const f32x4*)(col + u * DF_C + c), p1 = *(const f32x4*)(col + u * DF_C + c + 2);
        const double co = tw[2 * k], s = tw[2 * k + 1];        // e^{+i t} = co + i s
        const double a[4] = {(double)p0[0], (double)p0[2], (double)p1[0], (double)p1[2]};
        const double bb[4] = {(double)p0[1], (double)p0[3], (double)p1[1], (double)p1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          re[e] += a[e] * co - bb[e] * s;
          im[e] += bb[e] * co + a[e] * s;
        }
        k += h; if (k >= H) k -= H;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c + e < C) T[(((long)b * H + h) * W + v) * C + c0 + c + e] = float2{(float)re[e], (float)im[e]};
    }
  }
}
// pass 4: dx[w] = Re sum_v Y[v] e^{+2 pi i v w / W} along W (block = (b, h) row)
__global__ void __launch_bounds__(256) k_idft_rows_real(const float2* __restrict__ T, float* __restrict__ dx, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  double* const tw = (double*)smraw;                 // [W][2]
  float2* const row = (float2*)(tw + 2 * W);         // [W][DF_C]
  const int tid = threadIdx.x, c = (tid & 15) * 4, w0 = tid >> 4;
  const long bh = blockIdx.x;
  for (int k = tid; k < W; k += 256) {
    double s, co;
    sincospi(2.0 * (double)k / (double)W, &s, &co);
    tw[2 * k] = co; tw[2 * k + 1] = s;
  }
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    __syncthreads();
    for (int i = tid; i < W * DF_C; i += 256) {
      const int v = i >> 6, cc = i & 63;
      row[i] = c0 + cc < C ? T[(bh * W + v) * C + c0 + cc] : float2{0.f, 0.f};
    }
    __syncthreads();
    for (int w = w0; w < W; w += 16) {
      double re[4] = {0.0, 0.0, 0.0, 0.0};
      int k = 0;
      for (int v = 0; v < W; ++v) {
        const f32x4 p0 = *(const f32x4*)(row + v * DF_C + c), p1 = *(const f32x4*)(row + v * DF_C + c + 2);
        const double co = tw[2 * k], s = tw[2 * k + 1];
        re[0] += (double)p0[0] * co - (double)p0[1] * s;
        re[1] += (double)p0[2] * co - (double)p0[3] * s;
        re[2] += (double)p1[0] * co - (double)p1[1] * s;
        re[3] += (double)p1[2] * co - (double)p1[3] * s;
        k += w; if (k >= W) k -= W;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c + e < C) dx[(bh * W + w) * C + c0 + c + e] = (float)re[e];
    }
  }
}

// global average over the pixels: partial sums per (sample, pixel block), then ...
__global__ void __launch_bounds__(256) k_pool_partial(const float* __restrict__ x, double* __restrict__ part, long P, int C,
                                                      int nblk) {
  __shared__ double sm[4][DF_C];
  const int tid = threadIdx.x, c = tid & 63, r = tid >> 6;
  const int b = blockIdx.x / nblk, pb = blockIdx.x % nblk;
  const long p0 = (P * pb) / nblk, p1 = (P * (pb + 1)) / nblk;
  for (int c0 = 0; c0 < C; c0 += DF_C) {
    double a = 0.0;
    if (c0 + c < C)
      for (long p = p0 + r; p < p1; p += 4) a += (double)x[((long)b * P + p) * C + c0 + c];
    sm[r][c] = a;
    __syncthreads();
    if (r == 0 && c0 + c < C) part[((long)b * nblk + pb) * C + c0 + c] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
    __syncthreads();
  }
}
// ... the gate of one sample: mean, C -> Cm with ReLU, Cm -> C with sigmoid (1x1 convs on a 1x1 map, :66-68)
__global__ void __launch_bounds__(256) k_gate(const double* __restrict__ part, int nblk, long P, const float* __restrict__ w1,
                                              const float* __restrict__ b1, const float* __restrict__ w2,
                                              const float* __restrict__ b2, float* __restrict__ gate, int C, int Cm,
                                              int mid_act) {
  __shared__ float mean[256], mid[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < C) {
    double a = 0.0;
    for (int k = 0; k < nblk; ++k) a += part[((long)b * nblk + k) * C + tid];
    mean[tid] = (float)(a / (double)P);
  }
  __syncthreads();
  if (tid < Cm) {
    float a = b1 ? b1[tid] : 0.f;
    for (int k = 0; k < C; ++k) a += w1[tid * C + k] * mean[k];
    mid[tid] = mid_act ? a / (1.0f + expf(-a)) : fmaxf(a, 0.f);        // SiLU | ReLU
  }
  __syncthreads();
  if (tid < C) {
    float a = b2 ? b2[tid] : 0.f;
    for (int k = 0; k < Cm; ++k) a += w2[tid * Cm + k] * mid[k];
    gate[(long)b * C + tid] = 1.0f / (1.0f + expf(-a));
  }
}
// out = x0 + x1 * gate[sample][channel]
__global__ void __launch_bounds__(256) k_gate_apply(const float* __restrict__ x0, const float* __restrict__ x1,
                                                    const float* __restrict__ gate, float* __restrict__ out, long P, int C,
                                                    long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long b = i / (P * C);
    out[i] = (x0 ? x0[i] : 0.f) + x1[i] * gate[b * C + c];
  }
}
// kind 0: exact-erf GELU (nn.GELU default); 1: sigmoid
__global__ void __launch_bounds__(256) k_unary(const float* __restrict__ x, float* __restrict__ out, long n, int kind) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    out[i] = kind == 0 ? gelu_f(v) : 1.0f / (1.0f + expf(-v));
  }
}

// backward of k_unary: gelu from its INPUT x (exact erf derivative), sigmoid from its OUTPUT y
__global__ void __launch_bounds__(256) k_unary_bwd(const float* __restrict__ xy, const float* __restrict__ g, float* __restrict__ dx,
                                                   long n, int kind) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = xy[i];
    dx[i] = g[i] * (kind == 0 ? dgelu_f(v) : v * (1.0f - v));
  }
}

inline int ew_blocks(long n) { const long g = (n + 255) / 256; return (int)(g < 8192 ? g : 8192); }

}  // namespace

extern "C" {

/* out[b][i][j][c] = (|FFT2(x[b][.][.][c])| + eps)^gamma under fftshift2d; x, out [B][H][W][C]; workspace: 2 * B*H*W*C floats.
 * H, W <= 256 (one row / column of 64 channels in LDS). */
int srhip_fft2_mag_pow_shift(const float* x, float* out, float* workspace, int B, int H, int W, int C, float gamma, float eps,
                             void* stream) {
  SR_REQUIRE(x && out && workspace && B > 0 && H > 0 && W > 0 && C > 0, "fft2_mag_pow_shift: bad arguments");
  SR_REQUIRE(H <= 256 && W <= 256, "fft2_mag_pow_shift: H, W <= 256 (H=%d W=%d)", H, W);
  hipStream_t st = (hipStream_t)stream;
  const size_t l1 = (size_t)W * 16 + (size_t)W * DF_C * 4, l2 = (size_t)H * 16 + (size_t)H * DF_C * 8;
  static size_t r1 = 0, r2 = 0;
  if (l1 > r1) {
    if (hipFuncSetAttribute((const void*)k_dft_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1) != hipSuccess)
      return sr_fail(-5, "fft2_mag_pow_shift: cannot reserve %zu bytes of LDS", l1);
    r1 = l1;
  }
  if (l2 > r2) {
    if (hipFuncSetAttribute((const void*)k_dft_cols_mag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess)
      return sr_fail(-5, "fft2_mag_pow_shift: cannot reserve %zu bytes of LDS", l2);
    r2 = l2;
  }
  hipLaunchKernelGGL(k_dft_rows, dim3(B * H), dim3(256), l1, st, x, (float2*)workspace, H, W, C);
  hipLaunchKernelGGL(k_dft_cols_mag, dim3(B * W), dim3(256), l2, st, (const float2*)workspace, out, H, W, C, gamma, eps);
  SR_LAUNCH_CHECK("fft2_mag_pow_shift");
  return 0;
}

/* dx = d loss / d x of srhip_fft2_mag_pow_shift given g = d loss / d out (same shapes; workspace 2*B*H*W*C floats). */
int srhip_fft2_mag_pow_shift_bwd(const float* x, const float* g, float* dx, float* workspace, int B, int H, int W, int C,
                                 float gamma, float eps, void* stream) {
  SR_REQUIRE(x && g && dx && workspace && B > 0 && H > 0 && W > 0 && C > 0, "fft2_mag_pow_shift_bwd: bad arguments");
  SR_REQUIRE(H <= 256 && W <= 256, "fft2_mag_pow_shift_bwd: H, W <= 256 (H=%d W=%d)", H, W);
  hipStream_t st = (hipStream_t)stream;
  const size_t l1 = (size_t)W * 16 + (size_t)W * DF_C * 4, l2 = (size_t)H * 16 + (size_t)H * DF_C * 8;
  const size_t l4 = (size_t)W * 16 + (size_t)W * DF_C * 8;
  static size_t r1 = 0, r2 = 0, r4 = 0;
  if (l1 > r1) {
    if (hipFuncSetAttribute((const void*)k_dft_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1) != hipSuccess)
      return sr_fail(-5, "fft2_mag_pow_shift_bwd: cannot reserve %zu bytes of LDS", l1);
    r1 = l1;
  }
  if (l2 > r2) {
    if (hipFuncSetAttribute((const void*)k_dft_cols_gz, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_idft_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess)
      return sr_fail(-5, "fft2_mag_pow_shift_bwd: cannot reserve %zu bytes of LDS", l2);
    r2 = l2;
  }
  if (l4 > r4) {
    if (hipFuncSetAttribute((const void*)k_idft_rows_real, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4) != hipSuccess)
      return sr_fail(-5, "fft2_mag_pow_shift_bwd: cannot reserve %zu bytes of LDS", l4);
    r4 = l4;
  }
  float2* const T = (float2*)workspace;
  hipLaunchKernelGGL(k_dft_rows, dim3(B * H), dim3(256), l1, st, x, T, H, W, C);
  hipLaunchKernelGGL(k_dft_cols_gz, dim3(B * W), dim3(256), l2, st, T, g, H, W, C, gamma, eps);
  hipLaunchKernelGGL(k_idft_cols, dim3(B * W), dim3(256), l2, st, T, H, W, C);
  hipLaunchKernelGGL(k_idft_rows_real, dim3(B * H), dim3(256), l4, st, (const float2*)T, dx, H, W, C);
  SR_LAUNCH_CHECK("fft2_mag_pow_shift_bwd");
  return 0;
}

long srhip_channel_gate_ws(int B, long P, int C) { return (long)B * (P < 4096 ? 1 : 64) * C; }   /* doubles */

/* gate[b][c] = sigmoid(W2 act(W1 mean_p(feat[b][p][.]) + b1) + b2) (W1 [Cm][C], W2 [C][Cm]; act = ReLU (mid_act 0) or SiLU (1);
 * b1, b2 may be NULL); out = x0 + x1 * gate (x0 may be NULL).  feat, x0, x1, out: [B][P][C] contiguous; C <= 256, Cm <= 64;
 * workspace: srhip_channel_gate_ws doubles; gate: B*C floats. */
int srhip_channel_gate(const float* feat, const float* w1, const float* b1, const float* w2, const float* b2, const float* x0,
                       const float* x1, float* out, float* gate, double* workspace, int B, long P, int C, int Cm,
                       int mid_act, void* stream) {
  SR_REQUIRE(feat && w1 && w2 && x1 && out && gate && workspace && (mid_act == 0 || mid_act == 1), "channel_gate: bad operand");
  SR_REQUIRE(B > 0 && P > 0 && C > 0 && C <= 256 && Cm > 0 && Cm <= 64, "channel_gate: C <= 256, Cm <= 64 (C=%d Cm=%d)", C, Cm);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = P < 4096 ? 1 : 64;
  hipLaunchKernelGGL(k_pool_partial, dim3(B * nblk), dim3(256), 0, st, feat, workspace, P, C, nblk);
  hipLaunchKernelGGL(k_gate, dim3(B), dim3(256), 0, st, (const double*)workspace, nblk, P, w1, b1, w2, b2, gate, C, Cm, mid_act);
  const long n = (long)B * P * C;
  hipLaunchKernelGGL(k_gate_apply, dim3(ew_blocks(n)), dim3(256), 0, st, x0, x1, gate, out, P, C, n);
  SR_LAUNCH_CHECK("channel_gate");
  return 0;
}

/* kind 0: nn.GELU() (exact erf); kind 1: sigmoid.  out may alias x. */
int srhip_unary(const float* x, float* out, long n, int kind, void* stream) {
  SR_REQUIRE(x && out && n > 0 && (kind == 0 || kind == 1), "unary: bad arguments");
  hipLaunchKernelGGL(k_unary, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, kind);
  SR_LAUNCH_CHECK("unary");
  return 0;
}

/* dx = g * f'(.) for srhip_unary's functions: kind 0 (GELU) takes the op's INPUT in xy, kind 1 (sigmoid) its OUTPUT.  dx may be g. */
int srhip_unary_bwd(const float* xy, const float* g, float* dx, long n, int kind, void* stream) {
  SR_REQUIRE(xy && g && dx && n > 0 && (kind == 0 || kind == 1), "unary_bwd: bad arguments");
  hipLaunchKernelGGL(k_unary_bwd, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, xy, g, dx, n, kind);
  SR_LAUNCH_CHECK("unary_bwd");
  return 0;
}

}  // extern "C"
