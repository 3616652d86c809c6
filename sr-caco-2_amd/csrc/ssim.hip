// SSIM as a training loss (zero-padded "same" Gaussian window, value + gradient)
// and as an evaluation metric (11x11 "valid" window, optional ROI slots).
// HBM/LDS-bound separable Gaussian filtering; one 32x32 output tile per block.
//
// Loss  : dlib/loss/ssim.py:23-61, dlib/loss/main.py:154-186
// Metric: dlib/utils/utils_image.py:1010-1198
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TS = 32;        // output tile edge
constexpr int MAXWS = 19;     // largest window (README recipe uses 19)
constexpr int TIN = TS + MAXWS - 1;

struct Taps { float g[MAXWS]; };

__host__ Taps gaussian_taps(int ws, bool centered_half) {
  // loss: exp(-(i - ws//2)^2 / (2 sigma^2)) (ssim.py:23-27);
  // metric: coords - (ks-1)/2 (utils_image.py:1102-1117) -- identical for odd ws
  Taps t;
  float g[MAXWS];
  float s = 0.f;
  for (int i = 0; i < ws; ++i) {
    const float d = centered_half ? (float)i - (float)(ws - 1) / 2.0f : (float)(i - ws / 2);
    g[i] = expf(-(d * d) / (2.0f * 1.5f * 1.5f));
    s += g[i];
  }
  for (int i = 0; i < MAXWS; ++i) t.g[i] = i < ws ? g[i] / s : 0.f;
  return t;
}

__device__ __forceinline__ float u8f(float v) {
  v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
  return fminf(fmaxf(rintf(v), 0.f), 255.f);
}

// Blur NQ quantities derived from (x,y) over a TS x TS tile whose inputs start
// at (iy0, ix0) in the image (may be negative: zero padding).  Results land in
// out[q][4] for this thread's 4 output pixels (rows ty+8k, col tx).
template <int NQ, typename F>
__device__ __forceinline__ void blur_tile(const float* __restrict__ X, const float* __restrict__ Y,
                                          int H, int W, int iy0, int ix0, int ws, const Taps& tp,
                                          float* sx, float* sy, float* sh, F quant, float (&out)[NQ][4]) {
  const int tid = threadIdx.x;
  const int tin = TS + ws - 1;
  for (int i = tid; i < tin * tin; i += 256) {
    const int r = i / tin, c = i - r * tin;
    const int y = iy0 + r, x = ix0 + c;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    sx[r * TIN + c] = ok ? X[(long)y * W + x] : 0.f;
    sy[r * TIN + c] = ok ? Y[(long)y * W + x] : 0.f;
  }
  __syncthreads();
  // horizontal pass: sh[q][r][c], r < tin, c < TS
  for (int i = tid; i < tin * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    float a[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = 0.f;
    for (int k = 0; k < ws; ++k) {
      float v[NQ];
      quant(sx[r * TIN + c + k], sy[r * TIN + c + k], v);
#pragma unroll
      for (int q = 0; q < NQ; ++q) a[q] += tp.g[k] * v[q];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) sh[(q * TIN + r) * TS + c] = a[q];
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int r = ty + 8 * k4;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float a = 0.f;
      for (int k = 0; k < ws; ++k) a += tp.g[k] * sh[(q * TIN + r + k) * TS + tx];
      out[q][k4] = a;
    }
  }
  __syncthreads();
}

// ---------------- loss, pass 1: ssim map -> partial sums + 3 gradient maps ----
__global__ void __launch_bounds__(256) k_ssim_loss_fwd(const float* __restrict__ P, const float* __restrict__ T,
                                                       float* __restrict__ maps, double* __restrict__ part,
                                                       int H, int W, int ws, Taps tp) {
  __shared__ float sx[TIN * TIN], sy[TIN * TIN], sh[5 * TIN * TS];
  __shared__ double red[4];
  const int b = blockIdx.z, oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
  const long img = (long)b * H * W;
  float o[5][4];
  blur_tile<5>(P + img, T + img, H, W, oy0 - ws / 2, ox0 - ws / 2, ws, tp, sx, sy, sh,
               [](float x, float y, float (&v)[5]) { v[0] = x; v[1] = y; v[2] = x * x; v[3] = y * y; v[4] = x * y; },
               o);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  double acc = 0.0;
  const long plane = (long)gridDim.z * H * W;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int y = oy0 + ty + 8 * k4, x = ox0 + tx;
    if (y < H && x < W) {
      const float m1 = o[0][k4], m2 = o[1][k4];
      const float s11 = o[2][k4] - m1 * m1, s22 = o[3][k4] - m2 * m2, s12 = o[4][k4] - m1 * m2;
      const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2;
      const float B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
      const float inv = 1.f / (B1 * B2);
      const float S = A1 * A2 * inv;
      acc += (double)S;
      // partials w.r.t. blurred quantities of the FIRST image (pred)
      const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -S / B1, dB2 = -S / B2;
      const float dm1 = dA1 * 2.f * m2 + dB1 * 2.f * m1 - dA2 * 2.f * m2 - dB2 * 2.f * m1;
      const long o3 = img + (long)y * W + x;
      maps[o3] = dm1;
      maps[plane + o3] = dB2;          // d/d blur(x*x)
      maps[2 * plane + o3] = 2.f * dA2;  // d/d blur(x*y)
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    part[((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// ---------------- loss, pass 2: grad = c*(G*dm1 + 2x G*de11 + y G*de12) --------
__global__ void __launch_bounds__(256) k_ssim_loss_bwd(const float* __restrict__ P, const float* __restrict__ T,
                                                       const float* __restrict__ maps, float* __restrict__ grad,
                                                       int H, int W, int ws, Taps tp, float c, int accum) {
  __shared__ float sx[TIN * TIN], sy[TIN * TIN], sh[2 * TIN * TS];
  const int b = blockIdx.z, oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
  const long img = (long)b * H * W, plane = (long)gridDim.z * H * W;
  float o01[2][4], o2[1][4];
  blur_tile<2>(maps + img, maps + plane + img, H, W, oy0 - ws / 2, ox0 - ws / 2, ws, tp, sx, sy, sh,
               [](float x, float y, float (&v)[2]) { v[0] = x; v[1] = y; }, o01);
  blur_tile<1>(maps + 2 * plane + img, maps + 2 * plane + img, H, W, oy0 - ws / 2, ox0 - ws / 2, ws, tp,
               sx, sy, sh, [](float x, float y, float (&v)[1]) { v[0] = x; }, o2);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int y = oy0 + ty + 8 * k4, x = ox0 + tx;
    if (y < H && x < W) {
      const long o = img + (long)y * W + x;
      const float g = c * (o01[0][k4] + 2.f * P[o] * o01[1][k4] + T[o] * o2[0][k4]);
      grad[o] = accum ? grad[o] + g : g;
    }
  }
}
__global__ void __launch_bounds__(1024) k_sum_partials_d(const double* __restrict__ part, int n, double scale,
                                 float* __restrict__ out, int accum) {
  __shared__ double sh[16];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) a += part[i];
  a = wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    out[0] = (accum ? out[0] : 0.f) + (float)(t * scale);
  }
}

// ---------------- metric -------------------------------------------------------
constexpr int MAXTH = 9;
// raw[b][slot][2] doubles (sum, count), accumulated with atomics (zeroed by the caller)
__global__ void __launch_bounds__(256) k_ssim_metric(const float* __restrict__ E, const float* __restrict__ Hh,
                                                     double* __restrict__ raw, int H, int W, int border,
                                                     const int* __restrict__ ths, int nth, int in_is_u8,
                                                     Taps tp) {
  __shared__ float sx[TIN * TIN], sy[TIN * TIN], sh[5 * TIN * TS];
  const int ws = 11;
  const int b = blockIdx.z, oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
  const int h = H - 2 * border, w = W - 2 * border;      // cropped image
  const int oh = h - (ws - 1), ow = w - (ws - 1);        // valid output
  const long img = (long)b * H * W;
  const int tid = threadIdx.x;
  // stage the cropped inputs (u8-ised, /255) -- valid conv: inputs start at the output origin
  const int tin = TS + ws - 1;
  for (int i = tid; i < tin * tin; i += 256) {
    const int r = i / tin, c = i - r * tin;
    const int y = oy0 + r, x = ox0 + c;
    float a = 0.f, t = 0.f;
    if (y < h && x < w) {
      const long o = img + (long)(y + border) * W + x + border;
      a = (in_is_u8 ? E[o] : u8f(E[o])) / 255.0f;
      t = (in_is_u8 ? Hh[o] : u8f(Hh[o])) / 255.0f;
    }
    sx[r * TIN + c] = a; sy[r * TIN + c] = t;
  }
  __syncthreads();
  for (int i = tid; i < tin * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    for (int k = 0; k < ws; ++k) {
      const float x = sx[r * TIN + c + k], y = sy[r * TIN + c + k], g = tp.g[k];
      a0 += g * x; a1 += g * y; a2 += g * x * x; a3 += g * y * y; a4 += g * x * y;
    }
    sh[(0 * TIN + r) * TS + c] = a0; sh[(1 * TIN + r) * TS + c] = a1; sh[(2 * TIN + r) * TS + c] = a2;
    sh[(3 * TIN + r) * TS + c] = a3; sh[(4 * TIN + r) * TS + c] = a4;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  double sum[MAXTH], cnt[MAXTH];
#pragma unroll
  for (int k = 0; k < MAXTH; ++k) { sum[k] = 0; cnt[k] = 0; }
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int r = ty + 8 * k4;
    const int y = oy0 + r, x = ox0 + tx;
    if (y < oh && x < ow) {
      float m[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        float a = 0.f;
        for (int k = 0; k < ws; ++k) a += tp.g[k] * sh[(q * TIN + r + k) * TS + tx];
        m[q] = a;
      }
      const float sxx = m[2] - m[0] * m[0], syy = m[3] - m[1] * m[1], sxy = m[4] - m[0] * m[1];
      const float cs = (2.f * sxy + C2) / (sxx + syy + C2);
      const float ss = ((2.f * m[0] * m[1] + C1) / (m[0] * m[0] + m[1] * m[1] + C1)) * cs;
      // ROI pixel aligned with this output: centre of the window
      const float tcen = rintf(sy[(r + 5) * TIN + tx + 5] * 255.0f);
#pragma unroll
      for (int k = 0; k < MAXTH; ++k)
        if (k <= nth && (k == 0 || tcen >= (float)ths[k - 1])) { sum[k] += (double)ss; cnt[k] += 1.0; }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXTH; ++k) {
    if (k <= nth) {
      const double s = wave_sum_d(sum[k]), c = wave_sum_d(cnt[k]);
      if ((tid & 63) == 0 && c > 0) {
        atomicAdd(raw + ((long)b * (nth + 1) + k) * 2, s);
        atomicAdd(raw + ((long)b * (nth + 1) + k) * 2 + 1, c);
      }
    }
  }
}
__global__ void k_ssim_metric_fin(const double* __restrict__ raw, float* __restrict__ out, int n, int nth1,
                                  double npix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double c = (i % nth1 == 0) ? npix : (raw[2 * i + 1] == 0 ? 1.0 : raw[2 * i + 1]);
  out[i] = (float)(raw[2 * i] / c);
}

}  // namespace

extern "C" {

long srhip_ssim_loss_ws(int B, int H, int W) {
  const long tiles = (long)B * sr_cdiv(H, TS) * sr_cdiv(W, TS);
  return 3L * B * H * W + 2 * tiles + 16;   // 3 maps + double partials
}

int srhip_ssim_loss(const float* pred, const float* target, float* grad, float* loss_out,
                    float* workspace, int B, int H, int W, int ws, float lam, int grad_accum,
                    int loss_accum, void* stream) {
  SR_REQUIRE(ws % 2 == 1 && ws >= 3 && ws <= MAXWS, "ssim_loss: window %d (odd, <= %d)", ws, MAXWS);
  SR_REQUIRE(B > 0 && H > 0 && W > 0, "ssim_loss: empty input");
  hipStream_t st = (hipStream_t)stream;
  const Taps tp = gaussian_taps(ws, false);
  dim3 grid(sr_cdiv(W, TS), sr_cdiv(H, TS), B);
  float* maps = workspace;
  const long nmap = 3L * B * H * W;
  // double partials behind the maps, 8-byte aligned
  double* part = (double*)(((uintptr_t)(workspace + nmap) + 7) & ~(uintptr_t)7);
  const int ntile = grid.x * grid.y * grid.z;
  const double scale = -(double)lam / ((double)B * H * W);
  hipLaunchKernelGGL(k_ssim_loss_fwd, grid, dim3(256), 0, st, pred, target, maps, part, H, W, ws, tp);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, part, ntile, scale, loss_out, loss_accum);
  if (grad)
    hipLaunchKernelGGL(k_ssim_loss_bwd, grid, dim3(256), 0, st, pred, target, maps, grad, H, W, ws, tp,
                       (float)scale, grad_accum);
  SR_LAUNCH_CHECK("ssim_loss");
  return 0;
}

// workspace doubles: B*(nth+1)*2
int srhip_metrics_ssim(const float* E, const float* Hh, int B, int H, int W, int border,
                       const int* thresholds_dev, int nth, int inputs_are_u8, double* workspace,
                       float* out, void* stream) {
  SR_REQUIRE(nth >= 0 && nth <= MAXTH - 1, "metrics_ssim: at most %d ROI thresholds", MAXTH - 1);
  const int h = H - 2 * border, w = W - 2 * border;
  SR_REQUIRE(B > 0 && h >= 11 && w >= 11, "metrics_ssim: image smaller than the 11x11 window");
  hipStream_t st = (hipStream_t)stream;
  const Taps tp = gaussian_taps(11, true);
  const int n = B * (nth + 1);
  (void)hipMemsetAsync(workspace, 0, sizeof(double) * 2 * n, st);
  dim3 grid(sr_cdiv(w - 10, TS), sr_cdiv(h - 10, TS), B);
  hipLaunchKernelGGL(k_ssim_metric, grid, dim3(256), 0, st, E, Hh, workspace, H, W, border,
                     thresholds_dev, nth, inputs_are_u8, tp);
  hipLaunchKernelGGL(k_ssim_metric_fin, dim3(sr_cdiv(n, 64)), dim3(64), 0, st, workspace, out, n, nth + 1,
                     (double)(h - 10) * (w - 10));
  SR_LAUNCH_CHECK("metrics_ssim");
  return 0;
}

}  // extern "C"
