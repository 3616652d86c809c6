// Small HBM-bound helpers of the generic conv-net engine (srhip/tape.py: DBPN, SRFBN, ProSR): PReLU forward /
// backward with a deterministic slope gradient, strided (channel-slice) axpby, reflection padding and its adjoint.
#include "common.h"
#include "kernels.h"
#include "../../include/srhip.h"

namespace {

inline int to_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// nn.PReLU(num_parameters = 1): y = x > 0 ? x : a x  (dlib/models/network_dbpn.py:85, network_srfbn.py:44)
__global__ void __launch_bounds__(256) k_prelu_fwd(const float* __restrict__ x, const float* __restrict__ alpha,
                                                   float* __restrict__ y, long n4) {
  const float a = ldg_f(alpha);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = ((const f32x4*)x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : a * v[e];
    ((f32x4*)y)[i] = v;
  }
}
// dx = g (x > 0 ? 1 : a);  part[block] = sum over the block's elements of g min(x, 0)  (fp64; summed in block order)
__global__ void __launch_bounds__(256) k_prelu_bwd(const float* __restrict__ g, const float* __restrict__ x,
                                                   const float* __restrict__ alpha, float* __restrict__ dx,
                                                   double* __restrict__ part, long n4) {
  __shared__ double red[4];
  const float a = ldg_f(alpha);
  double acc = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 xv = ((const f32x4*)x)[i];
    f32x4 gv = ((const f32x4*)g)[i];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += xv[e] > 0.f ? 0.f : gv[e] * xv[e];
      gv[e] = xv[e] > 0.f ? gv[e] : a * gv[e];
    }
    acc += (double)s;
    ((f32x4*)dx)[i] = gv;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(64) k_prelu_fin(const double* __restrict__ part, int n, float* __restrict__ dalpha,
                                                  int accumulate) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += part[i];      // lane-strided, then a fixed-order wave sum: deterministic
  a = wave_sum_d(a);
  if (threadIdx.x == 0) dalpha[0] = (float)(a + (accumulate ? (double)dalpha[0] : 0.0));
}

// y[r][c] = a x[r][c] + b y[r][c] on row-major views with row pitches ldy / ldx (channel slices of NHWC tensors)
__global__ void __launch_bounds__(256) k_axpby2d(float* __restrict__ y, long ldy, const float* __restrict__ x, long ldx,
                                                 long rows, int c4, float a, float b) {
  const long n = rows * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const f32x4 xv = *(const f32x4*)(x + r * ldx + c);
    f32x4 yv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (b != 0.f) yv = *(const f32x4*)(y + r * ldy + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) yv[e] = a * xv[e] + b * yv[e];
    *(f32x4*)(y + r * ldy + c) = yv;
  }
}

// nn.ReflectionPad2d(1) on NHWC (network_prosr.py:44-86) and the crop that undoes it; adjoint = 1: the gradient of
// the padding (a gather: every source pixel sums the padded positions that mirror onto it -- no atomics)
__device__ __forceinline__ int refl(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
__global__ void __launch_bounds__(256) k_pad_reflect1(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                      int W, int c4, int adjoint) {
  // forward: in [B][H][W][C] -> out [B][H+2][W+2][C];  adjoint: in = padded gradient [B][H+2][W+2][C] -> out [B][H][W][C]
  const int Ho = adjoint ? H : H + 2, Wo = adjoint ? W : W + 2;
  const long n = (long)B * Ho * Wo * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long t = i;
    const int c = (int)(t % c4) * 4; t /= c4;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const long C = 4L * c4;
    if (!adjoint) {
      const int sy = refl(y - 1, H), sx = refl(x - 1, W);
      *(f32x4*)(out + (((long)b * Ho + y) * Wo + x) * C + c) = *(const f32x4*)(in + (((long)b * H + sy) * W + sx) * C + c);
    } else {
      // padded rows that mirror onto source row y: y + 1 always; padded row 0 onto y = 1; padded row H + 1 onto y = H - 2
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      const long Wp = W + 2, Hp = H + 2;
      // (H == 3: row 1 receives both mirrors)
      int yl[3], xl[3], cy = 0, cx = 0;
      yl[cy++] = y + 1; if (y == 1) yl[cy++] = 0; if (y == H - 2) yl[cy++] = H + 1;
      xl[cx++] = x + 1; if (x == 1) xl[cx++] = 0; if (x == W - 2) xl[cx++] = W + 1;
      for (int iy = 0; iy < cy; ++iy)
        for (int ix = 0; ix < cx; ++ix) {
          const f32x4 v = *(const f32x4*)(in + (((long)b * Hp + yl[iy]) * Wp + xl[ix]) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
      *(f32x4*)(out + (((long)b * H + y) * W + x) * C + c) = a;
    }
  }
}
// crop 1 pixel per side: in [B][H+2][W+2][C] -> out [B][H][W][C];  adjoint: in [B][H][W][C] -> out [B][H+2][W+2][C] zero border
__global__ void __launch_bounds__(256) k_crop1(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                               int c4, int adjoint) {
  const int Ho = adjoint ? H + 2 : H, Wo = adjoint ? W + 2 : W;
  const long n = (long)B * Ho * Wo * c4;
  const long C = 4L * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long t = i;
    const int c = (int)(t % c4) * 4; t /= c4;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!adjoint) v = *(const f32x4*)(in + (((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c);
    else if (y >= 1 && y <= H && x >= 1 && x <= W) v = *(const f32x4*)(in + (((long)b * H + y - 1) * W + x - 1) * C + c);
    *(f32x4*)(out + (((long)b * Ho + y) * Wo + x) * C + c) = v;
  }
}

// ---- ENLCA (network_enlcn.py:207-366): the element-wise pieces around its GEMMs; one wave per row
// F.normalize(x, p=2, dim=channel, eps) * k on token rows
__global__ void __launch_bounds__(256) k_l2norm_rows(float* __restrict__ x, long ld, long T, int C, float eps, float k,
                                                     float* __restrict__ fac) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  float* r = x + t * ld;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += r[c] * r[c];
  const float f = k / fmaxf(sqrtf(wave_sum(ss)), eps);
  for (int c = lane; c < C; c += 64) r[c] *= f;
  if (fac && lane == 0) fac[t] = f;               // training: the row's factor k / max(|x|, eps) for the backward
}
// backward of y = f x, f = k / max(|x|, eps):  dx = f (dy - y (y . dy) / k^2)  (|x| >= eps: the projection off the radial
// direction);  dx = f dy where the norm was clamped (f = k / eps).  In place on dy.
__global__ void __launch_bounds__(256) k_l2norm_rows_bwd(float* __restrict__ dy, long ldd, const float* __restrict__ y, long ldy,
                                                         const float* __restrict__ fac, long T, int C, float eps, float k) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  float* d = dy + t * ldd;
  const float* r = y + t * ldy;
  const float f = fac[t];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += r[c] * d[c];
  dot = wave_sum(dot);
  const float s = (f < k / eps) ? dot / (k * k) : 0.f;
  for (int c = lane; c < C; c += 64) d[c] = f * (d[c] - r[c] * s);
}
// backward of f = ratio (exp(dash - diag) + eps) with respect to dash: g (f - ratio eps), in place on g.  (The diag = |data|^2 / 2
// path is a gradient along data itself: the L2 normalisation in front of it projects exactly that direction out.)
__global__ void __launch_bounds__(256) k_performer_features_bwd(float* __restrict__ g, const float* __restrict__ f, long n, float c) {
  const long i = (blockIdx.x * 256L + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 gv = *(const f32x4*)(g + i);
  const f32x4 fv = *(const f32x4*)(f + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) gv[e] *= fv[e] - c;
  *(f32x4*)(g + i) = gv;
}
// backward of out = x + res_scale num[:, :Cy] / num[:, Cy] with respect to num (the residual path is dout itself):
//   dnum[:, c] = s dout[c], s = res_scale / den;  dnum[:, Cy] = -sum_c s dout[c] num[c] / den;  dnum[:, Cy + 1 ..] = 0
__global__ void __launch_bounds__(256) k_enlca_finish_bwd(const float* __restrict__ dout, const float* __restrict__ num, long ldn,
                                                          float* __restrict__ dnum, long T, int Cy, float res_scale) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  const float* r = num + t * ldn;
  float* o = dnum + t * ldn;
  const float den = r[Cy], s = res_scale / den;
  float acc = 0.f;
  for (int c = lane; c < Cy; c += 64) {
    const float d = s * dout[t * Cy + c];
    o[c] = d;
    acc += d * r[c];
  }
  acc = wave_sum(acc);
  for (int c = Cy + lane; c < (int)ldn; c += 64) o[c] = c == Cy ? -acc / den : 0.f;
}
// softmax_kernel :207-240: out[t][j] = ratio (exp(dash[t][j] - |data[t]|^2 / 2) + eps)
__global__ void __launch_bounds__(256) k_performer_features(float* __restrict__ dash, long ldd, const float* __restrict__ data,
                                                            long ldx, long T, int Fn, int C, float ratio, float eps) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  const float* r = data + t * ldx;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += r[c] * r[c];
  const float diag = wave_sum(ss) * 0.5f;
  float* d = dash + t * ldd;
  for (int j = lane; j < Fn; j += 64) d[j] = ratio * (expf(d[j] - diag) + eps);
}
// linear_attention's last step + ENLCA's residual: out = x + res_scale * num[:, :Cy] / num[:, Cy]
__global__ void __launch_bounds__(256) k_enlca_finish(const float* __restrict__ num, long ldn, const float* __restrict__ x,
                                                      float* __restrict__ out, long T, int Cy, float res_scale) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  const float* r = num + t * ldn;
  const float f = res_scale / r[Cy];
  for (int c = lane; c < Cy; c += 64) out[t * Cy + c] = x[t * Cy + c] + f * r[c];
}

}  // namespace

extern "C" {

int srhip_prelu_fwd(const float* x, const float* alpha, float* y, long n, void* stream) {
  SR_REQUIRE(x && alpha && y && n > 0 && n % 4 == 0, "prelu_fwd: n = %ld (positive multiple of 4)", n);
  hipLaunchKernelGGL(k_prelu_fwd, dim3(to_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, alpha, y, n / 4);
  SR_LAUNCH_CHECK("prelu_fwd");
  return 0;
}

/* workspace: 4096 doubles.  dx may alias g.  dalpha[0] (+)= sum g * min(x, 0). */
int srhip_prelu_bwd(const float* g, const float* x, const float* alpha, float* dx, float* dalpha, double* workspace,
                    long n, int accumulate, void* stream) {
  SR_REQUIRE(g && x && alpha && dx && dalpha && workspace && n > 0 && n % 4 == 0, "prelu_bwd: n = %ld (positive multiple of 4)", n);
  const int grid = to_grid(n / 4);
  hipLaunchKernelGGL(k_prelu_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, x, alpha, dx, workspace, n / 4);
  hipLaunchKernelGGL(k_prelu_fin, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, grid, dalpha, accumulate);
  SR_LAUNCH_CHECK("prelu_bwd");
  return 0;
}

int srhip_axpby2d(float* y, long ldy, const float* x, long ldx, long rows, int cols, float a, float b, void* stream) {
  SR_REQUIRE(y && x && rows > 0 && cols > 0 && cols % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0,
             "axpby2d: cols, ldy, ldx must be multiples of 4 (cols=%d)", cols);
  hipLaunchKernelGGL(k_axpby2d, dim3(to_grid(rows * (cols / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, x, ldx, rows,
                     cols / 4, a, b);
  SR_LAUNCH_CHECK("axpby2d");
  return 0;
}

int srhip_pad_reflect1(const float* in, float* out, int B, int H, int W, int C, int adjoint, void* stream) {
  SR_REQUIRE(in && out && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "pad_reflect1: H, W >= 2, C a multiple of 4");
  const long n = (long)B * (adjoint ? H : H + 2) * (adjoint ? W : W + 2) * (C / 4);
  hipLaunchKernelGGL(k_pad_reflect1, dim3(to_grid(n)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C / 4, adjoint);
  SR_LAUNCH_CHECK("pad_reflect1");
  return 0;
}

int srhip_crop1(const float* in, float* out, int B, int H, int W, int C, int adjoint, void* stream) {
  SR_REQUIRE(in && out && B > 0 && H >= 1 && W >= 1 && C > 0 && C % 4 == 0, "crop1: C a multiple of 4");
  const long n = (long)B * (adjoint ? H + 2 : H) * (adjoint ? W + 2 : W) * (C / 4);
  hipLaunchKernelGGL(k_crop1, dim3(to_grid(n)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C / 4, adjoint);
  SR_LAUNCH_CHECK("crop1");
  return 0;
}

int srhip_l2norm_rows(float* x, long ld, long T, int C, float eps, float k, void* stream) {
  SR_REQUIRE(x && T > 0 && C > 0 && ld >= C, "l2norm_rows: bad arguments");
  hipLaunchKernelGGL(k_l2norm_rows, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, T, C, eps, k, (float*)nullptr);
  SR_LAUNCH_CHECK("l2norm_rows");
  return 0;
}

int srhip_l2norm_rows_train(float* x, long ld, long T, int C, float eps, float k, float* factors, void* stream) {
  SR_REQUIRE(x && factors && T > 0 && C > 0 && ld >= C && eps > 0.f, "l2norm_rows_train: bad arguments");
  hipLaunchKernelGGL(k_l2norm_rows, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, T, C, eps, k, factors);
  SR_LAUNCH_CHECK("l2norm_rows_train");
  return 0;
}

int srhip_l2norm_rows_bwd(float* dy, long ldd, const float* y, long ldy, const float* factors, long T, int C, float eps, float k,
                          void* stream) {
  SR_REQUIRE(dy && y && factors && T > 0 && C > 0 && ldd >= C && ldy >= C && eps > 0.f, "l2norm_rows_bwd: bad arguments");
  hipLaunchKernelGGL(k_l2norm_rows_bwd, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, dy, ldd, y, ldy, factors, T, C, eps, k);
  SR_LAUNCH_CHECK("l2norm_rows_bwd");
  return 0;
}

int srhip_performer_features_bwd(float* g, const float* f, long n, float ratio_eps, void* stream) {
  SR_REQUIRE(g && f && n > 0 && n % 4 == 0, "performer_features_bwd: n = %ld (positive multiple of 4)", n);
  hipLaunchKernelGGL(k_performer_features_bwd, dim3(sr_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, g, f, n, ratio_eps);
  SR_LAUNCH_CHECK("performer_features_bwd");
  return 0;
}

int srhip_enlca_finish_bwd(const float* dout, const float* num, long ldn, float* dnum, long T, int Cy, float res_scale, void* stream) {
  SR_REQUIRE(dout && num && dnum && T > 0 && Cy > 0 && ldn > Cy, "enlca_finish_bwd: bad arguments");
  hipLaunchKernelGGL(k_enlca_finish_bwd, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, dout, num, ldn, dnum, T, Cy, res_scale);
  SR_LAUNCH_CHECK("enlca_finish_bwd");
  return 0;
}

int srhip_performer_features(float* dash, long ldd, const float* data, long ldx, long T, int F, int C, float ratio, float eps,
                             void* stream) {
  SR_REQUIRE(dash && data && T > 0 && F > 0 && C > 0 && ldd >= F && ldx >= C, "performer_features: bad arguments");
  hipLaunchKernelGGL(k_performer_features, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, dash, ldd, data, ldx, T, F, C,
                     ratio, eps);
  SR_LAUNCH_CHECK("performer_features");
  return 0;
}

int srhip_enlca_finish(const float* num, long ldn, const float* x, float* out, long T, int Cy, float res_scale, void* stream) {
  SR_REQUIRE(num && x && out && T > 0 && Cy > 0 && ldn > Cy, "enlca_finish: bad arguments");
  hipLaunchKernelGGL(k_enlca_finish, dim3(sr_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, num, ldn, x, out, T, Cy, res_scale);
  SR_LAUNCH_CHECK("enlca_finish");
  return 0;
}

}  // extern "C"
