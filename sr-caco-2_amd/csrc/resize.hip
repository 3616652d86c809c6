// cv2.resize(..., interpolation=cv2.INTER_CUBIC) on 1-channel images: the 'l_to_h_img' tensors of the reference's
// dataset (dlib/datasets/dataset_dpsr.py:659-683 _resize_low_to_scale: uint8 tiles at construction / evaluation,
// :813-821,:836,:905-906 float32 patches after the LR-only augmentations), consumed by the SRCNN-style nets
// (model_plain.py:184-195).
//
// cv2 is not in this image and its source is not under /root/reference: this restates OpenCV's published algorithm
// (imgproc/resize.cpp: interpolateCubic with A = -0.75; pixel centre mapping fx = (dx + 0.5) * scale - 0.5; border
// replicate; uint8 images in fixed point -- coefficients rounded to 1/2048 (INTER_RESIZE_COEF_BITS = 11), horizontal
// pass in int, vertical pass (sum + 2^21) >> 22 saturated; float32 images in plain float arithmetic).  PARITY UNPINNED
// against cv2 itself (OpenCV's vectorised vertical pass rounds a float sum instead of the integer one: on rare
// near-ties it may differ by one grey level); pinned bit-exact (uint8) / 1e-6 (float32) against oracle/cv2_cubic.py,
// the same restatement in numpy.
#include "common.h"
#include "../../include/srhip.h"

namespace {

// identical float arithmetic on host and device: no FMA contraction in the coefficient formulas
#pragma clang fp contract(off)
__host__ __device__ inline void cubic_coeffs(float x, float (&c)[4]) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

template <bool U8>
__global__ void __launch_bounds__(256) k_resize_cubic(const void* __restrict__ srcv, void* __restrict__ dstv, int B, int H,
                                                      int W, int Ho, int Wo, double scale_y, double scale_x) {
#pragma clang fp contract(off)
  const long n = (long)B * Ho * Wo;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int dx = (int)(i % Wo), dy = (int)((i / Wo) % Ho);
    const long b = i / ((long)Wo * Ho);
    float fx = (float)((dx + 0.5) * scale_x - 0.5), fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= sx; fy -= sy;
    float cx[4], cy[4];
    cubic_coeffs(fx, cx);
    cubic_coeffs(fy, cy);
    int xs[4], ys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xs[k] = min(max(sx - 1 + k, 0), W - 1);
      ys[k] = min(max(sy - 1 + k, 0), H - 1);
    }
    if (U8) {
      const unsigned char* src = (const unsigned char*)srcv + b * H * W;
      int ax[4], ay[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {       // saturate_cast<short>(c * 2048): round to nearest even, as cvRound
        ax[k] = (int)rintf(cx[k] * 2048.f);
        ay[k] = (int)rintf(cy[k] * 2048.f);
      }
      int acc = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned char* row = src + (long)ys[r] * W;
        const int h = row[xs[0]] * ax[0] + row[xs[1]] * ax[1] + row[xs[2]] * ax[2] + row[xs[3]] * ax[3];
        acc += h * ay[r];
      }
      const int v = (acc + (1 << 21)) >> 22;
      ((unsigned char*)dstv)[i] = (unsigned char)min(max(v, 0), 255);
    } else {
      const float* src = (const float*)srcv + b * H * W;
      float hr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* row = src + (long)ys[r] * W;
        hr[r] = row[xs[0]] * cx[0] + row[xs[1]] * cx[1] + row[xs[2]] * cx[2] + row[xs[3]] * cx[3];
      }
      ((float*)dstv)[i] = hr[0] * cy[0] + hr[1] * cy[1] + hr[2] * cy[2] + hr[3] * cy[3];
    }
  }
}

// uint8 -> float32 / 255 (util.uint2single, utils_image.py:322-323) and clip to [0, 1], in place variants used around the resize
__global__ void __launch_bounds__(256) k_u8_to_unit(const unsigned char* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)((double)src[i] / 255.0);
}
__global__ void __launch_bounds__(256) k_clip01(float* __restrict__ x, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}

inline int rs_grid(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int srhip_resize_cubic(const void* src, void* dst, int is_u8, int B, int H, int W, int Ho, int Wo, void* stream) {
  SR_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize_cubic: empty image / NULL argument");
  const double sy = (double)H / Ho, sx = (double)W / Wo;      // scale = 1 / inv_scale, inv_scale = dsize / ssize (resize.cpp)
  const double scale_y = 1.0 / ((double)Ho / H), scale_x = 1.0 / ((double)Wo / W);
  (void)sy; (void)sx;
  const long n = (long)B * Ho * Wo;
  if (is_u8) hipLaunchKernelGGL(k_resize_cubic<true>, dim3(rs_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, Ho, Wo, scale_y, scale_x);
  else hipLaunchKernelGGL(k_resize_cubic<false>, dim3(rs_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, Ho, Wo, scale_y, scale_x);
  SR_LAUNCH_CHECK("resize_cubic");
  return 0;
}

int srhip_u8_to_unit(const unsigned char* src, float* dst, long n, void* stream) {
  SR_REQUIRE(src && dst && n > 0, "u8_to_unit: empty");
  hipLaunchKernelGGL(k_u8_to_unit, dim3(rs_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  SR_LAUNCH_CHECK("u8_to_unit");
  return 0;
}

int srhip_clip01(float* x, long n, void* stream) {
  SR_REQUIRE(x && n > 0, "clip01: empty");
  hipLaunchKernelGGL(k_clip01, dim3(rs_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n);
  SR_LAUNCH_CHECK("clip01");
  return 0;
}

}  // extern "C"
