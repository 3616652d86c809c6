// Input-pipeline row f2: the device-side tail of DatasetDPSR.__getitem__ for a training batch --
// crop a P x P patch out of a resident uint8 tile, apply one of the 8 flip / rotate
// augmentations, convert to float32 in [0, 1] and lay the batch out as [B][1][P][P]
// (dlib/datasets/dataset_dpsr.py:866-894,914-915 with utils_image.py:322-323 uint2single,
// :381-382 single2tensor3, :469-487 augment_img).  Index-only + one exact conversion:
// bit-exact against the reference (np.float32(v / 255.): the division is done in float64).
#include "common.h"
#include "../../include/srhip.h"

namespace {

constexpr int MAXJOBS = 64;
struct PatchJobs {
  struct J { const unsigned char* img; int W, y0, x0, mode; } j[MAXJOBS];
};

// source position inside the cropped patch for output position (i, j), augment_img mode m
// (utils_image.py:469-487; rot90 = counter-clockwise, flipud = rows reversed):
//   0 identity   1 transpose   2 flipud   3 rot90 x3   4 fliplr   5 rot90   6 rot180   7 anti-transpose
__device__ __forceinline__ void aug_src(int m, int i, int j, int P, int& si, int& sj) {
  const int e = P - 1;
  switch (m) {
    case 0: si = i; sj = j; break;
    case 1: si = j; sj = i; break;
    case 2: si = e - i; sj = j; break;
    case 3: si = e - j; sj = i; break;
    case 4: si = i; sj = e - j; break;
    case 5: si = j; sj = e - i; break;
    case 6: si = e - i; sj = e - j; break;
    default: si = e - j; sj = e - i; break;
  }
}

__global__ void __launch_bounds__(256) k_patch_gather(PatchJobs jobs, int P, float* __restrict__ out) {
  const PatchJobs::J J = jobs.j[blockIdx.y];
  float* o = out + (long)blockIdx.y * P * P;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < P * P; idx += gridDim.x * 256) {
    const int i = idx / P, j = idx - i * P;
    int si, sj;
    aug_src(J.mode, i, j, P, si, sj);
    const unsigned char v = J.img[(long)(J.y0 + si) * J.W + J.x0 + sj];
    o[idx] = (float)((double)v / 255.0);
  }
}

// ROI-weighted patch origins (PatchSampler._roi, dataset_dpsr.py:330-369): over the (H-P) x (W-P)
// candidate origins, origin (r, c) has weight exp(5 * roi) + 1 with roi = img[r + P/2][c + P/2] >= th,
// i.e. e^5 + 1 on the region of interest and 2 elsewhere; the reference draws one multinomial
// sample from the normalised weights.  Here: inverse CDF of the SAME probabilities in row-major
// order from one uniform per patch -- cum(i) = W1 * (#roi origins <= i) + W0 * (#others <= i) in
// fp64, the origin is the first i with cum(i) > u * cum(last).  One block per patch: row counts
// in parallel, then one lane walks the row prefix and the chosen row.
constexpr double ROI_W1 = 149.4131591025766, ROI_W0 = 2.0;      // exp(5) + 1, exp(0) + 1
struct RoiJobs {
  struct J { const unsigned char* img; int H, W; } j[MAXJOBS];
};
__global__ void __launch_bounds__(256) k_roi_sample(RoiJobs jobs, int P, int th, const double* __restrict__ u,
                                                    int* __restrict__ rowcnt, int max_rows,
                                                    int* __restrict__ origin) {
  const RoiJobs::J J = jobs.j[blockIdx.x];
  const int Hc = J.H - P, Wc = J.W - P, half = P / 2;
  int* cnt = rowcnt + (long)blockIdx.x * max_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < Hc; r += 4) {                      // one wave per row
    const unsigned char* row = J.img + (long)(r + half) * J.W + half;
    int n = 0;
    for (int c = lane; c < Wc; c += 64) n += row[c] >= th;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) cnt[r] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long n1 = 0;
    for (int r = 0; r < Hc; ++r) n1 += cnt[r];
    const long n0 = (long)Hc * Wc - n1;
    const double target = u[blockIdx.x] * (ROI_W1 * (double)n1 + ROI_W0 * (double)n0);
    long c1 = 0, c0 = 0;
    int R = Hc - 1;
    for (int r = 0; r < Hc; ++r) {                           // first row whose cumulative weight exceeds the target
      const long a1 = c1 + cnt[r], a0 = c0 + (Wc - cnt[r]);
      if (ROI_W1 * (double)a1 + ROI_W0 * (double)a0 > target) { R = r; break; }
      c1 = a1; c0 = a0;
    }
    const unsigned char* row = J.img + (long)(R + half) * J.W + half;
    int C = Wc - 1;
    for (int c = 0; c < Wc; ++c) {
      if (row[c] >= th) ++c1; else ++c0;
      if (ROI_W1 * (double)c1 + ROI_W0 * (double)c0 > target) { C = c; break; }
    }
    origin[2 * blockIdx.x] = R;
    origin[2 * blockIdx.x + 1] = C;
  }
}

// Patch matrix of a 1-channel image for a k x k / stride 1 / pad k/2 convolution: out[t][dy*k+dx] =
// x[b][y+dy-k/2][x+dx-k/2] (zero outside), columns k*k .. ldo-1 zero.  Turns the wide first layer of
// SRCNN (nn.Conv2d(1, 1024, 5, 1, 2), network_srcnn.py:33) into a [T x 28] . [1024 x 28]^T GEMM on the
// bf16x3 kernels; its weight gradient is the TN contraction against the same matrix.
__global__ void __launch_bounds__(256) k_im2col_c1(const float* __restrict__ x, float* __restrict__ out, long ldo,
                                                   int B, int H, int W, int ks) {
  const long n = (long)B * H * W * ldo;
  const int half = ks / 2, kk = ks * ks;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long t = i / ldo;
    const int j = (int)(i - t * ldo);
    float v = 0.f;
    if (j < kk) {
      const int xx = (int)(t % W), yy = (int)((t / W) % H);
      const long b = t / ((long)W * H);
      const int sy = yy + j / ks - half, sx = xx + j % ks - half;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[(b * H + sy) * W + sx];
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" int srhip_im2col_c1(const float* x, float* out, long ldo, int B, int H, int W, int ksize, void* stream) {
  SR_REQUIRE(x && out && B > 0 && H > 0 && W > 0, "im2col_c1: empty image / NULL argument");
  SR_REQUIRE(ksize >= 1 && ksize % 2 == 1 && ksize * ksize <= ldo && ldo % 4 == 0,
             "im2col_c1: odd kernel size with ksize^2 <= ldo, ldo %% 4 == 0 (ksize=%d ldo=%ld)", ksize, ldo);
  const long n = (long)B * H * W * ldo;
  long g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_im2col_c1, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, out, ldo, B, H, W, ksize);
  SR_LAUNCH_CHECK("im2col_c1");
  return 0;
}

extern "C" long srhip_roi_sample_ws(int B, int max_rows) { return (long)B * max_rows; }

extern "C" int srhip_roi_sample(const srhip_patch_job* jobs, int B, int P, int threshold, const double* uniforms,
                                int* workspace, int max_rows, int* origins, void* stream) {
  SR_REQUIRE(B > 0 && P > 0 && uniforms && workspace && origins, "roi_sample: empty batch / NULL argument");
  hipStream_t st = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += MAXJOBS) {
    const int nb = B - b0 < MAXJOBS ? B - b0 : MAXJOBS;
    RoiJobs rj;
    for (int b = 0; b < nb; ++b) {
      const srhip_patch_job& q = jobs[b0 + b];
      SR_REQUIRE(q.img != nullptr && q.H > P && q.W > P, "roi_sample: job %d: tile %dx%d must exceed the patch %d",
                 b0 + b, q.H, q.W, P);
      SR_REQUIRE(q.H - P <= max_rows, "roi_sample: job %d: %d candidate rows > workspace rows %d", b0 + b, q.H - P, max_rows);
      rj.j[b].img = q.img; rj.j[b].H = q.H; rj.j[b].W = q.W;
    }
    hipLaunchKernelGGL(k_roi_sample, dim3(nb), dim3(256), 0, st, rj, P, threshold, uniforms + b0,
                       workspace + (long)b0 * max_rows, max_rows, origins + 2 * b0);
  }
  SR_LAUNCH_CHECK("roi_sample");
  return 0;
}

extern "C" int srhip_patch_gather(const srhip_patch_job* jobs, int B, int P, float* out, void* stream) {
  SR_REQUIRE(B > 0 && P > 0, "patch_gather: empty batch");
  for (int b = 0; b < B; ++b) {
    const srhip_patch_job& q = jobs[b];
    SR_REQUIRE(q.img != nullptr, "patch_gather: job %d has no image", b);
    SR_REQUIRE(q.mode >= 0 && q.mode <= 7, "patch_gather: job %d: augmentation mode %d (0..7)", b, q.mode);
    SR_REQUIRE(q.y0 >= 0 && q.x0 >= 0 && q.y0 + P <= q.H && q.x0 + P <= q.W,
               "patch_gather: job %d: crop (%d,%d)+%d outside the %dx%d tile", b, q.y0, q.x0, P, q.H, q.W);
  }
  hipStream_t st = (hipStream_t)stream;
  for (int b0 = 0; b0 < B; b0 += MAXJOBS) {
    const int nb = B - b0 < MAXJOBS ? B - b0 : MAXJOBS;
    PatchJobs pj;
    for (int b = 0; b < nb; ++b) {
      const srhip_patch_job& q = jobs[b0 + b];
      pj.j[b].img = q.img; pj.j[b].W = q.W; pj.j[b].y0 = q.y0; pj.j[b].x0 = q.x0; pj.j[b].mode = q.mode;
    }
    const int gx = sr_cdiv((long)P * P, 256) < 64 ? sr_cdiv((long)P * P, 256) : 64;
    hipLaunchKernelGGL(k_patch_gather, dim3(gx, nb), dim3(256), 0, st, pj, P, out + (long)b0 * P * P);
  }
  SR_LAUNCH_CHECK("patch_gather");
  return 0;
}
