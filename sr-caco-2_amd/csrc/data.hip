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

}  // namespace

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
