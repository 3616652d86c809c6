// Optional MasterLoss terms beyond L1 / L2 / SSIM: the point-wise Charbonnier
// and L2Sum terms and the local-variation family -- ImageGradientLoss, LaplacianFilterLoss,
// LocalVariationLoss and their Norm* variants (dlib/loss/main.py:102-151,328-674 with the
// operators of dlib/loss/local_variations.py:18-141).  Each term is ONE kernel that produces
// the value and d loss / d pred together (+ the shared partial-sum reducer).
//
// Local-variation terms: op = K 3x3 .. 7x7 stencils on the replicate-padded 1-channel image,
//   plain:  loss = lam * mean_{b,k,y,x} nrm(op_k(pred) - op_k(target))
//   Norm*:  loss = lam * mean_{b,y,x}   nrm(|op(pred)|_2 - |op(target)|_2)   (2-norm over k)
// nrm = square (NORM2) or abs (NORM1).  HBM-bound: pred and target are read once (a tile with
// a 2R halo in LDS), the gradient is written once.  The gradient is a GATHER -- for output
// pixel q and tap (k, off, w) it sums g_k(p) over the source pixels p whose padded read
// clamp(p + off) lands on q (one pixel in the interior, a short run on the image border) --
// so there are no atomics and the result is deterministic.
#include "common.h"
#include "kernels.h"
#include "../../include/srhip.h"

namespace {

constexpr int TS = 16;     // output tile edge (256 threads, one pixel each)

// point-wise terms: mode 0 L1 (optional weight), 1 L2, 2 Charbonnier sqrt(e^2 + eps), 3 L2Sum
__global__ void __launch_bounds__(256) k_loss_pointwise(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ wgt, float* __restrict__ grad,
                                                        double* __restrict__ part, long n, int mode, float gs,
                                                        float eps, int grad_accum) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float e = pred[i] - tgt[i];
    float v, g;
    if (mode == 0) {
      v = fabsf(e);
      g = (e > 0.f) ? gs : (e < 0.f ? -gs : 0.f);
      if (wgt) { v *= wgt[i]; g *= wgt[i]; }
    } else if (mode == 2) {
      v = sqrtf(e * e + eps);
      g = gs * e / v;
    } else {
      v = e * e;
      g = 2.f * gs * e;
    }
    acc += (double)v;
    if (grad) grad[i] = grad_accum ? grad[i] + g : g;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// BoundedPrediction (dlib/loss/main.py:189-237) through the extended log barrier
// (dlib/losses/elb.py:92-122):  y - eps <= y_hat <= y + eps  as two inequality constraints
//   z_r = y_hat - (y + eps) <= 0,  z_l = (y - eps) - y_hat <= 0   (both scaled by color_max if restore_range)
//   elb(z) = -(1/t) log(-z)                        for z <= -1/t^2
//          =  t z - (1/t) log(1/t^2) + 1/t         otherwise
// loss = lam * (mean elb(z_r) + mean elb(z_l)) / 2.  The constants follow the reference's f32 tensor
// arithmetic (ct = -(1/t^2); 1/t; log(1/t^2)).
__global__ void __launch_bounds__(256) k_loss_bounded(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      float* __restrict__ grad, double* __restrict__ part, long n,
                                                      float eps, float t, float scale, float gs, int grad_accum) {
  __shared__ double sh[4];
  const float t2 = t * t, inv_t2 = 1.f / t2, ct = -inv_t2, inv_t = 1.f / t;
  const float c_great = inv_t * logf(inv_t2);
  double acc = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float yh = pred[i] * scale, y = tgt[i] * scale;
    const float zr = yh - (y + eps), zl = y - eps - yh;
    float v = 0.f, g = 0.f;
    if (zr <= ct) { v += -inv_t * logf(-zr); g += -inv_t / zr; }
    else { v += t * zr - c_great + inv_t; g += t; }
    if (zl <= ct) { v += -inv_t * logf(-zl); g -= -inv_t / zl; }
    else { v += t * zl - c_great + inv_t; g -= t; }
    acc += (double)v;
    if (grad) grad[i] = grad_accum ? grad[i] + gs * g : gs * g;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// WeightsSparsityLoss (dlib/loss/main.py:938-959): lam * sum |w| over the (flat) parameters;
// d/dw = lam * sign(w), ACCUMULATED into the parameter gradients.
__global__ void __launch_bounds__(256) k_l1_sparsity(const float* __restrict__ w, float* __restrict__ grad,
                                                     double* __restrict__ part, long n, float lam) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = w[i];
    acc += (double)fabsf(v);
    if (grad) grad[i] += v > 0.f ? lam : (v < 0.f ? -lam : 0.f);
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
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

// source pixels p in [0, n) whose replicate-padded read p + off lands on q
__device__ __forceinline__ void src_range(int q, int off, int n, int& lo, int& hi) {
  lo = hi = q - off;
  if (q == 0) lo = 0;
  if (q == n - 1) hi = n - 1;
  lo = max(lo, 0);
  hi = min(hi, n - 1);
}

template <int OP, int R> struct Stencil;
// image gradient: k0 = x(y, x+1) - x(y, x-1), k1 = x(y+1, x) - x(y-1, x)   (local_variations.py:18-55)
template <> struct Stencil<0, 1> {
  static constexpr int K = 2, TAPS = 2;
  static __device__ __forceinline__ void tap(int k, int t, int& dy, int& dx, float& w) {
    const int s = t ? -1 : 1;
    dy = k ? s : 0; dx = k ? 0 : s; w = (float)s;
  }
};
// Laplacian: 8 x(p) - the 8 neighbours                                     (local_variations.py:58-91)
template <> struct Stencil<1, 1> {
  static constexpr int K = 1, TAPS = 9;
  static __device__ __forceinline__ void tap(int, int t, int& dy, int& dx, float& w) {
    dy = t / 3 - 1; dx = t % 3 - 1; w = (t == 4) ? 8.f : -1.f;
  }
};
// local variation: k = (i, j) != centre of a ksz x ksz window: x(p) - x(p + (i-c, j-c))   (:94-141)
template <int R> struct Stencil<2, R> {
  static constexpr int KS = 2 * R + 1, K = KS * KS - 1, TAPS = 2;
  static __device__ __forceinline__ void tap(int k, int t, int& dy, int& dx, float& w) {
    const int kk = k + (k >= K / 2);        // skip the centre
    dy = t ? kk / KS - R : 0; dx = t ? kk % KS - R : 0; w = t ? -1.f : 1.f;
  }
};

// r_k at LDS position (ly, lx) of image a (row pitch LW).  Explicit fma everywhere: pred and
// target must go through the SAME operation sequence -- where they are equal the reference's
// difference is exactly 0 and NORM1's sign(0) = 0 (with compiler-chosen contraction the two
// sides differed by an ulp and the L1 gradient there became +-1).
template <int OP, int R>
__device__ __forceinline__ float stencil_at(const float* a, int LW, int ly, int lx, int k) {
  typedef Stencil<OP, R> S;
  float r = 0.f;
#pragma unroll
  for (int t = 0; t < S::TAPS; ++t) {
    int dy, dx; float w;
    S::tap(k, t, dy, dx, w);
    r = __builtin_fmaf(w, a[(ly + dy) * LW + lx + dx], r);
  }
  return r;
}

__device__ __forceinline__ float nrm_val(float e, int l1) { return l1 ? fabsf(e) : e * e; }
__device__ __forceinline__ float nrm_der(float e, int l1) {
  return l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.f * e;
}

template <int OP, int R>
__global__ void __launch_bounds__(256) k_loss_stencil(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      float* __restrict__ grad, double* __restrict__ part, int H, int W,
                                                      int l1, int chan_norm, float gs, int grad_accum) {
  typedef Stencil<OP, R> S;
  constexpr int LW = TS + 4 * R;            // tile + 2R halo
  constexpr int CW = TS + 2 * R;            // tile + R halo (coefficients of the Norm* variants)
  __shared__ float sp[LW * LW], st[LW * LW], sd[LW * LW], coef[CW * CW];
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
  const float* P = pred + (long)b * H * W;
  const float* T = tgt + (long)b * H * W;
  for (int i = tid; i < LW * LW; i += 256) {
    const int ly = i / LW, lx = i - ly * LW;
    const int y = min(max(ty0 + ly - 2 * R, 0), H - 1), x = min(max(tx0 + lx - 2 * R, 0), W - 1);   // replicate pad
    const float a = P[(long)y * W + x], c = T[(long)y * W + x];
    sp[i] = a;
    st[i] = c;
    sd[i] = a - c;
  }
  // block-uniform: every pixel within R of the tile is at least R (>= 1) away from the image border
  const bool interior = ty0 >= 2 * R && ty0 + TS + 2 * R <= H && tx0 >= 2 * R && tx0 + TS + 2 * R <= W;
  __syncthreads();
  if (chan_norm) {
    // c(p) = nrm'(|op(pred)| - |op(target)|) / |op(pred)|   (0 where the norm is 0, as torch's norm backward)
    for (int i = tid; i < CW * CW; i += 256) {
      const int cy = i / CW, cx = i - cy * CW;
      const int y = ty0 + cy - R, x = tx0 + cx - R;
      float c = 0.f;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        float np = 0.f, nt = 0.f;
#pragma unroll
        for (int k = 0; k < S::K; ++k) {
          const float a = stencil_at<OP, R>(sp, LW, cy + R, cx + R, k), t = stencil_at<OP, R>(st, LW, cy + R, cx + R, k);
          np = __builtin_fmaf(a, a, np); nt = __builtin_fmaf(t, t, nt);
        }
        np = sqrtf(np); nt = sqrtf(nt);
        c = np > 0.f ? nrm_der(np - nt, l1) / np : 0.f;
      }
      coef[i] = c;
    }
    __syncthreads();
  }
  const int qy = tid / TS, qx = tid % TS;
  const int y = ty0 + qy, x = tx0 + qx;
  double val = 0.0;
  if (y < H && x < W) {
    const int ly = qy + 2 * R, lx = qx + 2 * R;
    // value at q
    if (chan_norm) {
      float np = 0.f, nt = 0.f;
#pragma unroll
      for (int k = 0; k < S::K; ++k) {
        const float a = stencil_at<OP, R>(sp, LW, ly, lx, k), t = stencil_at<OP, R>(st, LW, ly, lx, k);
        np = __builtin_fmaf(a, a, np); nt = __builtin_fmaf(t, t, nt);
      }
      val = (double)nrm_val(sqrtf(np) - sqrtf(nt), l1);
    } else {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < S::K; ++k)
        v += nrm_val(stencil_at<OP, R>(sp, LW, ly, lx, k) - stencil_at<OP, R>(st, LW, ly, lx, k), l1);
      val = (double)v;
    }
    // gradient at q: gather over taps and the source pixels whose padded read lands on q
    if (grad && interior) {
      // no pixel of this tile touches the padding: the one source pixel of tap off is q - off;
      // compile-time taps, straight-line code.  Plain terms are linear in d = pred - target.
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < S::K; ++k) {
#pragma unroll
        for (int t = 0; t < S::TAPS; ++t) {
          int dy, dx; float w;
          S::tap(k, t, dy, dx, w);
          const int ply = ly - dy, plx = lx - dx;
          float gk;
          if (chan_norm) gk = coef[(ply - R) * CW + plx - R] * stencil_at<OP, R>(sp, LW, ply, plx, k);
          else gk = nrm_der(stencil_at<OP, R>(sd, LW, ply, plx, k), l1);
          g = __builtin_fmaf(w, gk, g);
        }
      }
      const long o = ((long)b * H + y) * W + x;
      grad[o] = grad_accum ? grad[o] + gs * g : gs * g;
    } else if (grad) {
      float g = 0.f;
      for (int k = 0; k < S::K; ++k) {
#pragma unroll
        for (int t = 0; t < S::TAPS; ++t) {
          int dy, dx; float w;
          S::tap(k, t, dy, dx, w);
          int ylo, yhi, xlo, xhi;
          src_range(y, dy, H, ylo, yhi);
          src_range(x, dx, W, xlo, xhi);
          for (int py = ylo; py <= yhi; ++py)
            for (int px = xlo; px <= xhi; ++px) {
              const int ply = py - ty0 + 2 * R, plx = px - tx0 + 2 * R;
              const float rp = stencil_at<OP, R>(sp, LW, ply, plx, k);
              float gk;
              if (chan_norm) gk = coef[(ply - R) * CW + plx - R] * rp;
              else gk = nrm_der(rp - stencil_at<OP, R>(st, LW, ply, plx, k), l1);
              g += w * gk;
            }
        }
      }
      const long o = ((long)b * H + y) * W + x;
      grad[o] = grad_accum ? grad[o] + gs * g : gs * g;
    }
  }
  val = wave_sum_d(val);
  if ((tid & 63) == 0) sh[tid >> 6] = val;
  __syncthreads();
  if (tid == 0) part[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// LocalMoments (dlib/loss/main.py:240-325 with PatchMoments, dlib/loss/local_terms.py:14-66): per pixel
// the mean / unbiased variance of its 3x3 patch (reflect padding) for pred (m, v) and target (mt, vt);
//   kl = log(sqrt(v + 1) / sqrt(vt + 1)) + (vt + 1 + (mt - m)^2) / (2 (v + 1)) - 1/2
// counted ONLY where the target patch is exactly flat (vt == 0); loss = lam * mean_{b,y,x}(kl * [vt == 0]).
// Mean and variance by Welford's update in row-major patch order: a flat patch gives exactly 0 for any
// value (sum-then-divide does not), which is what the reference's equality test relies on.
// The gradient is a gather over the 3x3 centres p around q with the multiplicity of q in p's reflected patch.
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ void __launch_bounds__(256) k_loss_local_moments(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            float* __restrict__ grad, double* __restrict__ part, int H,
                                                            int W, float gs, int grad_accum) {
  constexpr int R = 1, LW = TS + 4 * R, CW = TS + 2 * R;
  __shared__ float sp[LW * LW], st[LW * LW];
  __shared__ float cm[CW * CW], cdm[CW * CW], cdv[CW * CW];      // per centre: mean(pred), dL/dm, dL/dv
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
  const float* P = pred + (long)b * H * W;
  const float* T = tgt + (long)b * H * W;
  for (int i = tid; i < LW * LW; i += 256) {
    const int ly = i / LW, lx = i - ly * LW;
    // reflect (H, W >= 2); positions further out than one reflection are never read
    const int y = min(max(reflect1(ty0 + ly - 2 * R, H), 0), H - 1), x = min(max(reflect1(tx0 + lx - 2 * R, W), 0), W - 1);
    sp[i] = P[(long)y * W + x];
    st[i] = T[(long)y * W + x];
  }
  __syncthreads();
  double val = 0.0;
  for (int i = tid; i < CW * CW; i += 256) {
    const int cy = i / CW, cx = i - cy * CW;
    const int y = ty0 + cy - R, x = tx0 + cx - R;
    float m = 0.f, dm = 0.f, dv = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      float mt = 0.f, m2 = 0.f, m2t = 0.f;
      int k = 0;
#pragma unroll
      for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
          ++k;
          const float a = sp[(cy + R + oy) * LW + cx + R + ox], c = st[(cy + R + oy) * LW + cx + R + ox];
          const float d1 = a - m;  m += d1 / (float)k;  m2 += d1 * (a - m);
          const float e1 = c - mt; mt += e1 / (float)k; m2t += e1 * (c - mt);
        }
      const float v = m2 / 8.f, vt = m2t / 8.f;
      if (vt == 0.f) {
        const float sv = v + 1.f, tv = vt + 1.f, dmu = mt - m;
        const float kl = logf(sqrtf(sv) / sqrtf(tv)) + (tv + dmu * dmu) / (2.f * sv) - 0.5f;
        dm = -dmu / sv;
        dv = 0.5f / sv - (tv + dmu * dmu) / (2.f * sv * sv);
        // the value counts once: for centres inside the tile proper
        if (cy >= R && cy < TS + R && cx >= R && cx < TS + R) val += (double)kl;
      }
    }
    cm[i] = m; cdm[i] = dm; cdv[i] = dv;
  }
  __syncthreads();
  const int qy = tid / TS, qx = tid % TS;
  const int y = ty0 + qy, x = tx0 + qx;
  if (grad && y < H && x < W) {
    const float xq = sp[(qy + 2 * R) * LW + qx + 2 * R];
    float g = 0.f;
#pragma unroll
    for (int py = -1; py <= 1; ++py) {
      const int yy = y + py;
      if (yy < 0 || yy >= H) continue;
      int cyn = 0;                                   // how often row y appears in the reflected patch rows of yy
#pragma unroll
      for (int o = -1; o <= 1; ++o) cyn += reflect1(yy + o, H) == y;
#pragma unroll
      for (int px = -1; px <= 1; ++px) {
        const int xx = x + px;
        if (xx < 0 || xx >= W) continue;
        int cxn = 0;
#pragma unroll
        for (int o = -1; o <= 1; ++o) cxn += reflect1(xx + o, W) == x;
        const int ci = (qy + R + py) * CW + qx + R + px;
        g += (float)(cyn * cxn) * (cdm[ci] * (1.f / 9.f) + cdv[ci] * (2.f / 8.f) * (xq - cm[ci]));
      }
    }
    const long o = ((long)b * H + y) * W + x;
    grad[o] = grad_accum ? grad[o] + gs * g : gs * g;
  }
  val = wave_sum_d(val);
  if ((tid & 63) == 0) sh[tid >> 6] = val;
  __syncthreads();
  if (tid == 0) part[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// HistogramMatch (dlib/loss/main.py:690-782) with SoftHistogram (dlib/loss/global_terms.py:17-72):
//   h[b][k] = sum_px sigmoid(s (x - c_k + d/2)) - sigmoid(s (x - c_k - d/2)),  c_k = d (k + 1/2), d = 1 / bins
//   p = (h + 1) / sum_k (h + 1)  for pred and target;  loss = lam * mean_{b,k} nrm(p_pred - p_target)
// The reference evaluates all bins for every pixel (b x bins x n sigmoid pairs).  In float32 a pair is
// exactly 0 once both arguments are beyond +17 (both sigmoids round to 1) or below -104 (both underflow
// to 0), i.e. for every bin further than RB = ceil(104 / (s d)) + 1 bins from the pixel's own -- with the
// default s = 1e5 only the neighbours -- so only those are visited: the same sums, 3 x 2 sigmoids per
// pixel instead of 256 x 2.
constexpr int HB = 64;                 // blocks per image
__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + expf(-z)); }

// MODE 0: soft histogram (above).  MODE 1: Gaussian KDE of KDEMatch (dlib/loss/global_terms.py:75-152):
//   p[b][k] = mean_px c1 exp(-(x - cs_k)^2 / c2),  cs = linspace(0, 1, bins), c1 = (2 pi bw)^-1/2, c2 = 2 bw
// (delta = 1 / (bins - 1), half = c2, sigma = c1 here); exp underflows to exactly 0 beyond
// |x - cs_k| > sqrt(104 c2), which bounds the bins a pixel touches the same way.
template <int MODE>
__device__ __forceinline__ float bin_center(int k, float delta, int bins) {
  if (MODE == 0) return delta * ((float)k + 0.5f);
  // torch.linspace(0, 1, bins): from the start below the middle, from the end above it
  return k < bins / 2 ? delta * (float)k : 1.f - delta * (float)(bins - 1 - k);
}
template <int MODE>
__device__ __forceinline__ int own_bin(float x, float delta, int bins) {
  return min(max((int)(MODE == 0 ? x / delta : x / delta + 0.5f), 0), bins - 1);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_soft_hist(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                   float* __restrict__ part, long n, int bins, float delta,
                                                   float half, float sigma, int rb) {
  extern __shared__ float hist[];                 // [2][bins]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * bins; i += 256) hist[i] = 0.f;
  __syncthreads();
  const long per = (n + HB - 1) / HB, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float x = (a ? tgt : pred)[(long)b * n + i];
      const int k0 = own_bin<MODE>(x, delta, bins);
      for (int k = max(k0 - rb, 0); k <= min(k0 + rb, bins - 1); ++k) {
        const float d = x - bin_center<MODE>(k, delta, bins);
        const float v = MODE == 0 ? sigmoid_f(sigma * (d + half)) - sigmoid_f(sigma * (d - half))
                                  : sigma * expf(-(d * d) / half);
        if (v != 0.f) atomicAdd(&hist[a * bins + k], v);
      }
    }
  }
  __syncthreads();
  float* o = part + ((long)b * HB + blockIdx.x) * 2 * bins;
  for (int i = threadIdx.x; i < 2 * bins; i += 256) o[i] = hist[i];
}

// one block per image: histograms, the metric's loss share and dL/dh(pred)
// kde > 0: KDEMatch's form -- p = mean over the kde pixels + 1e-4 on both sides, no normalisation
// kind 1 / 2: |P - Q| / (P - Q)^2 per bin;  3: KL, Q (log Q - log P) per bin (nn.KLDivLoss(batchmean)(log P, Q),
// dlib/loss/main.py:727-729,771-773);  4: Bhattacharyya through the extended log barrier, elb(-sum_k sqrt(P Q)) per image
// (main.py:677-687,775-777,891-892; dlib/losses/elb.py:92-122) with barrier parameter t
__global__ void __launch_bounds__(256) k_hist_finalize(const float* __restrict__ part, float* __restrict__ dldh,
                                                       double* __restrict__ lpart, int bins, int kind, float gs,
                                                       float kde, float t) {
  __shared__ float red[2][4];
  __shared__ float bc[3];
  const int b = blockIdx.x, k = threadIdx.x;
  float hp = 0.f, ht = 0.f;
  if (k < bins) {
    for (int j = 0; j < HB; ++j) {
      const float* o = part + ((long)b * HB + j) * 2 * bins;
      hp += o[k]; ht += o[bins + k];
    }
    if (kde > 0.f) { hp = hp / kde + 1e-4f; ht = ht / kde + 1e-4f; }
    else { hp += 1.f; ht += 1.f; }
  }
  float s1 = wave_sum(k < bins ? hp : 0.f), s2 = wave_sum(k < bins ? ht : 0.f);
  if ((k & 63) == 0) { red[0][k >> 6] = s1; red[1][k >> 6] = s2; }
  __syncthreads();
  const float Sp = red[0][0] + red[0][1] + red[0][2] + red[0][3], St = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  __syncthreads();
  const float P = k < bins ? (kde > 0.f ? hp : hp / Sp) : 0.f, Q = k < bins ? (kde > 0.f ? ht : ht / St) : 0.f;
  float val = 0.f, dLdP = 0.f;
  if (kind <= 2) {
    const float e = P - Q;
    val = k < bins ? nrm_val(e, kind == 1) : 0.f;
    dLdP = k < bins ? gs * nrm_der(e, kind == 1) : 0.f;                    // gs = lam / (B * bins)
  } else if (kind == 3) {
    val = k < bins ? Q * (logf(Q) - logf(P)) : 0.f;
    dLdP = k < bins ? -gs * Q / P : 0.f;                                   // gs = lam / B
  } else {
    const float sq = k < bins ? sqrtf(P * Q) : 0.f;
    const float ws = wave_sum(sq);
    if ((k & 63) == 0) red[0][k >> 6] = ws;
    __syncthreads();
    const float z = -(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    __syncthreads();
    const float ct = -(1.f / (t * t));
    float psi, dpsi;
    if (z <= ct) { psi = -(1.f / t) * logf(-z); dpsi = -1.f / (t * z); }
    else { psi = t * z - (1.f / t) * logf(1.f / (t * t)) + (1.f / t); dpsi = t; }
    val = k == 0 ? psi : 0.f;                                              // one value per image
    dLdP = k < bins ? gs * dpsi * (-0.5f * sqrtf(Q / P)) : 0.f;            // gs = lam / B
  }
  float lv = wave_sum(val), dp = wave_sum(dLdP * P);
  if ((k & 63) == 0) { red[0][k >> 6] = lv; red[1][k >> 6] = dp; }
  __syncthreads();
  if (k == 0) {
    bc[0] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    lpart[b] = (double)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
  }
  __syncthreads();
  if (k < bins) dldh[(long)b * bins + k] = kde > 0.f ? dLdP / kde            // d/dh of h / n + eps
                                                      : (dLdP - bc[0]) / Sp;  // d/dh of (h + 1) / sum(h + 1)
}

template <int MODE>
__global__ void __launch_bounds__(256) k_soft_hist_grad(const float* __restrict__ pred, const float* __restrict__ dldh,
                                                        float* __restrict__ grad, long n, int bins, float delta,
                                                        float half, float sigma, int rb, int grad_accum) {
  extern __shared__ float dl[];                   // [bins]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < bins; i += 256) dl[i] = dldh[(long)b * bins + i];
  __syncthreads();
  const long per = (n + HB - 1) / HB, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float x = pred[(long)b * n + i];
    const int k0 = own_bin<MODE>(x, delta, bins);
    float g = 0.f;
    for (int k = max(k0 - rb, 0); k <= min(k0 + rb, bins - 1); ++k) {
      const float d = x - bin_center<MODE>(k, delta, bins);
      if (MODE == 0) {
        const float sa = sigmoid_f(sigma * (d + half)), sb = sigmoid_f(sigma * (d - half));
        g += dl[k] * sigma * (sa * (1.f - sa) - sb * (1.f - sb));
      } else {
        g += dl[k] * sigma * expf(-(d * d) / half) * (-2.f * d / half);
      }
    }
    const long o = (long)b * n + i;
    grad[o] = grad_accum ? grad[o] + g : g;
  }
}

int ew_blocks(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 2048 ? g : 2048);
}

}  // namespace

extern "C" {

int srhip_loss_pointwise(const float* pred, const float* target, const float* weight, float* grad, float* loss_out,
                         double* workspace, long n, int mode, float lam, float eps, int grad_accum, int loss_accum,
                         void* stream) {
  SR_REQUIRE(n > 0, "loss_pointwise: empty input");
  SR_REQUIRE(mode >= 0 && mode <= 3, "loss_pointwise: mode %d", mode);
  SR_REQUIRE(mode != 2 || eps > 0.f, "loss_pointwise: Charbonnier needs eps > 0");
  SR_REQUIRE(weight == nullptr || mode == 0, "loss_pointwise: per-pixel weights apply to L1 only (loss/main.py:62-72)");
  hipStream_t st = (hipStream_t)stream;
  const int g = ew_blocks(n);
  const double denom = mode == 3 ? 1.0 : (double)n;          // L2Sum: MSELoss(reduction='sum')
  // Charbonnier: d/dpred sqrt((t-p)^2 + eps) = (p - t) / sqrt(...)
  hipLaunchKernelGGL(k_loss_pointwise, dim3(g), dim3(256), 0, st, pred, target, weight, grad, workspace, n, mode,
                     (float)((double)lam / denom), eps, grad_accum);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, workspace, g, (double)lam / denom, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_pointwise");
  return 0;
}

int srhip_loss_bounded(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                       long n, float lam, float eps, float t, float scale, int grad_accum, int loss_accum,
                       void* stream) {
  SR_REQUIRE(n > 0, "loss_bounded: empty input");
  SR_REQUIRE(eps >= 0.f && t > 0.f && scale > 0.f, "loss_bounded: eps >= 0, t > 0, scale > 0 (eps=%g t=%g scale=%g)",
             (double)eps, (double)t, (double)scale);
  hipStream_t st = (hipStream_t)stream;
  const int g = ew_blocks(n);
  const double k = (double)lam / (2.0 * (double)n);
  hipLaunchKernelGGL(k_loss_bounded, dim3(g), dim3(256), 0, st, pred, target, grad, workspace, n, eps, t, scale,
                     (float)(k * (double)scale), grad_accum);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, workspace, g, k, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_bounded");
  return 0;
}

int srhip_l1_sparsity(const float* w, float* grad, float* loss_out, double* workspace, long n, float lam,
                      int loss_accum, void* stream) {
  SR_REQUIRE(n > 0, "l1_sparsity: empty input");
  hipStream_t st = (hipStream_t)stream;
  const int g = ew_blocks(n);
  hipLaunchKernelGGL(k_l1_sparsity, dim3(g), dim3(256), 0, st, w, grad, workspace, n, lam);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, workspace, g, (double)lam, loss_out, loss_accum);
  SR_LAUNCH_CHECK("l1_sparsity");
  return 0;
}

int srhip_loss_local_moments(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                             int B, int H, int W, float lam, int grad_accum, int loss_accum, void* stream) {
  SR_REQUIRE(B > 0 && H >= 2 && W >= 2, "loss_local_moments: H, W >= 2 (reflect padding), got %dx%d", H, W);
  SR_REQUIRE(B <= 65535, "loss_local_moments: batch %d", B);
  hipStream_t st = (hipStream_t)stream;
  const double count = (double)B * H * W;
  dim3 grid(sr_cdiv(W, TS), sr_cdiv(H, TS), B);
  hipLaunchKernelGGL(k_loss_local_moments, grid, dim3(256), 0, st, pred, target, grad, workspace, H, W,
                     (float)((double)lam / count), grad_accum);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, workspace, (int)srhip_loss_stencil_ws(B, H, W),
                     (double)lam / count, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_local_moments");
  return 0;
}

long srhip_loss_hist_ws(int B, int bins) { return (long)B * HB * 2 * bins + (long)B * bins + 2L * B; }

int srhip_loss_hist(const float* pred, const float* target, float* grad, float* loss_out, float* workspace, int B,
                    long n, int bins, float sigma, int norm, float lam, int grad_accum, int loss_accum, float elb_t,
                    void* stream) {
  SR_REQUIRE(B > 0 && n > 0 && B <= 65535, "loss_hist: empty input");
  SR_REQUIRE(bins > 0 && bins <= 256, "loss_hist: 1..256 bins (got %d)", bins);
  SR_REQUIRE(sigma > 0.f && norm >= 1 && norm <= 4, "loss_hist: sigma > 0, norm 1 | 2 | 3 (KL) | 4 (Bhattacharyya)");
  SR_REQUIRE(norm != 4 || elb_t > 0.f, "loss_hist: the barrier parameter must be > 0");
  hipStream_t st = (hipStream_t)stream;
  // SoftHistogram(bins, min=0, max=1, sigma): delta = 1/bins as the reference's float32 tensor arithmetic sees it
  const float delta = (float)(1.0 / (double)bins), half = (float)((1.0 / (double)bins) / 2.0);
  const double reach = 104.0 / ((double)sigma * (double)delta);
  const int rb = reach >= (double)bins ? bins : (int)reach + 2;
  float* part = workspace;
  float* dldh = part + (long)B * HB * 2 * bins;
  double* lpart = (double*)(dldh + (long)B * bins + (((long)B * bins) & 1));       // 8-byte aligned
  hipLaunchKernelGGL(k_soft_hist<0>, dim3(HB, B), dim3(256), 2 * bins * sizeof(float), st, pred, target, part, n, bins,
                     delta, half, sigma, rb);
  const double cnt = norm <= 2 ? (double)B * bins : (double)B;      // KL (batchmean) and the barrier: mean over images
  hipLaunchKernelGGL(k_hist_finalize, dim3(B), dim3(256), 0, st, part, dldh, lpart, bins, norm,
                     (float)((double)lam / cnt), 0.f, elb_t);
  if (grad)
    hipLaunchKernelGGL(k_soft_hist_grad<0>, dim3(HB, B), dim3(256), bins * sizeof(float), st, pred, dldh, grad, n, bins,
                       delta, half, sigma, rb, grad_accum);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, lpart, B, (double)lam / cnt, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_hist");
  return 0;
}

int srhip_loss_kde(const float* pred, const float* target, float* grad, float* loss_out, float* workspace, int B,
                   long n, int bins, float kde_bw, int norm, float lam, int grad_accum, int loss_accum, float elb_t,
                   void* stream) {
  SR_REQUIRE(B > 0 && n > 0 && B <= 65535, "loss_kde: empty input");
  SR_REQUIRE(bins > 1 && bins <= 256, "loss_kde: 2..256 bins (got %d)", bins);
  SR_REQUIRE(kde_bw > 0.f && (norm == 1 || norm == 2 || norm == 4), "loss_kde: bandwidth > 0, norm 1, 2 or 4 (Bhattacharyya)");
  SR_REQUIRE(norm != 4 || elb_t > 0.f, "loss_kde: the barrier parameter must be > 0");
  hipStream_t st = (hipStream_t)stream;
  // GaussianKDE(kde_bw, nbin, max_color = 1, ndim = 1): float32 constants as the reference builds them
  const float c1 = (float)pow(2.0 * 3.14159265358979323846 * (double)kde_bw, -0.5), c2 = (float)(2.0 * (double)kde_bw);
  const float delta = (float)(1.0 / (double)(bins - 1));
  const double reach = sqrt(104.0 * (double)c2) / (double)delta;
  const int rb = reach >= (double)bins ? bins : (int)reach + 2;
  float* part = workspace;
  float* dldh = part + (long)B * HB * 2 * bins;
  double* lpart = (double*)(dldh + (long)B * bins + (((long)B * bins) & 1));
  hipLaunchKernelGGL(k_soft_hist<1>, dim3(HB, B), dim3(256), 2 * bins * sizeof(float), st, pred, target, part, n, bins,
                     delta, c2, c1, rb);
  // loss = norm(pred, trg).mean() / bins  (dlib/loss/main.py:893-895)
  const double cnt = norm <= 2 ? (double)B * bins * (double)bins : (double)B;
  hipLaunchKernelGGL(k_hist_finalize, dim3(B), dim3(256), 0, st, part, dldh, lpart, bins, norm,
                     (float)((double)lam / cnt), (float)n, elb_t);
  if (grad)
    hipLaunchKernelGGL(k_soft_hist_grad<1>, dim3(HB, B), dim3(256), bins * sizeof(float), st, pred, dldh, grad, n, bins,
                       delta, c2, c1, rb, grad_accum);
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, lpart, B, (double)lam / cnt, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_kde");
  return 0;
}

long srhip_loss_stencil_ws(int B, int H, int W) { return (long)B * sr_cdiv(H, TS) * sr_cdiv(W, TS); }

int srhip_loss_stencil(const float* pred, const float* target, float* grad, float* loss_out, double* workspace,
                       int B, int H, int W, int op, int ksz, int norm, int channel_norm, float lam, int grad_accum,
                       int loss_accum, void* stream) {
  SR_REQUIRE(B > 0 && H > 0 && W > 0, "loss_stencil: empty input");
  SR_REQUIRE(op >= 0 && op <= 2, "loss_stencil: op %d (0 image gradient, 1 Laplacian, 2 local variation)", op);
  SR_REQUIRE(op != 2 || ksz == 3 || ksz == 5 || ksz == 7, "loss_stencil: local variation window %d (3, 5 or 7)", ksz);
  SR_REQUIRE(norm == 1 || norm == 2, "loss_stencil: norm %d (1 or 2)", norm);
  SR_REQUIRE(B <= 65535, "loss_stencil: batch %d", B);
  hipStream_t st = (hipStream_t)stream;
  const int K = op == 0 ? 2 : (op == 1 ? 1 : ksz * ksz - 1);
  const double count = (double)B * H * W * (channel_norm ? 1 : K);
  const float gs = (float)((double)lam / count);
  dim3 grid(sr_cdiv(W, TS), sr_cdiv(H, TS), B);
  const int l1 = norm == 1;
#define SR_STENCIL(OP_, R_) \
  hipLaunchKernelGGL((k_loss_stencil<OP_, R_>), grid, dim3(256), 0, st, pred, target, grad, workspace, H, W, l1, \
                     channel_norm, gs, grad_accum)
  if (op == 0) SR_STENCIL(0, 1);
  else if (op == 1) SR_STENCIL(1, 1);
  else if (ksz == 3) SR_STENCIL(2, 1);
  else if (ksz == 5) SR_STENCIL(2, 2);
  else SR_STENCIL(2, 3);
#undef SR_STENCIL
  hipLaunchKernelGGL(k_sum_partials_d, dim3(1), dim3(1024), 0, st, workspace, (int)srhip_loss_stencil_ws(B, H, W),
                     (double)lam / count, loss_out, loss_accum);
  SR_LAUNCH_CHECK("loss_stencil");
  return 0;
}

}  // extern "C"
