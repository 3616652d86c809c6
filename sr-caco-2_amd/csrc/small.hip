// HBM-bound edge kernels: 1-channel 3x3 convolutions (the network's first and
// last layers on 1-channel microscopy patches) and the fused PSNR / MSE / NRMSE
// metric sweep.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int CO_PER_LANE = 4;  // Cout <= 256

// One wave works on a SEGMENT of 64 consecutive pixels of an image row: the 3 x 66
// input window sits in a wave-private LDS strip and slides through 9 registers, so a
// pixel costs 3 broadcast LDS reads + the FMAs + one coalesced access of its channel
// vector -- no per-pixel index arithmetic or bounds tests (the wave-per-pixel version
// spent ~60 instructions per pixel on those: 4-10x off the HBM time at 512 x 512).
constexpr int SEG = 64;

__device__ __forceinline__ void seg_decode(long seg, int H, int nsx, int& b, int& y, int& x0) {
  const int sx = (int)(seg % nsx);
  const long t = seg / nsx;
  y = (int)(t % H);
  b = (int)(t / H);
  x0 = sx * SEG;
}
// rows y-1..y+1, columns x0-1..x0+64 of the 1-channel image -> strip[3][66] (zero outside)
__device__ __forceinline__ void seg_stage(float* strip, const float* __restrict__ img, int b, int y, int x0,
                                          int H, int W, int lane) {
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    const int sy = y + rr - 1;
    const bool rok = sy >= 0 && sy < H;
    const float* row = img + ((long)b * H + (rok ? sy : 0)) * W;
    const int c0 = x0 - 1 + lane, c1 = x0 + 63 + lane;           // lanes 0..63, then lanes 0..1
    strip[rr * 66 + lane] = (rok && c0 >= 0 && c0 < W) ? row[c0] : 0.f;
    if (lane < 2) strip[rr * 66 + 64 + lane] = (rok && c1 < W) ? row[c1] : 0.f;
  }
}

// y[p][co] = b[co] + sum_t x[p+t] * w[co][t]     x: [B][H][W] (1 channel), y NHWC
// flip=1 uses w[co][8-t] (this is then the data-gradient of a Cout=1 conv).
// conv_first: network_swinir.py:786,945; head conv: network_nlsn.py:325.
__global__ void __launch_bounds__(256) k_conv_cin1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int B, int H, int W, int Co, int flip, long ldy) {
  __shared__ float strips[4][3 * 66];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = blockIdx.x * 4L + wv, nwave = gridDim.x * 4L;
  float* strip = strips[wv];
  float wr[CO_PER_LANE][9], br[CO_PER_LANE];
#pragma unroll
  for (int i = 0; i < CO_PER_LANE; ++i) {
    const int co = lane + 64 * i;
    br[i] = (co < Co && bias) ? bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[i][t] = co < Co ? w[co * 9 + ((flip & 1) ? 8 - t : t)] : 0.f;
  }
  const bool relu = (flip & 2) != 0;          // flags: bit 0 flipped taps, bit 1 ReLU on the output
  const int nsx = (W + SEG - 1) / SEG;
  const long nseg = (long)B * H * nsx;
  for (long seg = wave; seg < nseg; seg += nwave) {
    int b, yy, x0;
    seg_decode(seg, H, nsx, b, yy, x0);
    __builtin_amdgcn_wave_barrier();
    seg_stage(strip, x, b, yy, x0, H, W, lane);
    __builtin_amdgcn_wave_barrier();
    float xv[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { xv[rr][1] = strip[rr * 66]; xv[rr][2] = strip[rr * 66 + 1]; }
    const int npx = min(SEG, W - x0);
    float* yp = y + (((long)b * H + yy) * W + x0) * ldy;
    for (int px = 0; px < npx; ++px) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        xv[rr][0] = xv[rr][1]; xv[rr][1] = xv[rr][2]; xv[rr][2] = strip[rr * 66 + px + 2];
      }
#pragma unroll
      for (int i = 0; i < CO_PER_LANE; ++i) {
        const int co = lane + 64 * i;
        if (co < Co) {
          float a = br[i];
#pragma unroll
          for (int t = 0; t < 9; ++t) a += xv[t / 3][t % 3] * wr[i][t];
          yp[co] = relu ? fmaxf(a, 0.f) : a;
        }
      }
      yp += ldy;
    }
  }
}
// part[block][co][10]: dw[co][t] = sum_p dy[p][co]*x[p+t] (t<9), db[co] (t=9)
__global__ void __launch_bounds__(256) k_conv_cin1_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ part, int B, int H, int W, int Co,
                                                         long lddy) {
  __shared__ float strips[4][3 * 66];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = blockIdx.x * 4L + wv, nwave = gridDim.x * 4L;
  float* strip = strips[wv];
  float acc[CO_PER_LANE][10];
#pragma unroll
  for (int i = 0; i < CO_PER_LANE; ++i)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[i][t] = 0.f;
  const int nsx = (W + SEG - 1) / SEG;
  const long nseg = (long)B * H * nsx;
  for (long seg = wave; seg < nseg; seg += nwave) {
    int b, yy, x0;
    seg_decode(seg, H, nsx, b, yy, x0);
    __builtin_amdgcn_wave_barrier();
    seg_stage(strip, x, b, yy, x0, H, W, lane);
    __builtin_amdgcn_wave_barrier();
    float xv[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) { xv[rr][1] = strip[rr * 66]; xv[rr][2] = strip[rr * 66 + 1]; }
    const int npx = min(SEG, W - x0);
    const float* gp = dy + (((long)b * H + yy) * W + x0) * lddy;
    // eight pixels' gradients in flight per lane (one 256-byte row of dy per load: the walk is a pure stream and
    // ran at 1.6 TB/s with one load at a time)
    constexpr int PF = 8;
    for (int px0 = 0; px0 < npx; px0 += PF) {
      float gq[PF][CO_PER_LANE];
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int i = 0; i < CO_PER_LANE; ++i) {
          const int co = lane + 64 * i;
          gq[u][i] = (px0 + u < npx && co < Co) ? gp[(long)u * lddy + co] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (px0 + u < npx) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) {
            xv[rr][0] = xv[rr][1]; xv[rr][1] = xv[rr][2]; xv[rr][2] = strip[rr * 66 + px0 + u + 2];
          }
#pragma unroll
          for (int i = 0; i < CO_PER_LANE; ++i) {
            if (64 * i < Co) {           // (wave-uniform: a 64-channel tensor skips three quarters of these FMAs)
              const float g = gq[u][i];
#pragma unroll
              for (int t = 0; t < 9; ++t) acc[i][t] += g * xv[t / 3][t % 3];
              acc[i][9] += g;
            }
          }
        }
      }
      gp += (long)PF * lddy;
    }
  }
  // the block's four waves join their sums in LDS (wave after wave: the same bits every run) and leave ONE partial row:
  // the finisher, whose reads are a gather (one line per value), walks a quarter of the rows
  __shared__ float red[64 * CO_PER_LANE * 10];
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int i = 0; i < CO_PER_LANE; ++i) {
        const int co = lane + 64 * i;
        if (co < Co) {
#pragma unroll
          for (int t = 0; t < 10; ++t) {
            const float v = (w ? red[co * 10 + t] : 0.f) + acc[i][t];
            if (w < 3) red[co * 10 + t] = v;
            else part[((long)blockIdx.x * Co + co) * 10 + t] = v;
          }
        }
      }
    }
    if (w < 3) __syncthreads();
  }
}
// dw[co][t] = sum_blocks part (flip: written to 8-t), db[co]; one wave per output
__global__ void k_conv_cin1_wgrad_fin(const float* __restrict__ part, float* __restrict__ dw,
                                      float* __restrict__ db, int Co, int nwave, int flip) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= Co * 10) return;
  const int co = i / 10, t = i % 10;
  float a = 0.f;
  for (int wv = lane; wv < nwave; wv += 64) a += part[((long)wv * Co + co) * 10 + t];
  a = wave_sum(a);
  if (lane == 0) {
    if (t == 9) { if (db) db[co] = a; }
    else dw[co * 9 + (flip ? 8 - t : t)] = a;
  }
}
// Cout = 1: y[p] = b + sum_t sum_ci x[p+t][ci] * w[ci][t]   (x NHWC, Ci <= 256, Ci % 4 == 0)
// tail conv of the EDSR wiring, network_nlsn.py:347-350.
// Block = 128 threads, 16 x 32 output pixels, FOUR per thread (a vertical strip of 4).  The 34 x 18 halo goes through LDS in
// chunks of 16 channels (pixel pitch 20 floats: conflict-free ds_read_b128; 8-channel chunks fetch half sectors: 327 us),
// the weights straight from the kernel argument with uniform addresses (scalar loads into SGPRs, no LDS, no VGPRs): every
// input value is fetched from HBM once (plus the halo).  A strip reads 6 rows x 3 columns of halo vectors and 9 weight
// vectors per 4 channels -- 27 LDS reads for 4 pixels; with one pixel per thread (18 per pixel) the LDS carried 9.7 GB
// for the 537 MB of an 8 x 512 x 512 x 64 input and bounded the kernel (182 us, 2.9 TB/s).
constexpr int C1_CH = 16, C1_PIT = 20, C1_Q = C1_CH / 4;
constexpr int C1_TW = 16, C1_TH = 32, C1_HC = C1_TW + 2, C1_HR = C1_TH + 2;
__global__ void __launch_bounds__(128, 2) k_conv_cout1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int B, int H, int W, int Ci, long ldx) {
  __shared__ __attribute__((aligned(16))) float xs[C1_HR * C1_HC * C1_PIT];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
  const int px = tid & 15, pr = tid >> 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // the chunk's halo values wait in registers while the previous chunk is consumed: the global loads of chunk k+1
  // overlap the FMAs of chunk k
  constexpr int NI = C1_HR * C1_HC * C1_Q;
  constexpr int NV = (NI + 127) / 128;
  unsigned src[NV];                  // offsets from x in 16-byte units (ldx % 4 == 0)
  unsigned inb = 0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = min(tid + k * 128, NI - 1);
    const int hp = i / C1_Q, c4 = i % C1_Q;
    const int hy = hp / C1_HC, hx = hp - hy * C1_HC;
    const int sy = ty * C1_TH + hy - 1, sx = tx * C1_TW + hx - 1;
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) inb |= 1u << k;
    src[k] = (unsigned)(((((long)b * H + min(max(sy, 0), H - 1)) * W + min(max(sx, 0), W - 1)) * ldx + c4 * 4) >> 2);
  }
  // (Round 6, measured and dropped: TWO chunks in flight, with __syncthreads() or with LDS-only barriers around the store --
  // 252 registers, 267 us against 163 on EDSR's 537-MB tail input.)
  f32x4 v[NV];
  auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c4 = min(tid + k * 128, NI - 1) % C1_Q;
      const bool ok = ((inb >> k) & 1u) && c0 + c4 * 4 < Ci;
      v[k] = *(const f32x4*)(ok ? x + (long)src[k] * 4 + c0 : x);
      if (!ok) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < Ci; c0 += C1_CH) {
    __syncthreads();                                   // previous chunk consumed
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + k * 128;
      if (i < NI) *(f32x4*)(xs + (i / C1_Q) * C1_PIT + (i % C1_Q) * 4) = v[k];
    }
    __syncthreads();
    if (c0 + C1_CH < Ci) fetch(c0 + C1_CH);
#pragma unroll
    for (int c4 = 0; c4 < C1_Q; ++c4) {
      if (c0 + c4 * 4 >= Ci) break;                   // (uniform) Ci % 4 == 0
      float wq[9][4];                                  // uniform: w[ci][tap] of the four channels
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wq[t][j] = w[(long)(c0 + c4 * 4 + j) * 9 + t];
#pragma unroll
      for (int r = 0; r < 6; ++r) {                    // halo row 4 pr + r feeds the strip's pixels p = r - 2 .. r
        f32x4 xv[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          xv[dx] = *(const f32x4*)(xs + ((4 * pr + r) * C1_HC + px + dx) * C1_PIT + c4 * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int tyy = r - p;
          if (tyy >= 0 && tyy < 3) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float* wt = wq[tyy * 3 + dx];
              acc[p] += xv[dx].x * wt[0] + xv[dx].y * wt[1] + xv[dx].z * wt[2] + xv[dx].w * wt[3];
            }
          }
        }
      }
    }
  }
  const int ox = tx * C1_TW + px;
  const float bb = bias ? bias[0] : 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int oy = ty * C1_TH + 4 * pr + p;
    if (oy < H && ox < W) y[((long)b * H + oy) * W + ox] = acc[p] + bb;
  }
}

// ----------------------------------------------------------------------------
// metrics: tensor2uint82float + PSNR / PSNR_Y / MSE / NRMSE in one pass, for
// "no ROI" and up to 8 ROI thresholds (utils_trainer.py:961-1032,
// utils_image.py:369-372,843-1007,618-653).  SSE is a sum of integers in
// double: exact and order independent.
// ----------------------------------------------------------------------------
constexpr int MAXTH = 9;  // slot 0 = no ROI
struct MetAcc {
  double sse, ssey;
  long cnt;
  float ymax, ymin;
};
__device__ __forceinline__ float u8f(float v) {  // (clamp(v,0,1)*255).round().clamp(0,255)
  v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
  return fminf(fmaxf(rintf(v), 0.f), 255.f);
}
__device__ __forceinline__ float ycb(float u8) {  // gray -> Y in [0,255] (float path)
  const float v255 = (u8 / 255.0f) * 255.0f;        // _rgb_tensor(E)/255 -> ycbcr multiplies back
  const float y = (65.481f * v255 + 128.553f * v255 + 24.966f * v255) / 255.0f + 16.0f;
  return fminf(fmaxf(y / 255.0f, 0.f), 1.f) * 255.0f;
}
__global__ void k_u8ify(const float* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = u8f(in[i]);
}
// raw[b][th][blk][8] doubles: sse, ssey, cnt, ymax, ymin_roi, ymin_all
__global__ void __launch_bounds__(256) k_metrics_pass(const float* __restrict__ E, const float* __restrict__ Hh,
                                                      double* __restrict__ raw, int H, int W, int border,
                                                      const int* __restrict__ ths, int nth, int in_is_u8) {
  const int b = blockIdx.y, nb = gridDim.x;
  const int h = H - 2 * border, w = W - 2 * border;
  const long n = (long)h * w;
  MetAcc acc[MAXTH];
  float ymin_all = 3.0e38f;
#pragma unroll
  for (int k = 0; k < MAXTH; ++k) { acc[k].sse = 0; acc[k].ssey = 0; acc[k].cnt = 0; acc[k].ymax = 0.f; acc[k].ymin = 3.0e38f; }
  int thv[MAXTH];
#pragma unroll
  for (int k = 0; k < MAXTH; ++k) thv[k] = (k >= 1 && k <= nth) ? ths[k - 1] : 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += nb * 256L) {
    const int yy = i / w + border, xx = i % w + border;
    const long o = ((long)b * H + yy) * W + xx;
    const float a = in_is_u8 ? E[o] : u8f(E[o]);
    const float t = in_is_u8 ? Hh[o] : u8f(Hh[o]);
    const double d = (double)a - (double)t;
    const double dyv = (double)ycb(a) - (double)ycb(t);
    ymin_all = fminf(ymin_all, t);
#pragma unroll
    for (int k = 0; k < MAXTH; ++k) {
      if (k <= nth && (k == 0 || t >= (float)thv[k])) {
        acc[k].sse += d * d; acc[k].ssey += dyv * dyv; acc[k].cnt += 1;
        acc[k].ymax = fmaxf(acc[k].ymax, t); acc[k].ymin = fminf(acc[k].ymin, t);
      }
    }
  }
  __shared__ double sh[4][6];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double ya = wave_min_d((double)ymin_all);
#pragma unroll
  for (int k = 0; k < MAXTH; ++k) {
    if (k > nth) break;                            // block-uniform
    const double s0 = wave_sum_d(acc[k].sse), s1 = wave_sum_d(acc[k].ssey);
    const double s2 = wave_sum_d((double)acc[k].cnt);
    const double s3 = wave_max_d((double)acc[k].ymax), s4 = wave_min_d((double)acc[k].ymin);
    __syncthreads();
    if (lane == 0) { sh[wv][0] = s0; sh[wv][1] = s1; sh[wv][2] = s2; sh[wv][3] = s3; sh[wv][4] = s4; sh[wv][5] = ya; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* o = raw + (((long)b * (nth + 1) + k) * nb + blockIdx.x) * 8;
      o[0] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
      o[1] = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
      o[2] = sh[0][2] + sh[1][2] + sh[2][2] + sh[3][2];
      o[3] = fmax(fmax(sh[0][3], sh[1][3]), fmax(sh[2][3], sh[3][3]));
      o[4] = fmin(fmin(sh[0][4], sh[1][4]), fmin(sh[2][4], sh[3][4]));
      o[5] = fmin(fmin(sh[0][5], sh[1][5]), fmin(sh[2][5], sh[3][5]));
    }
  }
}
// out[b][th][4] doubles: psnr, psnr_y, mse, nrmse
__global__ void k_metrics_fin(const double* __restrict__ raw, double* __restrict__ out, int nimg_th, int nb,
                              int nth1, long npix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg_th) return;
  const int k = i % nth1;
  double sse = 0, ssey = 0, cnt = 0, ymax = 0, ymin = 3.0e38, ymin_all = 3.0e38;
  for (int j = 0; j < nb; ++j) {
    const double* r = raw + ((long)i * nb + j) * 8;
    sse += r[0]; ssey += r[1]; cnt += r[2];
    ymax = fmax(ymax, r[3]); ymin = fmin(ymin, r[4]); ymin_all = fmin(ymin_all, r[5]);
  }
  double lo;
  if (k == 0) lo = ymin_all;
  else {
    const double min_masked = (cnt < (double)npix) ? 0.0 : ymin;   // min(y*roi)
    lo = fmax(ymin_all, min_masked);
  }
  const double denom_cnt = (k == 0) ? (double)npix : (cnt == 0 ? 1.0 : cnt);
  double mse = sse / denom_cnt, msey = ssey / denom_cnt;
  const double mse_p = mse < 1e-45 ? 1e-45 : mse, msey_p = msey < 1e-45 ? 1e-45 : msey;
  double den = ymax - lo;
  if (den == 0) den = 1.0;
  out[i * 4 + 0] = 20.0 * log10(255.0 / sqrt(mse_p));
  out[i * 4 + 1] = 20.0 * log10(255.0 / sqrt(msey_p));
  out[i * 4 + 2] = mse;
  out[i * 4 + 3] = sqrt(mse) / den;
}

}  // namespace

extern "C" {

int srhip_conv3x3_cin1_fwd(const float* x, const float* w, const float* bias, float* y, long ldy, int B,
                           int H, int W, int Co, int flip, void* stream) {
  SR_REQUIRE(Co <= 64 * CO_PER_LANE, "conv_cin1: Cout=%d > %d", Co, 64 * CO_PER_LANE);
  const long npix = (long)B * H * W;
  if (npix <= 0) return 0;
  long blocks = ((long)B * H * sr_cdiv(W, SEG) + 3) / 4;          // one segment per wave, at most 2048 blocks
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_conv_cin1_fwd, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     y, B, H, W, Co, flip, ldy);
  SR_LAUNCH_CHECK("conv_cin1_fwd");
  return 0;
}

long srhip_conv3x3_cin1_wgrad_ws(int Co) { return 8192L * Co * 10; }  // floats: [blocks][Co][10] (2048 blocks at most: a bound)

int srhip_conv3x3_cin1_wgrad(const float* x, const float* dy, long lddy, float* dw, float* db,
                             float* workspace, int B, int H, int W, int Co, int flip, void* stream) {
  SR_REQUIRE(Co <= 64 * CO_PER_LANE, "conv_cin1: Cout=%d > %d", Co, 64 * CO_PER_LANE);
  hipStream_t st = (hipStream_t)stream;
  // one 64-pixel row segment per wave and iteration; up to 8192 waves (the loop is a
  // dependent load -> FMA chain per pixel: parallelism across waves hides its latency)
  long segs = (long)B * H * sr_cdiv(W, SEG);
  const int blocks = (int)(segs >= 8192 ? 2048 : (segs + 3) / 4);
  hipLaunchKernelGGL(k_conv_cin1_wgrad, dim3(blocks), dim3(256), 0, st, x, dy, workspace, B, H, W, Co,
                     lddy);
  hipLaunchKernelGGL(k_conv_cin1_wgrad_fin, dim3(sr_cdiv(Co * 10, 4)), dim3(256), 0, st, workspace,
                     dw, db, Co, blocks, flip);
  SR_LAUNCH_CHECK("conv_cin1_wgrad");
  return 0;
}

int srhip_conv3x3_cout1_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, int B,
                            int H, int W, int Ci, void* stream) {
  SR_REQUIRE(Ci <= 64 * CO_PER_LANE, "conv_cout1: Cin=%d > %d", Ci, 64 * CO_PER_LANE);
  SR_REQUIRE(Ci % 4 == 0 && ldx % 4 == 0, "conv_cout1: Cin and ldx must be multiples of 4 (Cin=%d)", Ci);
  SR_REQUIRE((long)B * H * W * ldx < (1L << 34), "conv_cout1: input beyond 64 GB (32-bit offsets in 16-byte units)");
  if ((long)B * H * W <= 0) return 0;
  hipLaunchKernelGGL(k_conv_cout1_fwd, dim3(sr_cdiv(W, C1_TW), sr_cdiv(H, C1_TH), B), dim3(128), 0,
                     (hipStream_t)stream, x, w, bias, y, B, H, W, Ci, ldx);
  SR_LAUNCH_CHECK("conv_cout1_fwd");
  return 0;
}

int srhip_tensor2uint82float(const float* in, float* out, long n, void* stream) {
  if (n <= 0) return 0;
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_u8ify, dim3((int)g), dim3(256), 0, (hipStream_t)stream, in, out, n);
  SR_LAUNCH_CHECK("tensor2uint82float");
  return 0;
}

// workspace doubles: B*(nth+1)*64*8 ; out doubles: B*(nth+1)*4
long srhip_metrics_ws(int B, int nth) { return (long)B * (nth + 1) * 64 * 8; }

int srhip_metrics_psnr_family(const float* E, const float* Hh, int B, int H, int W, int border,
                              const int* thresholds_dev, int nth, int inputs_are_u8, double* workspace,
                              double* out, void* stream) {
  SR_REQUIRE(nth >= 0 && nth <= MAXTH - 1, "metrics: at most %d ROI thresholds", MAXTH - 1);
  SR_REQUIRE(B > 0 && H > 2 * border && W > 2 * border, "metrics: empty image after border crop");
  hipStream_t st = (hipStream_t)stream;
  const int nb = 64;
  hipLaunchKernelGGL(k_metrics_pass, dim3(nb, B), dim3(256), 0, st, E, Hh, workspace, H, W, border,
                     thresholds_dev, nth, inputs_are_u8);
  const int n = B * (nth + 1);
  hipLaunchKernelGGL(k_metrics_fin, dim3(sr_cdiv(n, 64)), dim3(64), 0, st, workspace, out, n, nb,
                     nth + 1, (long)(H - 2 * border) * (W - 2 * border));
  SR_LAUNCH_CHECK("metrics_psnr_family");
  return 0;
}

}  // extern "C"
