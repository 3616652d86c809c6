// Pieces of OmniSR's omni self-attention blocks, evaluation forward (reference dlib/models/network_omni_sr.py): depthwise
// 3x3 convs (MBConv :178, Gated_Conv_FeedForward :318, Channel_Attention :348), the 64-token window / grid attention with
// relative-position bias (Attention.forward :258-306), the channel attention of a window / of a grid position
// (Channel_Attention(.._grid).forward :353-428), the gated GELU (:326), and ESA's max pooling, bilinear resize and gate
// (:104-114).  Channels last throughout; simple one-pass kernels: the net's cost sits in its 1x1 convs (GEMMs).
#include "common.h"

namespace {

inline int ew_blocks(long n) { const long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }

// depthwise 3x3, padding 1: out[p][c] = bias[c] + sum_t w[c][t] x[p + t][c]
__global__ void __launch_bounds__(256) k_dwconv3x3(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, long ldo, int B,
                                                   int H, int W, int C) {
  const long n = (long)B * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xx = (int)(p % W), y = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    float a = bias ? bias[c] : 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xq = xx + dx;
        if (yy >= 0 && yy < H && xq >= 0 && xq < W) a += w[c * 9 + (dy + 1) * 3 + dx + 1] * x[((b * H + yy) * W + xq) * ldx + c];
      }
    out[p * ldo + c] = a;
  }
}

// the same with four channels per lane (C, ldx, ldo multiples of 4): 16-byte loads, a quarter of the index arithmetic
__global__ void __launch_bounds__(256) k_dwconv3x3_v4(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, long ldo, int B,
                                                      int H, int W, int C4) {
  const long n = (long)B * H * W * C4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long p = i / C4;
    const int xx = (int)(p % W), y = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    float4 a = bias ? *reinterpret_cast<const float4*>(bias + c) : float4{0.f, 0.f, 0.f, 0.f};
    float wr[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[k][t] = w[(c + k) * 9 + t];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xq = xx + dx;
        if (yy >= 0 && yy < H && xq >= 0 && xq < W) {
          const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + yy) * W + xq) * ldx + c);
          const int t = (dy + 1) * 3 + dx + 1;
          a.x += wr[0][t] * v.x;
          a.y += wr[1][t] * v.y;
          a.z += wr[2][t] * v.z;
          a.w += wr[3][t] * v.w;
        }
      }
    *reinterpret_cast<float4*>(out + p * ldo + c) = a;
  }
}

// attention of one (group of n <= 64 tokens, head): rows of qkv [.., 3C] = q | k | v, head-major channels
constexpr int WA_N = 64, WA_D = 32;
__global__ void __launch_bounds__(256) k_group_attention(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                         float* __restrict__ out, int n, int C, int heads, int dh, float scale) {
  __shared__ float sq[WA_N][WA_D + 1], sk[WA_N][WA_D + 1], sv[WA_N][WA_D + 1], ss[WA_N][WA_N + 1];
  const int tid = threadIdx.x;
  const long grp = blockIdx.x / heads;
  const int h = blockIdx.x % heads;
  const float* base = qkv + grp * n * 3L * C + h * dh;
  for (int i = tid; i < WA_N * dh; i += 256) {            // rows n .. 63 are zero (the tiles below walk all 64)
    const int r = i / dh, e = i - r * dh;
    const bool in = r < n;
    sq[r][e] = in ? base[r * 3L * C + e] * scale : 0.f;
    sk[r][e] = in ? base[r * 3L * C + C + e] : 0.f;
    sv[r][e] = in ? base[r * 3L * C + 2 * C + e] : 0.f;
  }
  __syncthreads();
  {                                                       // S: a 4 x 4 tile of (query, key) pairs per thread
    const int i0 = (tid >> 4) * 4, j0 = (tid & 15) * 4;
    float a[4][4] = {};
    for (int e = 0; e < dh; ++e) {
      float qv[4], kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { qv[u] = sq[i0 + u][e]; kv[u] = sk[j0 + u][e]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) a[u][v] += qv[u] * kv[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = i0 + u, j = j0 + v;
        if (i < n && j < n) ss[i][j] = a[u][v] + (bias ? bias[((long)h * n + i) * n + j] : 0.f);
      }
  }
  __syncthreads();
  for (int i = tid >> 2; i < n; i += 64) {               // four lanes per row
    const int q4 = tid & 3;
    float mx = -3.0e38f;
    for (int j = q4; j < n; j += 4) mx = fmaxf(mx, ss[i][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    for (int j = q4; j < n; j += 4) { const float e = expf(ss[i][j] - mx); ss[i][j] = e; sum += e; }
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.0f / sum;
    for (int j = q4; j < n; j += 4) ss[i][j] *= inv;
  }
  __syncthreads();
  for (int p = tid; p < (WA_N / 4) * dh; p += 256) {      // O: four rows of one channel per thread
    const int i0 = (p / dh) * 4, e = p % dh;
    float a[4] = {};
    for (int j = 0; j < n; ++j) {
      const float v = sv[j][e];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += ss[i0 + u][j] * v;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < n) out[(grp * n + i0 + u) * (long)C + h * dh + e] = a[u];
  }
}

// channel attention of one (sample, group, head): d x d attention between the channels' L-vectors.
// pixel of (group gi, element l): window form: group = window (gy, gx), l = (py, px); grid form: group = (py, px), l = window
constexpr int CA_D = 16;
__device__ __forceinline__ long ca_pixel(int b, int gi, int l, int H, int W, int ps, int grid) {
  const int nx = W / ps;
  int wy, wx, py, px;
  if (!grid) { wy = gi / nx; wx = gi % nx; py = l / ps; px = l % ps; }
  else { py = gi / ps; px = gi % ps; wy = l / nx; wx = l % nx; }
  return ((long)b * H + wy * ps + py) * W + wx * ps + px;
}
constexpr int CA_L = 64;
__global__ void __launch_bounds__(256) k_channel_attention(const float* __restrict__ qkv, const float* __restrict__ temp,
                                                           float* __restrict__ out, int H, int W, int C, int heads, int ps,
                                                           int grid, int ngroups, int L) {
  __shared__ float gram[CA_D][CA_D + 1], nq[CA_D], nk[CA_D];
  __shared__ float tq[CA_L][CA_D + 1], tk[CA_L][CA_D + 1], tv[CA_L][CA_D + 1];     // the group's q / k / v of this head
  __shared__ long pixs[CA_L];
  const int tid = threadIdx.x;
  const int h = blockIdx.x % heads;
  const int gi = (blockIdx.x / heads) % ngroups;
  const int b = blockIdx.x / (heads * ngroups);
  const int d = C / heads;
  const int i = tid & 15, j = (tid >> 4) & 15;           // (i, j) channel pair; 256 threads = 16 x 16
  float g = 0.f, qq = 0.f, kk = 0.f;
  // Gram matrix q_i . k_j and the squared norms over the L elements, CA_L of them staged at a time
  for (int l0 = 0; l0 < L; l0 += CA_L) {
    const int nl = min(CA_L, L - l0);
    __syncthreads();
    for (int t = tid; t < nl; t += 256) pixs[t] = ca_pixel(b, gi, l0 + t, H, W, ps, grid);
    __syncthreads();
    for (int t = tid; t < nl * d; t += 256) {
      const int l = t / d, c = t - l * d;
      const float* px = qkv + pixs[l] * 3L * C + h * d + c;
      tq[l][c] = px[0];
      tk[l][c] = px[C];
      tv[l][c] = px[2 * C];
    }
    __syncthreads();
    if (i < d && j < d)
      for (int l = 0; l < nl; ++l) {
        const float qv = tq[l][i], kv = tk[l][j];
        g += qv * kv;
        qq += qv * qv;
        kk += kv * kv;
      }
  }
  if (j == 0) nq[i] = qq;
  if (i == 0) nk[j] = kk;
  __syncthreads();
  if (i < d && j < d)
    gram[i][j] = g / (fmaxf(sqrtf(nq[i]), 1e-12f) * fmaxf(sqrtf(nk[j]), 1e-12f)) * temp[h];
  __syncthreads();
  if (tid < d) {                                           // softmax over j of row tid
    float mx = -3.0e38f;
    for (int jj = 0; jj < d; ++jj) mx = fmaxf(mx, gram[tid][jj]);
    float sum = 0.f;
    for (int jj = 0; jj < d; ++jj) { const float e = expf(gram[tid][jj] - mx); gram[tid][jj] = e; sum += e; }
    for (int jj = 0; jj < d; ++jj) gram[tid][jj] /= sum;
  }
  __syncthreads();
  if (L <= CA_L) {                                         // the values are still staged
    for (int p = tid; p < L * d; p += 256) {               // out_i[l] = sum_j attn[i][j] v_j[l]
      const int l = p / d, ii = p - l * d;
      float a = 0.f;
      for (int jj = 0; jj < d; ++jj) a += gram[ii][jj] * tv[l][jj];
      out[pixs[l] * C + h * d + ii] = a;
    }
  } else {
    for (int p = tid; p < L * d; p += 256) {
      const int l = p / d, ii = p - l * d;
      const long pix = ca_pixel(b, gi, l, H, W, ps, grid);
      const float* v = qkv + pix * 3L * C + 2 * C + h * d;
      float a = 0.f;
      for (int jj = 0; jj < d; ++jj) a += gram[ii][jj] * v[jj];
      out[pix * C + h * d + ii] = a;
    }
  }
}

// out[t][c] = gelu(x[t][c]) * x[t][C + c]
__global__ void __launch_bounds__(256) k_gelu_gate(const float* __restrict__ x, float* __restrict__ out, long T, int C) {
  const long n = T * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long t = i / C;
    const int c = (int)(i - t * C);
    out[i] = gelu_f(x[t * 2 * C + c]) * x[t * 2 * C + C + c];
  }
}
// F.max_pool2d(k, stride s), no padding
__global__ void __launch_bounds__(256) k_maxpool(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                                 int k, int s, int Ho, int Wo) {
  const long n = (long)B * Ho * Wo * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    const long b = p / ((long)Wo * Ho);
    float m = -3.0e38f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) m = fmaxf(m, x[((b * H + yo * s + dy) * W + xo * s + dx) * C + c]);
    out[i] = m;
  }
}
// F.interpolate(mode='bilinear', align_corners=False) to (Ho, Wo)
__global__ void __launch_bounds__(256) k_bilinear(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                                  int Ho, int Wo) {
  const long n = (long)B * Ho * Wo * C;
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    const long b = p / ((long)Wo * Ho);
    const float fy = fmaxf(((float)yo + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)xo + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* xb = x + b * H * W * C + c;
    const float v00 = xb[((long)y0 * W + x0) * C], v01 = xb[((long)y0 * W + x1) * C];
    const float v10 = xb[((long)y1 * W + x0) * C], v11 = xb[((long)y1 * W + x1) * C];
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}
// out = x * sigmoid(g)
__global__ void __launch_bounds__(256) k_mul_sigmoid(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out,
                                                     long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] / (1.0f + expf(-g[i]));
}

// ---- pieces of the training path (srhip/omnisr_engine.py::_forward_tape): element-wise product, a periodic addend (the
// relative-position bias of every window) and its adjoint (the sum over the periods), max pooling's gradient
__global__ void __launch_bounds__(256) k_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] * b[i];
}
__global__ void __launch_bounds__(256) k_add_periodic(float* __restrict__ x, const float* __restrict__ v, long n, long period) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] += v[i % period];
}
// out[j] = sum over k of x[k * period + j]: one thread per j, k ascending (a fixed order)
__global__ void __launch_bounds__(256) k_sum_periodic(const float* __restrict__ x, float* __restrict__ out, long count, long period) {
  const long j = blockIdx.x * 256L + threadIdx.x;
  if (j >= period) return;
  float a = 0.f;
  for (long k = 0; k < count; ++k) a += x[k * period + j];
  out[j] = a;
}
// gradient of k_maxpool as a gather: an input pixel takes the gradient of every window whose first maximum it is
__global__ void __launch_bounds__(256) k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx,
                                                     int B, int H, int W, int C, int k, int s, int Ho, int Wo) {
  const long n = (long)B * H * W * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xx = (int)(p % W), y = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    float a = 0.f;
    for (int yo = min(y / s, Ho - 1); yo >= 0 && yo * s + k > y; --yo)
      for (int xo = min(xx / s, Wo - 1); xo >= 0 && xo * s + k > xx; --xo) {
        float m = -3.0e38f;
        int am = -1;
        for (int dy = 0; dy < k; ++dy)
          for (int dxx = 0; dxx < k; ++dxx) {
            const float v = x[((b * H + yo * s + dy) * W + xo * s + dxx) * C + c];
            if (v > m) { m = v; am = dy * k + dxx; }
          }
        if (am == (y - yo * s) * k + (xx - xo * s)) a += g[((b * Ho + yo) * Wo + xo) * C + c];
      }
    dx[i] = a;
  }
}

}  // namespace

extern "C" {

/* nn.Conv2d(C, C, 3, padding=1, groups=C) on channels-last data: x / out pixels of ldx / ldo floats; w [C][9]; bias may be NULL */
int srhip_dwconv3x3(const float* x, long ldx, const float* w, const float* bias, float* out, long ldo, int B, int H, int W, int C,
                    void* stream) {
  SR_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldo >= C, "dwconv3x3: bad arguments");
  const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((size_t)x | (size_t)out | (size_t)bias) % 16 == 0;
  if (v4)
    hipLaunchKernelGGL(k_dwconv3x3_v4, dim3(ew_blocks((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias,
                       out, ldo, B, H, W, C / 4);
  else
    hipLaunchKernelGGL(k_dwconv3x3, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, out, ldo,
                       B, H, W, C);
  SR_LAUNCH_CHECK("dwconv3x3");
  return 0;
}

/* softmax(scale q k^T + bias) v per (group of n consecutive token rows, head): qkv [groups*n][3C] (q | k | v, head-major
 * channels of dh), bias [heads][n][n] or NULL, out [groups*n][C].  n <= 64, dh <= 32. */
int srhip_group_attention(const float* qkv, const float* bias, float* out, long groups, int n, int C, int heads, float scale,
                          void* stream) {
  SR_REQUIRE(qkv && out && groups > 0 && n > 0 && n <= WA_N && heads > 0 && C % heads == 0 && C / heads <= WA_D,
             "group_attention: n <= 64, head dim <= 32 (n=%d C=%d heads=%d)", n, C, heads);
  SR_REQUIRE(groups * heads < (1L << 31), "group_attention: too many groups");
  hipLaunchKernelGGL(k_group_attention, dim3((unsigned)(groups * heads)), dim3(256), 0, (hipStream_t)stream, qkv, bias, out, n, C,
                     heads, C / heads, scale);
  SR_LAUNCH_CHECK("group_attention");
  return 0;
}

/* OmniSR's channel attention on qkv [B][H][W][3C] -> out [B][H][W][C]: per (sample, group, head) the d x d attention between
 * L2-normalised channel vectors times temperature[head]; grid = 0: group = ps x ps window, vector = its pixels; grid = 1:
 * group = in-window position, vector = the windows.  d = C / heads <= 16; H, W multiples of ps. */
int srhip_channel_attention(const float* qkv, const float* temperature, float* out, int B, int H, int W, int C, int heads, int ps,
                            int grid, void* stream) {
  SR_REQUIRE(qkv && temperature && out && B > 0 && heads > 0 && C % heads == 0 && C / heads <= CA_D && ps > 0 && H % ps == 0 &&
             W % ps == 0, "channel_attention: head dim <= 16, H, W multiples of the window (C=%d heads=%d)", C, heads);
  const int nwin = (H / ps) * (W / ps), npos = ps * ps;
  const int ngroups = grid ? npos : nwin, L = grid ? nwin : npos;
  hipLaunchKernelGGL(k_channel_attention, dim3(B * ngroups * heads), dim3(256), 0, (hipStream_t)stream, qkv, temperature, out, H, W,
                     C, heads, ps, grid, ngroups, L);
  SR_LAUNCH_CHECK("channel_attention");
  return 0;
}

/* out[t][c] = gelu(x[t][c]) * x[t][C + c], x [T][2C] (Gated_Conv_FeedForward :325-326) */
int srhip_gelu_gate(const float* x, float* out, long T, int C, void* stream) {
  SR_REQUIRE(x && out && T > 0 && C > 0, "gelu_gate: bad arguments");
  hipLaunchKernelGGL(k_gelu_gate, dim3(ew_blocks(T * C)), dim3(256), 0, (hipStream_t)stream, x, out, T, C);
  SR_LAUNCH_CHECK("gelu_gate");
  return 0;
}

/* F.max_pool2d(x, k, stride = s) on [B][H][W][C] -> [B][(H-k)/s+1][(W-k)/s+1][C] */
int srhip_maxpool2d(const float* x, float* out, int B, int H, int W, int C, int k, int s, void* stream) {
  SR_REQUIRE(x && out && B > 0 && C > 0 && k > 0 && s > 0 && H >= k && W >= k, "maxpool2d: bad arguments (H=%d W=%d k=%d)", H, W, k);
  const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
  hipLaunchKernelGGL(k_maxpool, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, k, s,
                     Ho, Wo);
  SR_LAUNCH_CHECK("maxpool2d");
  return 0;
}

/* F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=False) on channels-last data */
int srhip_bilinear_resize(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, void* stream) {
  SR_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "bilinear_resize: bad arguments");
  hipLaunchKernelGGL(k_bilinear, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, Ho, Wo);
  SR_LAUNCH_CHECK("bilinear_resize");
  return 0;
}

/* out = x * sigmoid(g) (ESA :113-114); out may alias x */
int srhip_mul_sigmoid(const float* x, const float* g, float* out, long n, void* stream) {
  SR_REQUIRE(x && g && out && n > 0, "mul_sigmoid: bad arguments");
  hipLaunchKernelGGL(k_mul_sigmoid, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, g, out, n);
  SR_LAUNCH_CHECK("mul_sigmoid");
  return 0;
}

/* ---- training path of OmniSR (the tape graph): small pieces its backward is composed from ---- */
/* out = a * b element-wise; out may alias a or b */
int srhip_mul(const float* a, const float* b, float* out, long n, void* stream) {
  SR_REQUIRE(a && b && out && n > 0, "mul: bad arguments");
  hipLaunchKernelGGL(k_mul, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  SR_LAUNCH_CHECK("mul");
  return 0;
}
/* x[i] += v[i % period]: the relative-position bias [heads][n][n] added to the logits of every window (network_omni_sr.py:291-294) */
int srhip_add_periodic(float* x, const float* v, long n, long period, void* stream) {
  SR_REQUIRE(x && v && n > 0 && period > 0 && n % period == 0, "add_periodic: n must be a multiple of the period");
  hipLaunchKernelGGL(k_add_periodic, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, v, n, period);
  SR_LAUNCH_CHECK("add_periodic");
  return 0;
}
/* out[j] = sum_k x[k * period + j], k < n / period: the adjoint of srhip_add_periodic (the bias gradient); fixed order */
int srhip_sum_periodic(const float* x, float* out, long n, long period, void* stream) {
  SR_REQUIRE(x && out && n > 0 && period > 0 && n % period == 0, "sum_periodic: n must be a multiple of the period");
  hipLaunchKernelGGL(k_sum_periodic, dim3(sr_cdiv(period, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n / period, period);
  SR_LAUNCH_CHECK("sum_periodic");
  return 0;
}
/* gradient of srhip_maxpool2d: dx [B][H][W][C] from x and g [B][Ho][Wo][C] (a gather: deterministic) */
int srhip_maxpool2d_bwd(const float* x, const float* g, float* dx, int B, int H, int W, int C, int k, int s, void* stream) {
  SR_REQUIRE(x && g && dx && B > 0 && C > 0 && k > 0 && s > 0 && H >= k && W >= k, "maxpool2d_bwd: bad arguments");
  const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, g, dx, B, H, W, C, k,
                     s, Ho, Wo);
  SR_LAUNCH_CHECK("maxpool2d_bwd");
  return 0;
}

}  // extern "C"
