// Pieces of GRL's mixed-attention blocks, evaluation forward (reference dlib/models/network_grl.py): the cosine attention
// of Attention.attn (:338-355) under AffineTransform (:296-319) -- clamped per-head logit scale, 16 sigmoid(CPB MLP) bias
// gathered by a relative-position index, the shifted-window mask -- between the windows of two channels-last images that
// share a window grid (8x8 windows against themselves for WindowAttention :381-412; 4x4 anchor windows against 8x8 stripes
// and back for AnchorStripeAttention :463-514), and AnchorLinear's average pooling (:611-620).  roll / window_partition /
// window_reverse are address arithmetic.  The Linear / conv layers around run on the GEMM and conv kernels.
#include "common.h"

namespace {

inline int ew_blocks(long n) { const long g = (n + 255) / 256; return (int)(g < 16384 ? g : 16384); }

// nn.AvgPool2d(k, k) on channels-last data: sums in (ky, kx) order, divided by k*k
__global__ void __launch_bounds__(256) k_avgpool(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                                 int k, int Ho, int Wo) {
  const long n = (long)B * Ho * Wo * C;
  const float div = (float)(k * k);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    const long b = p / ((long)Wo * Ho);
    float a = 0.f;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) a += x[((b * H + yo * k + ky) * W + xo * k + kx) * C + c];
    out[i] = a / div;
  }
}

// biasT[h][j][i] = 16 sigmoid(table[index[i][j]][h]): key-major, so that the queries of a wave read consecutive words
__global__ void __launch_bounds__(256) k_cpb_bias(const float* __restrict__ table, const long long* __restrict__ index,
                                                  float* __restrict__ biasT, int heads, int N1, int N2, int entries) {
  const int n = heads * N1 * N2;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
    const int i = t % N1, j = (t / N1) % N2, h = t / (N1 * N2);
    long long e = index[(long)i * N2 + j];
    e = e < 0 ? 0 : (e >= entries ? entries - 1 : e);
    biasT[t] = 16.f / (1.f + expf(-table[e * heads + h]));
  }
}

struct Side {            // the tokens of one side: a channels-last image cut into wh x ww windows
  const float* p;        // first channel of head 0
  long ld;               // floats between pixels
  int H, W, wh, ww;
};

struct CosAttnArgs {
  Side q, k;
  const float* v;        // values: the key side's geometry
  long ldv;
  const float* logit_scale;
  const float* biasT;    // [heads][N2][N1]
  float* out;            // at the query token's pixel
  long ldo;
  int heads, d, shift, nwy, nwx;
};

__device__ __forceinline__ int region(int p, int n, int w, int s) { return p < n - w ? 0 : (p < n - s ? 1 : 2); }

// token t of window (wy, wx) of sample b: its pixel in the unrolled image and (shift > 0) its region id in the rolled one
__device__ __forceinline__ long token_pixel(const Side& s, long b, int wy, int wx, int t, int shift, int& rid) {
  const int yr = wy * s.wh + t / s.ww, xr = wx * s.ww + t % s.ww;
  rid = shift > 0 ? region(yr, s.H, s.wh, shift) * 3 + region(xr, s.W, s.ww, shift) : 0;
  int y = yr + shift, x = xr + shift;
  y = y >= s.H ? y - s.H : y;
  x = x >= s.W ? x - s.W : x;
  return (b * s.H + y) * s.W + x;
}

// One 256-thread block per (window, head).  Keys (unit length), values and queries (unit length times the logit scale) are
// staged in LDS with coalesced loads; G = 256 / N1 (a power of two) neighbouring lanes share a query and walk interleaved
// keys with a running-maximum softmax each, their partial (max, sum, accumulator) triples are merged by lane shuffles, and
// the rows leave through LDS so that the stores are contiguous per token.  N1, N2 <= 64, head width <= DM.
template <int DM>
__global__ void __launch_bounds__(256, DM <= 32 ? 4 : 2) k_cos_attn(const CosAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[64][DM + 4], Vs[64][DM + 4], Qs[64][DM + 4];   // rows 16 B aligned, 4 banks apart
  __shared__ int Rk[64], Rq[64];
  __shared__ long Pq[64];
  const int tid = threadIdx.x, d = a.d;
  const int h = blockIdx.x % a.heads;
  long w = blockIdx.x / a.heads;
  const int wx = (int)(w % a.nwx), wy = (int)((w / a.nwx) % a.nwy);
  const long b = w / ((long)a.nwx * a.nwy);
  const int N1 = a.q.wh * a.q.ww, N2 = a.k.wh * a.k.ww;
  {                                                  // all global loads in flight before the first LDS store
    constexpr int IT = 64 * DM / 256;
    float kr[IT], vr[IT], qr[IT];
    const int c = tid % DM;
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int r = tid / DM + u * (256 / DM);
      kr[u] = vr[u] = qr[u] = 0.f;
      int rid;
      if (r < N2) {
        const long px = token_pixel(a.k, b, wy, wx, r, a.shift, rid);
        if (c < d) {
          kr[u] = a.k.p[px * a.k.ld + h * d + c];
          vr[u] = a.v[px * a.ldv + h * d + c];
        }
        if (c == 0) Rk[r] = rid;
      }
      if (r < N1) {
        const long px = token_pixel(a.q, b, wy, wx, r, a.shift, rid);
        if (c < d) qr[u] = a.q.p[px * a.q.ld + h * d + c];
        if (c == 0) { Rq[r] = rid; Pq[r] = px; }
      }
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {                   // columns d .. DM-1 are zero: the dot products run over DM
      const int r = tid / DM + u * (256 / DM);
      Ks[r][c] = kr[u];
      Vs[r][c] = vr[u];
      Qs[r][c] = qr[u];
    }
  }
  __syncthreads();
  if (tid < N2) {                                    // F.normalize(k, dim=-1): k / max(|k|, 1e-12)
    float s = 0.f;
    for (int c = 0; c < d; ++c) s += Ks[tid][c] * Ks[tid][c];
    const float r = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = 0; c < d; ++c) Ks[tid][c] *= r;
  } else if (tid >= 128 && tid - 128 < N1) {         // F.normalize(q) * clamp(logit_scale, max = log 100).exp()
    const int i = tid - 128;
    float s = 0.f;
    for (int c = 0; c < d; ++c) s += Qs[i][c] * Qs[i][c];
    const float r = expf(fminf(a.logit_scale[h], 4.605170185988092f)) / fmaxf(sqrtf(s), 1e-12f);
    for (int c = 0; c < d; ++c) Qs[i][c] *= r;
  }
  __syncthreads();
  int G = 64;
  while (G * N1 > 256) G >>= 1;
  const int i = tid / G, g = tid % G;
  const bool live = i < N1;
  float q[DM], acc[DM];
#pragma unroll
  for (int c = 0; c < DM; ++c) {
    q[c] = live ? Qs[i][c] : 0.f;
    acc[c] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  if (live) {
    const float* bias = a.biasT + (long)h * N2 * N1 + i;
    const int rq = Rq[i];
    float bz[16];                                    // G >= 4, N2 <= 64: at most 16 keys per lane
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = g + u * G;
      bz[u] = j < N2 ? bias[(long)j * N1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = g + u * G;
      if (j < N2) {
        float e = bz[u];
#pragma unroll
        for (int c = 0; c < DM; c += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(&Ks[j][c]);
          e += q[c] * kk.x + q[c + 1] * kk.y + q[c + 2] * kk.z + q[c + 3] * kk.w;
        }
        if (a.shift > 0 && Rk[j] != rq) e += -100.f;
        const float mn = fmaxf(m, e), cf = expf(m - mn), p = expf(e - mn);
        l = l * cf + p;
#pragma unroll
        for (int c = 0; c < DM; c += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(&Vs[j][c]);
          acc[c] = acc[c] * cf + p * vv.x;
          acc[c + 1] = acc[c + 1] * cf + p * vv.y;
          acc[c + 2] = acc[c + 2] * cf + p * vv.z;
          acc[c + 3] = acc[c + 3] * cf + p * vv.w;
        }
        m = mn;
      }
    }
  }
  for (int o = 1; o < G; o <<= 1) {                  // the G lanes of a query are neighbours in one wave
    const float m2 = __shfl_xor(m, o), l2 = __shfl_xor(l, o);
    const float mn = fmaxf(m, m2);
    const float c1 = m == -INFINITY ? 0.f : expf(m - mn), c2 = m2 == -INFINITY ? 0.f : expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int c = 0; c < DM; ++c) acc[c] = acc[c] * c1 + __shfl_xor(acc[c], o) * c2;
    m = mn;
  }
  if (live && g == 0) {                              // row i of Qs was read by this wave only
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < DM; ++c) Qs[i][c] = acc[c] * inv;
  }
  __syncthreads();
  for (int t = tid; t < N1 * d; t += 256) {
    const int r = t / d, c = t - r * d;
    a.out[Pq[r] * a.ldo + h * d + c] = Qs[r][c];
  }
}

}  // namespace

extern "C" {

/* nn.AvgPool2d(k, k) (AnchorLinear, network_grl.py:603-620) on channels-last data: [B,H,W,C] -> [B,H/k,W/k,C] */
int srhip_avgpool2d(const float* x, float* out, int B, int H, int W, int C, int k, void* stream) {
  SR_REQUIRE(x && out && B > 0 && C > 0 && k > 0 && H >= k && W >= k, "avgpool2d: bad arguments (H=%d W=%d k=%d)", H, W, k);
  const int Ho = H / k, Wo = W / k;
  hipLaunchKernelGGL(k_avgpool, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, k,
                     Ho, Wo);
  SR_LAUNCH_CHECK("avgpool2d");
  return 0;
}

/* AffineTransform's bias (network_grl.py:305-311): biasT[h][j][i] = 16 sigmoid(table[index[i][j]][h]); table [entries][heads]
 * is the CPB MLP's output over the relative-coordinates table, index [N1][N2] int64 */
int srhip_cpb_bias(const float* table, const long long* index, float* biasT, int heads, int N1, int N2, int entries, void* stream) {
  SR_REQUIRE(table && index && biasT && heads > 0 && N1 > 0 && N2 > 0 && entries > 0, "cpb_bias: bad arguments");
  hipLaunchKernelGGL(k_cpb_bias, dim3(ew_blocks((long)heads * N1 * N2)), dim3(256), 0, (hipStream_t)stream, table, index, biasT,
                     heads, N1, N2, entries);
  SR_LAUNCH_CHECK("cpb_bias");
  return 0;
}

/* softmax(exp(min(logit_scale, log 100)) cos(q, k) + bias + mask) v per (window, head) (Attention.attn, network_grl.py:338-355)
 * between the qwh x qww windows of a [B,qH,qW,.] image and the kwh x kww windows of a [B,kH,kW,.] image on the same window
 * grid.  q / k / v / out point at the first channel of head 0; ld* = floats between pixels; head h owns channels [h d, (h+1) d).
 * shift > 0 (both sides the same geometry): windows of the image rolled by -shift with the 3 x 3 region mask of
 * calculate_mask (:1607-1622); the result lands at the query token's own pixel, i.e. window_reverse + roll back are done. */
int srhip_cosine_window_attention(const float* q, long ldq, int qH, int qW, int qwh, int qww, const float* k, long ldk,
                                  const float* v, long ldv, int kH, int kW, int kwh, int kww, const float* logit_scale,
                                  const float* biasT, float* out, long ldo, int B, int heads, int d, int shift, void* stream) {
  SR_REQUIRE(q && k && v && logit_scale && biasT && out && B > 0 && heads > 0 && d > 0, "cosine_window_attention: bad arguments");
  SR_REQUIRE(qwh > 0 && qww > 0 && kwh > 0 && kww > 0 && qH % qwh == 0 && qW % qww == 0 && kH % kwh == 0 && kW % kww == 0 &&
                 qH / qwh == kH / kwh && qW / qww == kW / kww,
             "cosine_window_attention: %dx%d in %dx%d windows against %dx%d in %dx%d windows is not one window grid", qH, qW, qwh,
             qww, kH, kW, kwh, kww);
  SR_REQUIRE(qwh * qww <= 64 && kwh * kww <= 64 && d <= 64, "cosine_window_attention: windows of at most 64 tokens, heads of at most 64 channels");
  SR_REQUIRE(shift >= 0 && (shift == 0 || (qH == kH && qW == kW && qwh == kwh && qww == kww && shift < qwh && shift < qww)),
             "cosine_window_attention: a shift needs the same geometry on both sides");
  CosAttnArgs a;
  a.q = Side{q, ldq, qH, qW, qwh, qww};
  a.k = Side{k, ldk, kH, kW, kwh, kww};
  a.v = v;
  a.ldv = ldv;
  a.logit_scale = logit_scale;
  a.biasT = biasT;
  a.out = out;
  a.ldo = ldo;
  a.heads = heads;
  a.d = d;
  a.shift = shift;
  a.nwy = qH / qwh;
  a.nwx = qW / qww;
  const long blocks = (long)B * a.nwy * a.nwx * heads;
  SR_REQUIRE(blocks < (1L << 31), "cosine_window_attention: too many windows");
  if (d <= 32)
    hipLaunchKernelGGL(k_cos_attn<32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_cos_attn<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  SR_LAUNCH_CHECK("cosine_window_attention");
  return 0;
}

}  // extern "C"
