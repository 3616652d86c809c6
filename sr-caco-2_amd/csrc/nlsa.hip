// Non-Local Sparse Attention of NLSN, evaluation forward -- reference NonLocalSparseAttention.forward,
// dlib/models/network_nlsn.py:131-268: locality-sensitive hashing of the matching embedding (argmax over [r, -r] of its
// random rotations, :145-170), the tokens ordered by hash code (:199-207), attention of every 144-token chunk against
// itself and its two neighbouring chunks with the keys L2-normalised (:224-243), and the rounds combined by a softmax over
// their log-sum-exp scores (:258-262).
//
// Layout: everything stays token-major (channels last).  The order is ONE stable counting sort per (sample, round) of the
// 64-bit keys (sample, round, code | token) by their hash code (<= 128 buckets: k_bucket_order, in-tree -- no sort library),
// where the reference's torch.sort leaves the order of equal codes to the implementation.  The attention kernel gathers its rows through that order and writes its result and score
// at the TOKEN's own position of its round, so the un-sort (:251-253) is free and padded rows are simply not written.
#include "common.h"

namespace {

constexpr int TOK_BITS = 20;                 // tokens per sample < 2^20

// keys of one (token, round): code = argmax over cat([r, -r]) with torch.argmax's first-maximum rule
__global__ void __launch_bounds__(256) k_lsh_keys(const float* __restrict__ rot, long ldr, unsigned long long* __restrict__ keys,
                                                  int N, int L, int nh, int hbh) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)N * nh * L) return;
  const int l = (int)(idx % L);
  const int h = (int)((idx / L) % nh);
  const int n = (int)(idx / ((long)L * nh));
  const float* r = rot + ((long)n * L + l) * ldr + (long)h * hbh;
  float best = r[0];
  int code = 0;
  for (int i = 1; i < hbh; ++i)
    if (r[i] > best) { best = r[i]; code = i; }
  for (int i = 0; i < hbh; ++i)
    if (-r[i] > best) { best = -r[i]; code = hbh + i; }
  const unsigned long long grp = ((unsigned long long)(n * nh + h) * (2 * hbh) + code);
  keys[idx] = (grp << TOK_BITS) | (unsigned)l;
}

// Stable counting sort of the L keys of one (sample, round) by hash code: one block per group.  Histogram (integer LDS
// atomics), exclusive scan, then the tokens in tiles of 1024 in their own order: a token's slot = start of its code + tokens
// of that code in earlier tiles + in earlier waves of this tile + in lower lanes of its wave (ballot per distinct code).
constexpr int BO_T = 1024, BO_W = BO_T / 64, BO_HB = 128;
__global__ void __launch_bounds__(BO_T) k_bucket_order(const unsigned long long* __restrict__ keys,
                                                        unsigned long long* __restrict__ order, int L, int hb) {
  __shared__ int base[BO_HB];
  __shared__ int wcnt[BO_W][BO_HB];
  const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned long long* k = keys + (long)g * L;
  unsigned long long* o = order + (long)g * L;
  const unsigned long long grp0 = (unsigned long long)g * hb;
  if (tid < BO_HB) base[tid] = 0;
  for (int i = tid; i < BO_W * BO_HB; i += BO_T) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  for (int l = tid; l < L; l += BO_T) atomicAdd(&base[(int)((k[l] >> TOK_BITS) - grp0)], 1);
  __syncthreads();
  if (wave == 0) {                 // exclusive scan of <= 128 counts: two per lane
    const int a = base[2 * lane], b = base[2 * lane + 1];
    int s = a + b;
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(s, d);
      if (lane >= d) s += t;
    }
    base[2 * lane] = s - a - b;
    base[2 * lane + 1] = s - b;
  }
  __syncthreads();
  for (int l0 = 0; l0 < L; l0 += BO_T) {
    const int l = l0 + tid;
    const bool live = l < L;
    const unsigned long long key = live ? k[l] : 0ull;
    const int code = live ? (int)((key >> TOK_BITS) - grp0) : -1;
    int rank = 0;
    unsigned long long todo = __ballot(live);
    while (todo) {
      const int c0 = __shfl(code, __ffsll((long long)todo) - 1);
      const unsigned long long same = __ballot(code == c0);
      if (code == c0) rank = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][c0] = __popcll(same);
      todo &= ~same;
    }
    __syncthreads();
    if (live) {
      int before = base[code];
      for (int w = 0; w < wave; ++w) before += wcnt[w][code];
      o[before + rank] = key;
    }
    __syncthreads();
    if (tid < BO_HB) {
      int t = 0;
      for (int w = 0; w < BO_W; ++w) { t += wcnt[w][tid]; wcnt[w][tid] = 0; }
      base[tid] += t;
    }
    __syncthreads();
  }
}

// One block = one chunk of one (sample, round): queries in tiles of QT rows against 3 * cs keys.
constexpr int QT = 48;
constexpr int CE_MAX = 64;
constexpr int KP = CE_MAX + 4;               // key row pitch in LDS: 16-byte aligned rows, lanes walking the keys 4 banks apart

__global__ void __launch_bounds__(256) k_nlsa_attention(const float* __restrict__ xe, const float* __restrict__ ye,
                                                        const unsigned long long* __restrict__ order, float* __restrict__ ret,
                                                        float* __restrict__ score, int L, int Ce, int Cy, int nh, int cs,
                                                        int nchunks) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const sq = sm;                               // [QT][Ce]
  float* const sk = sq + QT * CE_MAX;                 // [cs][KP] one key chunk, normalised
  float* const ss = sk + (size_t)cs * KP;             // [QT][3 cs] scores, then probabilities
  int* const ktok = (int*)(ss + (size_t)QT * 3 * cs); // [3 cs] token of key j
  float* const rl = (float*)(ktok + 3 * cs);          // [QT] log-sum-exp of the row
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x % nchunks;
  const int nhh = blockIdx.x / nchunks;               // n * nh + h
  const int n = nhh / nh;
  const unsigned long long* ord = order + (long)nhh * L;
  const int padding = L % cs ? cs - L % cs : 0;
  // sorted position -> token (positions L .. L + padding - 1 repeat the last `padding` positions, :213-218)
  auto tok_of = [&](int pos) {
    if (pos >= L) pos -= padding;
    return (int)(ord[pos] & ((1u << TOK_BITS) - 1));
  };
  const int chunks3[3] = {chunk, (chunk + nchunks - 1) % nchunks, (chunk + 1) % nchunks};      // own, back, forward (:173-176)
  for (int j = tid; j < 3 * cs; j += 256) ktok[j] = tok_of(chunks3[j / cs] * cs + j % cs);
  const float* xb = xe + (long)n * L * Ce;
  const float* yb = ye + (long)n * L * Cy;
  const int K3 = 3 * cs;
  for (int q0 = 0; q0 < cs; q0 += QT) {
    const int nq = min(QT, cs - q0);
    __syncthreads();
    for (int i = tid; i < nq * CE_MAX; i += 256) {
      const int r = i / CE_MAX, e = i - r * CE_MAX;
      sq[i] = e < Ce ? xb[(long)ktok[q0 + r] * Ce + e] : 0.f;    // a chunk's queries are its own keys' rows, unnormalised
    }
    for (int kt = 0; kt < 3; ++kt) {
      __syncthreads();
      for (int i = tid; i < cs * CE_MAX; i += 256) {
        const int r = i / CE_MAX, e = i - r * CE_MAX;
        sk[r * KP + e] = e < Ce ? xb[(long)ktok[kt * cs + r] * Ce + e] : 0.f;
      }
      __syncthreads();
      for (int r = tid; r < cs; r += 256) {                      // F.normalize(p = 2, eps = 5e-5) of the key rows (:224)
        float s2 = 0.f;
        for (int e = 0; e < Ce; ++e) s2 += sk[r * KP + e] * sk[r * KP + e];
        const float f = 1.0f / fmaxf(sqrtf(s2), 5e-5f);
        for (int e = 0; e < Ce; ++e) sk[r * KP + e] *= f;
      }
      __syncthreads();
      const int Ce4 = (Ce + 3) & ~3;                              // (columns Ce .. Ce4 - 1 of both tiles are zero)
      for (int p = tid; p < nq * cs; p += 256) {
        const int i = p / cs, j = p - i * cs;
        float a = 0.f;
        for (int e = 0; e < Ce4; e += 4) {
          const f32x4 qv = *(const f32x4*)(sq + i * CE_MAX + e), kv = *(const f32x4*)(sk + j * KP + e);
          a += (qv.x * kv.x + qv.y * kv.y) + (qv.z * kv.z + qv.w * kv.w);
        }
        ss[i * K3 + kt * cs + j] = a;
      }
    }
    __syncthreads();
    // log-sum-exp of each row and the probabilities (:235-237): one wave per row
    for (int i = tid >> 6; i < nq; i += 4) {
      const int lane = tid & 63;
      float mx = -3.0e38f;
      for (int j = lane; j < K3; j += 64) mx = fmaxf(mx, ss[i * K3 + j]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < K3; j += 64) sum += expf(ss[i * K3 + j] - mx);
      sum = wave_sum(sum);
      const float lse = mx + logf(sum);
      for (int j = lane; j < K3; j += 64) ss[i * K3 + j] = expf(ss[i * K3 + j] - lse);
      if (lane == 0) rl[i] = lse;
    }
    __syncthreads();
    // O = P . V: thread = output channel, QT row accumulators
    for (int c = tid; c < Cy; c += 256) {
      float acc[QT];
#pragma unroll
      for (int i = 0; i < QT; ++i) acc[i] = 0.f;
      int j = 0;
      for (; j + 8 <= K3; j += 8) {                  // eight value rows in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = yb[(long)ktok[j + u] * Cy + c];
#pragma unroll
        for (int i = 0; i < QT; ++i) {
          const f32x4 p0 = *(const f32x4*)(ss + i * K3 + j), p1 = *(const f32x4*)(ss + i * K3 + j + 4);
          acc[i] += (p0.x * v[0] + p0.y * v[1]) + (p0.z * v[2] + p0.w * v[3]) + (p1.x * v[4] + p1.y * v[5]) +
                    (p1.z * v[6] + p1.w * v[7]);
        }
      }
      for (; j < K3; ++j) {
        const float v = yb[(long)ktok[j] * Cy + c];
#pragma unroll
        for (int i = 0; i < QT; ++i) acc[i] += ss[i * K3 + j] * v;
      }
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        const int pos = chunk * cs + q0 + i;
        if (i < nq && pos < L) ret[((long)nhh * L + ktok[q0 + i]) * Cy + c] = acc[i];
      }
    }
    for (int i = tid; i < nq; i += 256) {
      const int pos = chunk * cs + q0 + i;
      if (pos < L) score[(long)nhh * L + ktok[q0 + i]] = rl[i];
    }
  }
}

// The same block on the exact-f32 matrix core (v_mfma_f32_32x32x2_f32; Cy a multiple of 32 up to 256, cs a multiple of 8, Ce a
// multiple of 4), eight waves: 32 queries at a time; S = Q K^T as one 32 x 32 tile per wave (rows of pitch 68 = 4 * 17
// floats: every lane half reads four consecutive features of its row with one b128 and feeds them to four MFMAs, the same k
// permutation on both operands) with the keys' 1 / max(|k|, eps) -- computed once per block -- applied to the scores; the row
// softmax as before; O = P V, one 32-channel tile per wave, P from LDS (pitch 4 * odd) and the value rows straight from
// global memory in the B-operand layout (lane = channel, 128-byte segments), 32 keys (16 loads per lane) ahead.
constexpr int MQ = 32, MP = CE_MAX + 4, MW = 8, MTH = 64 * MW;
__global__ void __launch_bounds__(MTH) k_nlsa_attention_mfma(const float* __restrict__ xe, const float* __restrict__ ye,
                                                             const unsigned long long* __restrict__ order,
                                                             float* __restrict__ ret, float* __restrict__ score, int L, int Ce,
                                                             int Cy, int nh, int cs, int nchunks, int SP) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const sq = sm;                               // [MQ][MP]
  float* const sk = sq + MQ * MP;                     // [cs][MP] one key chunk
  float* const ss = sk + (size_t)cs * MP;             // [MQ][SP] scores, then probabilities
  int* const ktok = (int*)(ss + (size_t)MQ * SP);     // [3 cs] token of key j
  float* const kf = (float*)(ktok + 3 * cs);          // [3 cs] 1 / max(|k_j|, 5e-5)
  float* const rl = kf + 3 * cs;                      // [MQ] log-sum-exp of the row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int chunk = blockIdx.x % nchunks;
  const int nhh = blockIdx.x / nchunks;               // n * nh + h
  const int n = nhh / nh;
  const unsigned long long* ord = order + (long)nhh * L;
  const int padding = L % cs ? cs - L % cs : 0;
  auto tok_of = [&](int pos) {
    if (pos >= L) pos -= padding;
    return (int)(ord[pos] & ((1u << TOK_BITS) - 1));
  };
  const int chunks3[3] = {chunk, (chunk + nchunks - 1) % nchunks, (chunk + 1) % nchunks};      // own, back, forward (:173-176)
  const float* xb = xe + (long)n * L * Ce;
  const float* yb = ye + (long)n * L * Cy;
  const int K3 = 3 * cs, C4 = Ce >> 2, G8 = (Ce + 7) >> 3, nct = (cs + 31) >> 5, ncy = Cy >> 5;
  for (int j = tid; j < K3; j += MTH) {               // F.normalize(p = 2, eps = 5e-5) of the key rows (:224), as a factor
    const int tok = tok_of(chunks3[j / cs] * cs + j % cs);
    ktok[j] = tok;
    const f32x4* row = (const f32x4*)(xb + (long)tok * Ce);
    float s2 = 0.f;
    for (int e = 0; e < C4; ++e) {
      const f32x4 v = row[e];
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    kf[j] = 1.0f / fmaxf(sqrtf(s2), 5e-5f);
  }
  for (int q0 = 0; q0 < cs; q0 += MQ) {
    const int nq = min(MQ, cs - q0);
    __syncthreads();
    for (int i = tid; i < MQ * (CE_MAX / 4); i += MTH) {           // a chunk's queries are its own keys' rows, unnormalised
      const int rr = i / (CE_MAX / 4), e = i - rr * (CE_MAX / 4);
      *(f32x4*)(sq + rr * MP + 4 * e) = (rr < nq && e < C4) ? *(const f32x4*)(xb + (long)ktok[q0 + rr] * Ce + 4 * e)
                                                            : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int kt = 0; kt < 3; ++kt) {
      __syncthreads();
      for (int i = tid; i < cs * (CE_MAX / 4); i += MTH) {
        const int rr = i / (CE_MAX / 4), e = i - rr * (CE_MAX / 4);
        *(f32x4*)(sk + rr * MP + 4 * e) = e < C4 ? *(const f32x4*)(xb + (long)ktok[kt * cs + rr] * Ce + 4 * e)
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      __syncthreads();
      for (int ct = wave; ct < nct; ct += MW) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const float* ap = sq + r * MP + 4 * h;
        const float* bp = sk + min(ct * 32 + r, cs - 1) * MP + 4 * h;
        for (int g = 0; g < G8; ++g) {
          const f32x4 fa = *(const f32x4*)(ap + g * 8), fb = *(const f32x4*)(bp + g * 8);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = mfma32(fa[t], fb[t], acc);
        }
        const int col = ct * 32 + r;
        if (col < cs) {
          const float f = kf[kt * cs + col];
#pragma unroll
          for (int q = 0; q < 16; ++q) ss[mfma_row(q, lane) * SP + kt * cs + col] = acc[q] * f;
        }
      }
    }
    __syncthreads();
    // log-sum-exp of each row and the probabilities (:235-237): one wave per row
    for (int i = wave; i < nq; i += MW) {
      float mx = -3.0e38f;
      for (int j = lane; j < K3; j += 64) mx = fmaxf(mx, ss[i * SP + j]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < K3; j += 64) sum += expf(ss[i * SP + j] - mx);
      sum = wave_sum(sum);
      const float lse = mx + logf(sum);
      for (int j = lane; j < K3; j += 64) ss[i * SP + j] = expf(ss[i * SP + j] - lse);
      if (lane == 0) rl[i] = lse;
    }
    __syncthreads();
    // O = P . V: wave = 32 output channels
    if (wave < ncy) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const float* pp = ss + r * SP + 4 * h;
      const float* ycol = yb + wave * 32 + r;
      const int NG = K3 >> 3;                          // groups of 8 keys: this lane half's four, then the other's
      float va[4][4], vb[4][4];
      auto fetch = [&](int g0, float (&v)[4][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = min((g0 + u) * 8 + 4 * h + t, K3 - 1);
            v[u][t] = ycol[(long)ktok[j] * Cy];
          }
      };
      auto run = [&](int g0, const float (&v)[4][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (g0 + u < NG) {
            const f32x4 fa = *(const f32x4*)(pp + (g0 + u) * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = mfma32(fa[t], v[u][t], acc);
          }
        }
      };
      fetch(0, va);
      for (int g = 0; g < NG; g += 8) {
        if (g + 4 < NG) fetch(g + 4, vb);
        run(g, va);
        if (g + 8 < NG) fetch(g + 8, va);
        if (g + 4 < NG) run(g + 4, vb);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = mfma_row(q, lane);
        const int pos = chunk * cs + q0 + i;
        if (i < nq && pos < L) ret[((long)nhh * L + ktok[q0 + i]) * Cy + wave * 32 + r] = acc[q];
      }
    }
    for (int i = tid; i < nq; i += MTH) {
      const int pos = chunk * cs + q0 + i;
      if (pos < L) score[(long)nhh * L + ktok[q0 + i]] = rl[i];
    }
  }
}

// softmax over the rounds of the scores, weighted sum, residual (:256-266): one wave per token
__global__ void __launch_bounds__(256) k_nlsa_combine(const float* __restrict__ ret, const float* __restrict__ score,
                                                      const float* __restrict__ x, float* __restrict__ out, int N, int L,
                                                      int Cy, int nh, float res_scale) {
  const long t = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= (long)N * L) return;
  const int n = (int)(t / L), l = (int)(t % L);
  float mx = -3.0e38f;
  for (int h = 0; h < nh; ++h) mx = fmaxf(mx, score[((long)n * nh + h) * L + l]);
  float den = 0.f;
  for (int h = 0; h < nh; ++h) den += expf(score[((long)n * nh + h) * L + l] - mx);
  for (int c = lane; c < Cy; c += 64) {
    float a = 0.f;
    for (int h = 0; h < nh; ++h)
      a += ret[(((long)n * nh + h) * L + l) * Cy + c] * (expf(score[((long)n * nh + h) * L + l] - mx) / den);
    out[t * Cy + c] = a * res_scale + x[t * Cy + c];
  }
}

}  // namespace

extern "C" {

/* hash buckets of a sample with L tokens (network_nlsn.py:193-194) */
int srhip_nlsa_hash_buckets(int L, int chunk_size) {
  const int q = L / chunk_size;
  const int hb = q + q % 2;
  return hb < 128 ? hb : 128;
}

/* the in-tree counting sort keeps its histograms in LDS: no global workspace (the query stays in the ABI; callers may pass
 * any non-NULL pointer) */
long srhip_nlsa_sort_ws(long n_items) { (void)n_items; return 16; }

/* rotated [N*L][ld] = x_embed . rotations ([.., n_hashes * hb/2] columns, round-major) -> order [N][n_hashes][L]: the
 * tokens of every (sample, round) by hash code, ties by token index (64-bit keys, low 20 bits = token). */
int srhip_nlsa_order(const float* rotated, long ld, unsigned long long* keys_tmp, unsigned long long* order, void* workspace,
                     long ws_bytes, int N, int L, int n_hashes, int hash_buckets, void* stream) {
  SR_REQUIRE(rotated && keys_tmp && order && workspace, "nlsa_order: null operand");
  SR_REQUIRE(N > 0 && n_hashes > 0 && L > 0 && L < (1 << TOK_BITS) && hash_buckets >= 2 && hash_buckets % 2 == 0 &&
             hash_buckets <= BO_HB, "nlsa_order: L = %d (< 2^20), hash_buckets = %d (even, 2 ... 128)", L, hash_buckets);
  const long items = (long)N * n_hashes * L;
  SR_REQUIRE(items < (1L << 31), "nlsa_order: %ld items", items);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lsh_keys, dim3(sr_cdiv(items, 256)), dim3(256), 0, st, rotated, ld, keys_tmp, N, L, n_hashes,
                     hash_buckets / 2);
  (void)ws_bytes;
  hipLaunchKernelGGL(k_bucket_order, dim3(N * n_hashes), dim3(BO_T), 0, st, keys_tmp, order, L, hash_buckets);
  SR_LAUNCH_CHECK("nlsa_order");
  return 0;
}

/* ret [N][n_hashes][L][Cy], score [N][n_hashes][L] (token positions) from x_embed [N*L][Ce], y_embed [N*L][Cy] and the
 * order of srhip_nlsa_order; then out [N*L][Cy] = x + res_scale * sum_h softmax_h(score) ret_h. */
int srhip_nlsa_attention(const float* x_embed, const float* y_embed, const unsigned long long* order, float* ret, float* score,
                         const float* x, float* out, int N, int L, int Ce, int Cy, int n_hashes, int chunk_size,
                         float res_scale, void* stream) {
  SR_REQUIRE(x_embed && y_embed && order && ret && score && x && out, "nlsa_attention: null operand");
  SR_REQUIRE(Ce > 0 && Ce <= CE_MAX && Cy > 0 && chunk_size > 0 && chunk_size % 4 == 0 && L >= chunk_size,
             "nlsa_attention: Ce = %d (<= 64), chunk_size = %d (a multiple of 4, <= L = %d)", Ce, chunk_size, L);
  const int nchunks = sr_cdiv(L, chunk_size);
  hipStream_t st = (hipStream_t)stream;
  if (Cy % 32 == 0 && Cy <= 32 * MW && chunk_size % 8 == 0 && Ce % 4 == 0) {            // the matrix-core kernel
    const int K3 = 3 * chunk_size, SP = K3 + (12 - K3 % 8) % 8;      // pitch = 4 * odd floats
    const size_t lds = ((size_t)MQ * MP + (size_t)chunk_size * MP + (size_t)MQ * SP + 6 * chunk_size + MQ) * 4;
    SR_REQUIRE(lds <= 160 * 1024, "nlsa_attention: chunk_size %d needs %zu bytes of LDS", chunk_size, lds);
    static size_t reserved_m = 0;
    if (lds > reserved_m) {
      if (hipFuncSetAttribute((const void*)k_nlsa_attention_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sr_fail(-5, "nlsa_attention: cannot reserve %zu bytes of LDS", lds);
      reserved_m = lds;
    }
    hipLaunchKernelGGL(k_nlsa_attention_mfma, dim3(N * n_hashes * nchunks), dim3(MTH), lds, st, x_embed, y_embed, order, ret, score,
                       L, Ce, Cy, n_hashes, chunk_size, nchunks, SP);
  } else {
    const size_t lds = ((size_t)QT * CE_MAX + (size_t)chunk_size * KP + (size_t)QT * 3 * chunk_size + 3 * chunk_size + QT) * 4;
    SR_REQUIRE(lds <= 160 * 1024, "nlsa_attention: chunk_size %d needs %zu bytes of LDS", chunk_size, lds);
    static size_t reserved = 0;
    if (lds > reserved) {
      if (hipFuncSetAttribute((const void*)k_nlsa_attention, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sr_fail(-5, "nlsa_attention: cannot reserve %zu bytes of LDS", lds);
      reserved = lds;
    }
    hipLaunchKernelGGL(k_nlsa_attention, dim3(N * n_hashes * nchunks), dim3(256), lds, st, x_embed, y_embed, order, ret, score, L,
                       Ce, Cy, n_hashes, chunk_size, nchunks);
  }
  hipLaunchKernelGGL(k_nlsa_combine, dim3(sr_cdiv((long)N * L, 4)), dim3(256), 0, st, ret, score, x, out, N, L, Cy, n_hashes,
                     res_scale);
  SR_LAUNCH_CHECK("nlsa_attention");
  return 0;
}

}  // extern "C"
