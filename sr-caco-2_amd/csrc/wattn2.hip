// Window / shifted-window attention core on the two-plane fp16 split MFMA (three products, block exponents) --
// WindowAttention.forward's q@k^T, +bias, +mask, softmax, attn@v and the roll / window_partition /
// window_reverse around it (dlib/models/network_swinir.py:48-80,153-176,297-331), window 8x8, head dim <= 32.
//
// Why: on the exact-f32 MFMA (wattn.hip) the two contractions of a (window, head) cost 8192 matrix-core cycles;
// as three fp16 products on v_mfma_f32_16x16x32_f16 they cost 1536 -- and the operand forms below need NO LDS:
//
//   * a row-form fragment F(X, T): lane (r, g) holds X[16 T + r][8 g .. 8 g + 7] -- 32 contiguous bytes of the
//     token's qkv row, four 8-byte loads; the block exponent of a row comes out of its four lanes (two shuffles);
//   * S^T tile (J, I) = F(K, J) x F(Q, I): lane (c, g) holds the scores of QUERY 16 I + c against keys
//     16 J + 4 g .. + 3 -- a softmax row is 16 registers of a lane plus its three partner lanes (two shuffles);
//   * those registers ARE the B operand of O^T = V^T . P^T: the contraction index (the key) may run in any order
//     as long as both operands agree, so the k octet of lane (c, g) is defined as the keys it already holds
//     (32 JJ + 4 g + t and 32 JJ + 16 + 4 g + t), and the V^T operand is gathered in that order straight from
//     global memory; P never moves;
//   * O^T tile (jd, I) leaves lane (c, g) with four consecutive head-dim entries of one query: 8-byte stores.
//
// f32-grade results: q, k rows and v columns carry a power-of-two scale (max in [8192, 16384)), p in (0, 1] a
// fixed 2^14; sums in f32.
#include "common.h"
#include "kernels.h"
#include "wattn2_dev.h"

namespace {

template <int D>
__global__ void __launch_bounds__(256, 3) k_wattn2_fwd(const float* __restrict__ qkv, float* __restrict__ out,
                                                      const float* __restrict__ biasF, long total, int H, int W,
                                                      int C, int heads, int shift, float scale) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long gid = sr_xcd_block(blockIdx.x, gridDim.x) * 4L + wv;
  if (gid >= total) return;                 // waves are independent: no block-level barrier below
  const W2Geom geo = w2_decode(gid, heads, W / 8, H / 8, shift);
  w2_fwd_body<D>(qkv, out, biasF, geo, H, W, C, shift, scale, lane);
}

// column-form operand, gathered: lane (r, g) = head-dim entry 16 jd + r of the rows at positions w2_kpos(JJ, g, 0..7)
// (the k order of the registers of an S^T / S tile pair), under ONE power-of-two scale per head-dim column (over all 64
// rows); rinv[jd] = 2^-s of column 16 jd + r
template <int D>
__device__ __forceinline__ void w2_gather_cols(const float* __restrict__ base, long pitch, const W2Geom& geo, int H, int W,
                                               int shift, int c, int g, u32x4 (&hi)[2][2], u32x4 (&lo)[2][2],
                                               float (&rinv)[2]) {
  float raw[2][2][8];
#pragma unroll
  for (int JJ = 0; JJ < 2; ++JJ)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int tk = w2_token(geo, w2_kpos(JJ, g, t), H, W, shift);
      const float* p = base + (long)tk * pitch + c;
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) raw[JJ][jd][t] = (16 * jd + c < D) ? ldg_f(p + 16 * jd) : 0.f;
    }
#pragma unroll
  for (int jd = 0; jd < 2; ++jd) {
    float mx = 0.f;
#pragma unroll
    for (int JJ = 0; JJ < 2; ++JJ)
#pragma unroll
      for (int t = 0; t < 8; ++t) mx = fmaxf(mx, fabsf(raw[JJ][jd][t]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float sc = pow2_scale(mx);
    rinv[jd] = pow2_inv(sc);
#pragma unroll
    for (int JJ = 0; JJ < 2; ++JJ) {
      unsigned h[4], l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) split2_pair(raw[JJ][jd][2 * t] * sc, raw[JJ][jd][2 * t + 1] * sc, h[t], l[t]);
      hi[JJ][jd] = u32x4{h[0], h[1], h[2], h[3]};
      lo[JJ][jd] = u32x4{l[0], l[1], l[2], l[3]};
    }
  }
}
// B operand made of the registers of a tile pair: slots (2t, 2t+1) = tile 2 JJ + (t >> 1), registers 2 (t & 1), + 1
__device__ __forceinline__ void w2_pack_pair(const f32x4& t0, const f32x4& t1, float sc, u32x4& hi, u32x4& lo) {
  unsigned h[4], l[4];
  split2_pair(t0[0] * sc, t0[1] * sc, h[0], l[0]);
  split2_pair(t0[2] * sc, t0[3] * sc, h[1], l[1]);
  split2_pair(t1[0] * sc, t1[1] * sc, h[2], l[2]);
  split2_pair(t1[2] * sc, t1[3] * sc, h[3], l[3]);
  hi = u32x4{h[0], h[1], h[2], h[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x4 mfma3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x4 acc) {
  acc = mfma16h(ah, bl, acc);
  acc = mfma16h(al, bh, acc);
  return mfma16h(ah, bh, acc);
}

// Backward of the attention core, ONE kernel, one wave per (window, head); a block = four consecutive windows of one head.
//   query side (per query tile I): S^T and dP^T = V . dO^T against all keys -> softmax row, delta = sum_k P dP,
//     dS = P (dP - delta); dS is this lane's B operand of dQ^T = K^T . dS^T (K gathered column-form) and goes to the
//     wave's LDS tile for the bias gradient; the row's log-sum-exp and delta are kept in LDS for the key side;
//   key side (per key tile J): S and dP in the OTHER orientation (lane = key, registers = queries; P re-made from the
//     log-sum-exp) are the B operands of dV^T = dO^T . P and dK^T = Q^T . dS (Q, dO gathered column-form).
// Every operand is read from global memory in the form the matrix core takes it (the re-reads of the key side hit L1 /
// L2); nothing but the bias-gradient tile and 2 KB of row statistics goes through LDS.  The block's four dS tiles are
// added in wave order and stored as ONE partial tile (plain stores): k_dbias2_reduce sums the partials in fp64.
// phase timestamps of every wave (tools/mb_attn_phases.py): experiment builds only
#ifdef SRHIP_EXPERIMENTS
long long* g_w2_dbg = nullptr;
#define SR_TS(K) \
  if (dbg && lane == 0) dbg[((long)blockIdx.x * 4 + wv) * 16 + (K)] = (long long)wall_clock64();
#else
#define SR_TS(K)
#endif

constexpr int W2_DS = 4096;                        // floats of a dS tile
constexpr int W2_ST = 6 * 64;                      // per wave: lse, delta, 2^-s of the Q rows (x scale), of the dO, K and V rows
template <int D>
__global__ void __launch_bounds__(256, 2) k_wattn2_bwd(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, const float* __restrict__ biasF,
                                                      const float* __restrict__ biasG, float* __restrict__ part,
                                                      int nwin, int H, int W, int C, int heads, int shift, float scale,
                                                      long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float w2s[];
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  SR_TS(0)
  float* const dsl = w2s + wv * W2_DS;
  float* const stl = w2s + 4 * W2_DS + wv * W2_ST;
  const int lb = sr_xcd_block(blockIdx.x, gridDim.x);
  const int head = lb % heads, wg = lb / heads;
  const int widx = wg * 4 + wv;
  if (widx < nwin) {
    W2Geom geo;
    const int nWx = W / 8, nWy = H / 8;
    geo.head = head;
    geo.wx = widx % nWx;
    geo.wy = (widx / nWx) % nWy;
    geo.b = widx / (nWx * nWy);
    geo.last_row = shift > 0 && geo.wy == nWy - 1;
    geo.last_col = shift > 0 && geo.wx == nWx - 1;
    const long C3 = 3L * C;
    const float* qb = qkv + head * D;
    const float* gb = dout + head * D;
    float* dqb = dqkv + head * D;
    int tok[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) tok[T] = w2_token(geo, 16 * T + c, H, W, shift);
    const bool lane_masked = geo.last_col && (((c >> 2) & 1) != (g & 1));

    // ================= query side
    {
      u32x4 kh[4], kl[4], vh[4], vl[4];
      {
        float raw[4][8], rk[4], rv[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], qb + C, C3, tok[T], g);
#pragma unroll
        for (int T = 0; T < 4; ++T) rk[T] = w2_split_row(raw[T], kh[T], kl[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], qb + 2 * C, C3, tok[T], g);
#pragma unroll
        for (int T = 0; T < 4; ++T) rv[T] = w2_split_row(raw[T], vh[T], vl[T]);
        if (g == 0) {
#pragma unroll
          for (int T = 0; T < 4; ++T) { stl[256 + 16 * T + c] = rk[T]; stl[320 + 16 * T + c] = rv[T]; }
        }
        __builtin_amdgcn_wave_barrier();
      }
      SR_TS(1)
      u32x4 kth[2][2], ktl[2][2];
      float rkt[2];
      w2_gather_cols<D>(qb + C, C3, geo, H, W, shift, c, g, kth, ktl, rkt);
      SR_TS(2)
      const float* bimg = biasF + (long)head * 4096;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        float raw[8];
        u32x4 qh, ql, gh, gl;
        w2_load_row<D>(raw, qb, C3, tok[I], g);
        const float rq = w2_split_row(raw, qh, ql);
        w2_load_row<D>(raw, gb, C, tok[I], g);
        const float rg = w2_split_row(raw, gh, gl);
        if (g == 0) { stl[128 + 16 * I + c] = rq * scale; stl[192 + 16 * I + c] = rg; }
        f32x4 S[4], P[4];
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          S[J] = mfma3(kh[J], kl[J], qh, ql, f32x4{0.f, 0.f, 0.f, 0.f});
          P[J] = mfma3(vh[J], vl[J], gh, gl, f32x4{0.f, 0.f, 0.f, 0.f});      // dP^T
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const f32x4 bv = *(const f32x4*)(bimg + w2_img_index(I, J, lane));
          const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
          const f32x4 rkk = *(const f32x4*)(stl + 256 + 16 * J + 4 * g), rvv = *(const f32x4*)(stl + 320 + 16 * J + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float sv = S[J][e] * (rq * scale * rkk[e]) + bv[e];
            sv += masked ? -100.f : 0.f;
            S[J][e] = sv;
            mx = fmaxf(mx, sv);
            P[J][e] *= rg * rvv[e];
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = __expf(S[J][e] - mx);
            S[J][e] = pe;
            sum += pe;
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
        float dl = 0.f;
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            S[J][e] *= inv;
            dl += S[J][e] * P[J][e];
          }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0) { stl[16 * I + c] = mx + __logf(sum); stl[64 + 16 * I + c] = dl; }
        float dmx = 0.f;
#pragma unroll
        for (int J = 0; J < 4; ++J) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dsv = S[J][e] * (P[J][e] - dl);
            S[J][e] = dsv;
            dmx = fmaxf(dmx, fabsf(dsv));
          }
          *(f32x4*)(dsl + w2_img_index(I, J, lane)) = S[J];
        }
        dmx = fmaxf(dmx, __shfl_xor(dmx, 16, 64));
        dmx = fmaxf(dmx, __shfl_xor(dmx, 32, 64));
        const float dsc = pow2_scale(dmx), dri = pow2_inv(dsc);
        f32x4 O[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int JJ = 0; JJ < 2; ++JJ) {
          u32x4 dh, dlo;
          w2_pack_pair(S[2 * JJ], S[2 * JJ + 1], dsc, dh, dlo);
#pragma unroll
          for (int jd = 0; jd < 2; ++jd) O[jd] = mfma3(kth[JJ][jd], ktl[JJ][jd], dh, dlo, O[jd]);
        }
        float* op = dqb + (long)tok[I] * C3 + 4 * g;
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          const int d0 = 16 * jd + 4 * g;
          float sc4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) sc4[e] = __shfl(rkt[jd], 4 * g + e, 64) * (scale * dri);     // 2^-s of column 16 jd + 4 g + e
          if (d0 < D) *(float2*)(op + 16 * jd) = float2{O[jd][0] * sc4[0], O[jd][1] * sc4[1]};
          if (d0 + 2 < D) *(float2*)(op + 16 * jd + 2) = float2{O[jd][2] * sc4[2], O[jd][3] * sc4[3]};
        }
        SR_TS(3 + I)
      }
    }
    __builtin_amdgcn_wave_barrier();      // the row statistics written by lanes g = 0 are read by every lane below

    // ================= key side
    {
      u32x4 qh[4], ql[4], gh[4], gl[4];
      {
        float raw[4][8];
#pragma unroll
        for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], qb, C3, tok[T], g);
#pragma unroll
        for (int T = 0; T < 4; ++T) (void)w2_split_row(raw[T], qh[T], ql[T]);
#pragma unroll
        for (int T = 0; T < 4; ++T) w2_load_row<D>(raw[T], gb, C, tok[T], g);
#pragma unroll
        for (int T = 0; T < 4; ++T) (void)w2_split_row(raw[T], gh[T], gl[T]);
      }
      SR_TS(7)
      u32x4 qth[2][2], qtl[2][2], gth[2][2], gtl[2][2];
      float rqt[2], rgt[2];
      w2_gather_cols<D>(qb, C3, geo, H, W, shift, c, g, qth, qtl, rqt);
      w2_gather_cols<D>(gb, C, geo, H, W, shift, c, g, gth, gtl, rgt);
      SR_TS(8)
      const float* bimg = biasG + (long)head * 4096;
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        float raw[8];
        u32x4 kh, kl, vh, vl;
        w2_load_row<D>(raw, qb + C, C3, tok[J], g);
        const float rk = w2_split_row(raw, kh, kl);
        w2_load_row<D>(raw, qb + 2 * C, C3, tok[J], g);
        const float rv = w2_split_row(raw, vh, vl);
        f32x4 S[4], P[4];      // lane (c, g): key 16 J + c against queries 16 I + 4 g + e
        float dmx = 0.f;
#pragma unroll
        for (int I = 0; I < 4; ++I) {
          S[I] = mfma3(qh[I], ql[I], kh, kl, f32x4{0.f, 0.f, 0.f, 0.f});
          P[I] = mfma3(gh[I], gl[I], vh, vl, f32x4{0.f, 0.f, 0.f, 0.f});
          const f32x4 bv = *(const f32x4*)(bimg + w2_img_index(J, I, lane));
          const f32x4 lse = *(const f32x4*)(stl + 16 * I + 4 * g), dl = *(const f32x4*)(stl + 64 + 16 * I + 4 * g);
          const f32x4 rq = *(const f32x4*)(stl + 128 + 16 * I + 4 * g), rg = *(const f32x4*)(stl + 192 + 16 * I + 4 * g);
          const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float sv = S[I][e] * (rq[e] * rk) + bv[e];
            sv += masked ? -100.f : 0.f;
            const float pe = __expf(sv - lse[e]);
            const float dsv = pe * (P[I][e] * (rg[e] * rv) - dl[e]);
            P[I][e] = pe;
            S[I][e] = dsv;
            dmx = fmaxf(dmx, fabsf(dsv));
          }
        }
        dmx = fmaxf(dmx, __shfl_xor(dmx, 16, 64));
        dmx = fmaxf(dmx, __shfl_xor(dmx, 32, 64));
        const float dsc = pow2_scale(dmx), dri = pow2_inv(dsc);
        f32x4 OV[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 OK[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int II = 0; II < 2; ++II) {
          u32x4 ph, pl, dh, dlo;
          w2_pack_pair(P[2 * II], P[2 * II + 1], 16384.f, ph, pl);
          w2_pack_pair(S[2 * II], S[2 * II + 1], dsc, dh, dlo);
#pragma unroll
          for (int jd = 0; jd < 2; ++jd) {
            OV[jd] = mfma3(gth[II][jd], gtl[II][jd], ph, pl, OV[jd]);
            OK[jd] = mfma3(qth[II][jd], qtl[II][jd], dh, dlo, OK[jd]);
          }
        }
        float* op = dqb + (long)tok[J] * C3 + 4 * g;
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          const int d0 = 16 * jd + 4 * g;
          float sk[4], sv4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sk[e] = __shfl(rqt[jd], 4 * g + e, 64) * (scale * dri);
            sv4[e] = __shfl(rgt[jd], 4 * g + e, 64) * (1.0f / 16384.f);
          }
          if (d0 < D) {
            *(float2*)(op + C + 16 * jd) = float2{OK[jd][0] * sk[0], OK[jd][1] * sk[1]};
            *(float2*)(op + 2 * C + 16 * jd) = float2{OV[jd][0] * sv4[0], OV[jd][1] * sv4[1]};
          }
          if (d0 + 2 < D) {
            *(float2*)(op + C + 16 * jd + 2) = float2{OK[jd][2] * sk[2], OK[jd][3] * sk[3]};
            *(float2*)(op + 2 * C + 16 * jd + 2) = float2{OV[jd][2] * sv4[2], OV[jd][3] * sv4[3]};
          }
        }
        SR_TS(9 + J)
      }
    }
  } else {
    for (int i = lane; i < W2_DS / 4; i += 64) ((f32x4*)dsl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (part) {       // the block's partial bias-gradient tile: the four waves' tiles in wave order (deterministic)
    __syncthreads();
    float* dst = part + ((long)wg * heads + head) * W2_DS;
    for (int i = threadIdx.x; i < W2_DS / 4; i += 256) {
      const f32x4 a = ((const f32x4*)w2s)[i], b = ((const f32x4*)(w2s + W2_DS))[i];
      const f32x4 cc = ((const f32x4*)(w2s + 2 * W2_DS))[i], d = ((const f32x4*)(w2s + 3 * W2_DS))[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (a[e] + b[e]) + (cc[e] + d[e]);
      ((f32x4*)dst)[i] = o;
    }
  }
  SR_TS(13)
}

// partial tiles [nparts][heads][4096] (accumulator order of k_wattn2_bwd) -> the bias-gradient image in the order
// srhip_bias_grad reads (wa_dimg_index, wattn.hip: [a][b][q][lane] of the 32x32 tiles, key = mfma_row(q, lane) + 32 a,
// query = (lane & 31) + 32 b); fp64 sums in partial order
// (blockIdx.y = attention block of a batched call: its partials at part + y * part_stride, its image at dimg + y * img_stride)
__global__ void __launch_bounds__(256) k_dbias2_reduce(const float* __restrict__ part, int nparts, int heads,
                                                       float* __restrict__ dimg, long part_stride, long img_stride) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= heads * 4096) return;
  part += (long)blockIdx.y * part_stride;
  dimg += (long)blockIdx.y * img_stride;
  const int hd = o >> 12, el = o & 4095;
  const int ln = el & 63, q = (el >> 6) & 15, b = (el >> 10) & 1, a = el >> 11;
  const int key = mfma_row(q, ln) + 32 * a, query = (ln & 31) + 32 * b;
  const int mine = w2_img_index(query >> 4, key >> 4, 16 * ((key >> 2) & 3) + (query & 15)) + (key & 3);
  const float* p = part + (long)hd * W2_DS + mine;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = 0;
  for (; i + 4 <= nparts; i += 4) {
    s0 += (double)p[(long)(i + 0) * heads * W2_DS]; s1 += (double)p[(long)(i + 1) * heads * W2_DS];
    s2 += (double)p[(long)(i + 2) * heads * W2_DS]; s3 += (double)p[(long)(i + 3) * heads * W2_DS];
  }
  for (; i < nparts; ++i) s0 += (double)p[(long)i * heads * W2_DS];
  dimg[o] = (float)((s0 + s1) + (s2 + s3));
}

// bias table (225, heads) -> images in the accumulator orders of k_wattn2_*:
//   imgF[head][I][J][lane][e] = bias(query 16 I + c, key 16 J + 4 g + e)      (S^T tiles: forward, query side)
//   imgG[head][J][I][lane][e] = bias(query 16 I + 4 g + e, key 16 J + c)      (S tiles: key side of the backward)
__global__ void k_bias_expand2(const float* __restrict__ table, float* __restrict__ imgF, float* __restrict__ imgG,
                               int heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * 4096) return;
  const int hd = i >> 12, rem = i & 4095;
  const int e = rem & 3, lane = (rem >> 2) & 63, t1 = (rem >> 8) & 3, t0 = rem >> 10;
  auto rpi = [](int query, int key) { return ((query >> 3) - (key >> 3) + 7) * 15 + ((query & 7) - (key & 7) + 7); };
  imgF[i] = ldg_f(table + rpi(16 * t0 + (lane & 15), 16 * t1 + 4 * (lane >> 4) + e) * heads + hd);
  if (imgG) imgG[i] = ldg_f(table + rpi(16 * t1 + 4 * (lane >> 4) + e, 16 * t0 + (lane & 15)) * heads + hd);
}

}  // namespace

extern "C" {

int srhip_bias_expand_f16x2(const float* table, float* biasF, float* biasG, int heads, void* stream) {
  SR_REQUIRE(table && biasF, "bias_expand_f16x2: null operand");
  hipLaunchKernelGGL(k_bias_expand2, dim3(sr_cdiv(heads * 4096, 256)), dim3(256), 0, (hipStream_t)stream, table,
                     biasF, biasG, heads);
  SR_LAUNCH_CHECK("bias_expand_f16x2");
  return 0;
}

static int w2_check(int B, int H, int W, int C, int heads, int shift) {
  SR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
             "window_attention_f16x2: H, W must be positive multiples of the 8x8 window (H=%d W=%d)", H, W);
  SR_REQUIRE(heads > 0 && C % heads == 0, "window_attention_f16x2: C %% heads != 0");
  SR_REQUIRE(shift == 0 || shift == 4, "window_attention_f16x2: shift must be 0 or 4 (got %d)", shift);
  SR_REQUIRE(shift == 0 || (H > 8 && W > 8), "window_attention_f16x2: shifted windows need H, W > 8");
  const int D = C / heads;
  SR_REQUIRE(D == 30 || D == 10 || D == 16 || D == 32, "window_attention_f16x2: head dim %d not built", D);
  SR_REQUIRE(C % 2 == 0, "window_attention_f16x2: C must be even (8-byte row accesses)");
  return 0;
}

int srhip_window_attention_fwd_f16x2(const float* qkv, float* out, const float* biasF, int B, int H, int W, int C,
                                     int heads, int shift, void* stream) {
  if (int rc = w2_check(B, H, W, C, heads, shift)) return rc;
  const int D = C / heads;
  const long total = (long)B * (H / 8) * (W / 8) * heads;
  const float scale = 1.0f / sqrtf((float)D);
  dim3 grid(sr_cdiv(total, 4)), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define SR_WA(D_) \
  if (D == D_) hipLaunchKernelGGL((k_wattn2_fwd<D_>), grid, blk, 0, st, qkv, out, biasF, total, H, W, C, heads, shift, scale);
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  SR_LAUNCH_CHECK("window_attention_fwd_f16x2");
  return 0;
}

#ifdef SRHIP_EXPERIMENTS
int srhip_wattn2_debug_buffer(long long* buf) { g_w2_dbg = buf; return 0; }      // [blocks][4][16] wall-clock stamps
#endif

long srhip_window_attention_bwd_f16x2_ws(int B, int H, int W, int heads) {
  const int nwin = B * (H / 8) * (W / 8);
  return (long)sr_cdiv(nwin, 4) * heads * W2_DS;        // one partial bias-gradient tile per block
}

// dbiasT (may be NULL) is overwritten with the bias-gradient image in the order srhip_bias_grad reads.  dbiasT NULL with a
// workspace: the partial tiles are written and left for srhip_window_attention_dbias_reduce_f16x2.
int srhip_window_attention_bwd_f16x2(const float* qkv, const float* dout, float* dqkv, const float* biasF,
                                     const float* biasG, float* dbiasT, float* workspace, int B, int H, int W, int C,
                                     int heads, int shift, void* stream) {
  if (int rc = w2_check(B, H, W, C, heads, shift)) return rc;
  SR_REQUIRE(!dbiasT || workspace, "window_attention_bwd_f16x2: workspace required for the bias gradient");
  const int D = C / heads;
  const int nwin = B * (H / 8) * (W / 8), nparts = sr_cdiv(nwin, 4);
  const float scale = 1.0f / sqrtf((float)D);
  hipStream_t st = (hipStream_t)stream;
  constexpr int LDS = (4 * W2_DS + 4 * W2_ST) * 4;
  dim3 grid(nparts * heads), blk(256);
  float* part = workspace;
  long long* dbgp = nullptr;
#ifdef SRHIP_EXPERIMENTS
  dbgp = g_w2_dbg;
#endif
#define SR_WA(D_)                                                                                                    \
  if (D == D_) {                                                                                                     \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute((const void*)k_wattn2_bwd<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
        return sr_fail(-5, "window_attention_bwd_f16x2: cannot reserve %d bytes of LDS", LDS);                       \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((k_wattn2_bwd<D_>), grid, blk, LDS, st, qkv, dout, dqkv, biasF, biasG, part, nwin, H, W, C,   \
                       heads, shift, scale, dbgp);                                                                   \
  }
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  if (dbiasT)
    hipLaunchKernelGGL(k_dbias2_reduce, dim3(sr_cdiv(heads * 4096, 256)), dim3(256), 0, st, part, nparts, heads, dbiasT, 0L, 0L);
  SR_LAUNCH_CHECK("window_attention_bwd_f16x2");
  return 0;
}

// The bias-gradient images of nblocks attention blocks (same B, H, W, heads) from the partial tiles their backward
// launches left at workspace + i * ws_stride, in one launch: dbiasT + i * img_stride.
int srhip_window_attention_dbias_reduce_f16x2(const float* workspace, long ws_stride, int nblocks, float* dbiasT,
                                              long img_stride, int B, int H, int W, int heads, void* stream) {
  SR_REQUIRE(workspace && dbiasT && nblocks > 0 && nblocks <= 65535, "window_attention_dbias_reduce_f16x2: bad arguments");
  const int nparts = sr_cdiv(B * (H / 8) * (W / 8), 4);
  hipLaunchKernelGGL(k_dbias2_reduce, dim3(sr_cdiv(heads * 4096, 256), nblocks), dim3(256), 0, (hipStream_t)stream,
                     workspace, nparts, heads, dbiasT, ws_stride, img_stride);
  SR_LAUNCH_CHECK("window_attention_dbias_reduce_f16x2");
  return 0;
}

}  // extern "C"
