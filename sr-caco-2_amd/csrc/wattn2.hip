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

// Backward of the attention core: a block of four waves takes FOUR consecutive windows of one head, one after the other.
// Per window the block's 256 threads load the 64 q, k, v and dO rows of the head ONCE (8 floats per thread and tensor),
// split them under their row exponents and leave four LDS images [64 rows][32 entries] x {hi, lo} -- one image serves
// both operand forms:
//   row form     (the contraction runs over the head dim)  ds_read_b128 of a row's 16-byte chunk
//   column form  (the contraction runs over the rows)      ds_read_b64_tr_b16: the hardware's transposed read hands
//                lane (r, g) entry 16 jd + r of the rows w2_kpos(JJ, g, 0..7); the rows' exponents cannot be factored
//                out of such a sum, so their 2^-s go into the OTHER operand (P, dS) before that one is split
// so nothing is gathered from global memory inside the tile loops (the gathers were 2/3 of the one-wave-per-item kernel
// this replaces: 97 us, 300 MB of traffic for 178 MB of operands).
//   query side  wave I: S^T and dP^T = V . dO^T of query tile I against all keys -> softmax rows, delta = sum_k P dP,
//               dS^T: this lane's B operand of dQ^T = K^T . dS^T, and added to the block's bias-gradient tile in LDS;
//               log-sum-exp and delta of the rows go to LDS
//   key side    wave J: S and dP with the key on the lane (P re-made from the log-sum-exp) are the B operands of
//               dV^T = dO^T . P and dK^T = Q^T . dS
// Registers decide the shape (168 at three blocks per CU; a spilled address is reloaded through the same counter as the
// row loads and serialises them): the bias comes from the head's 225-entry table in LDS (index = lane constant +
// 30 (I - J) -+ e) and the bias-gradient sums live in LDS, which leaves room for the next window's rows to travel in
// registers while this one is computed; the cross-lane reductions are v_permlane16/32_swap (no LDS round trip).
// The block's dS sums leave as ONE partial tile (plain stores): k_dbias2_reduce sums the partials in fp64.
constexpr int W2_DS = 4096;                        // floats of a dS tile
constexpr int W3_PL = 64 * 64;                     // bytes of one plane of an image: 64 rows x 32 fp16
constexpr int W3_IMG = 2 * W3_PL;
constexpr int W3_WPB = 4;                          // windows per block (= the partial tiles' granularity)
constexpr int W3_LDS = 4 * W3_IMG + (W2_DS + 256 + 4 * 64 + 2 * 64) * 4;      // images, dS sums, bias table, row data

typedef short w3_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
template <int CTRL>
__device__ __forceinline__ float w3_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ int w3_slot(int row, int u) { return (row * 4 + (u ^ ((row >> 2) & 3))) * 16; }
// row-form fragment of tile T of an image: lane (c, g) = row 16 T + c, entries 8 g .. 8 g + 7
__device__ __forceinline__ void w3_rows(const unsigned char* img, int T, int c, int g, u32x4& hi, u32x4& lo) {
  const unsigned char* p = img + w3_slot(16 * T + c, g);
  hi = *(const u32x4*)p;
  lo = *(const u32x4*)(p + W3_PL);
}
// column-form fragment (JJ, jd): lane (r, g) = entry 16 jd + r of the rows w2_kpos(JJ, g, 0..7).  Lane 4 q + p of a
// 16-lane group supplies the address of row R0 + q, entries 16 jd + 4 p .. + 3 and receives entry 16 jd + (lane & 15)
// of rows R0 .. R0 + 3
__device__ __forceinline__ u32x4 w3_cols1(const unsigned char* plane, int JJ, int jd, int c, int g) {
  const int q = c >> 2, pp = c & 3;
  unsigned r[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 32 * JJ + 16 * h + 4 * g + q;
    const unsigned char* a = plane + w3_slot(row, 2 * jd + (pp >> 1)) + 8 * (pp & 1);
    const w3_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)a);
    const sr_u32x2 u = __builtin_bit_cast(sr_u32x2, v);
    r[2 * h] = u.x; r[2 * h + 1] = u.y;
  }
  return u32x4{r[0], r[1], r[2], r[3]};
}
__device__ __forceinline__ void w3_cols(const unsigned char* img, int JJ, int jd, int c, int g, u32x4& hi, u32x4& lo) {
  hi = w3_cols1(img, JJ, jd, c, g);
  lo = w3_cols1(img + W3_PL, JJ, jd, c, g);
}
#ifndef W3_OCC
#define W3_OCC 3
#endif
// phase timestamps of every wave (tools/mb_attn_phases.py): experiment builds only
#ifdef SRHIP_EXPERIMENTS
__device__ long long* g_w3_dbg = nullptr;
__device__ int g_w3_mode = 0;      // ablations (srhip_wattn2_debug_mode): 1 no arithmetic | 2 no stores | 4 no row loads
#define W3_MODE(B) (g_w3_mode & (B))
#define SR_TS(K) \
  if (g_w3_dbg && lane == 0) g_w3_dbg[(((long)blockIdx.x * 4 + wv) * W3_WPB + wi) * 8 + (K)] = (long long)wall_clock64();
#else
#define SR_TS(K)
#define W3_MODE(B) 0
#endif
template <int D>
__global__ void __launch_bounds__(256, W3_OCC) k_wattn3_bwd(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, const float* __restrict__ biasF,
                                                      float* __restrict__ part,
                                                      int nwin, int H, int W, int C, int heads, int shift, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char w3s[];
  unsigned char* const imQ = w3s;
  unsigned char* const imK = w3s + W3_IMG;
  unsigned char* const imV = w3s + 2 * W3_IMG;
  unsigned char* const imG = w3s + 3 * W3_IMG;
  float* const dsl = (float*)(w3s + 4 * W3_IMG);   // sum over the block's windows of the dS^T tiles (S^T accumulator order)
  float* const tab = dsl + W2_DS;                  // the head's relative-position bias table [15][15]
  float* const rs = tab + 256;                     // 2^-s of the rows: [0] Q (x scale), [64] K, [128] V, [192] dO
  float* const lse = rs + 256;
  float* const dlt = lse + 64;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = sr_xcd_block(blockIdx.x, gridDim.x);
  const int head = lb % heads, wg = lb / heads;
  const int nWx = W / 8, nWy = H / 8;
  const long C3 = 3L * C;
  const float* const qb = qkv + head * D;
  const float* const gb = dout + head * D;
  float* const dqb = dqkv + head * D;
  auto geom = [&](int widx) {
    W2Geom geo;
    geo.head = head;
    geo.wx = widx % nWx;
    geo.wy = (widx / nWx) % nWy;
    geo.b = widx / (nWx * nWy);
    geo.last_row = shift > 0 && geo.wy == nWy - 1;
    geo.last_col = shift > 0 && geo.wx == nWx - 1;
    return geo;
  };
  // the table entry (dy + 7) * 15 + dx + 7 read back from the S^T image of the head: any (query, key) pair at that offset
  if (tid < 225) {
    const int dy = tid / 15 - 7, dx = tid % 15 - 7;
    const int qy = max(dy, 0), ky = qy - dy, qx = max(dx, 0), kx = qx - dx;
    const int query = 8 * qy + qx, key = 8 * ky + kx;
    tab[tid] = ldg_f(biasF + (long)head * W2_DS + w2_img_index(query >> 4, key >> 4, 16 * ((key & 15) >> 2) + (query & 15)) + (key & 3));
  }
  // index of the bias of (query 16 I + c, key 16 J + 4 g + e) = bq + 30 (I - J) - e; of (query 16 I + 4 g + e, key 16 J + c)
  // = bk + 30 (I - J) + e
  const int bq = ((c >> 3) - (g >> 1) + 7) * 15 + (c & 7) - 4 * (g & 1) + 7;
  const int bk = ((g >> 1) - (c >> 3) + 7) * 15 + 4 * (g & 1) - (c & 7) + 7;
  // Loading role: wave wv takes rows 16 wv .. + 15 of every tensor, eight lanes per row, 16 bytes per lane (piece p =
  // floats 4 p .. 4 p + 3 of the head's D), two instructions per tensor -- an instruction covers eight whole 4 D-byte row
  // segments (8-16 cache lines; as 8-byte pieces, four lanes per row, it was 32 lines per instruction and four
  // instructions per tensor: the address pipe of the CU, not HBM, set the kernel's time).  A piece with two valid floats
  // (D = 30: p = 7) reads floats 4 p - 2 .. 4 p + 1 and keeps the upper half; an empty one reads piece 0: every load is
  // unconditional and inside the row.
  const int lp = lane & 7, lrow = 16 * wv + (lane >> 3);
  constexpr int kFull = D / 4;                     // pieces with four valid floats
  const int poff = lp < kFull ? 4 * lp : (4 * lp + 2 <= D ? 4 * lp - 2 : 0);
  f32x4 raw[4][2];
  auto fetch = [&](int widx) {
    const W2Geom geo = geom(widx);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tk = w2_token(geo, lrow + 8 * i, H, W, shift);
      if (W3_MODE(4)) {
#pragma unroll
        for (int X = 0; X < 4; ++X)
#pragma unroll
          for (int t = 0; t < 4; ++t) raw[X][i][t] = 0.01f * (float)((tk + 7 * t + 13 * X) & 63) - 0.3f;
        continue;
      }
      const float* q = qb + (long)tk * C3 + poff;
      raw[0][i] = *(w3_gp4)q;
      raw[1][i] = *(w3_gp4)(q + C);
      raw[2][i] = *(w3_gp4)(q + 2 * C);
      raw[3][i] = *(w3_gp4)(gb + (long)tk * C + poff);
    }
  };
  const int w0 = wg * W3_WPB;
  fetch(w0);

  for (int wi = 0; wi < W3_WPB; ++wi) {
    const int widx = w0 + wi;
    if (widx >= nwin) break;                       // block-uniform
    const W2Geom geo = geom(widx);
    SR_TS(0)
    // ---- the window's rows -> images (row maximum over the row's eight lanes: three DPP steps)
#pragma unroll
    for (int X = 0; X < 4; ++X)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = raw[X][i];
        if (lp >= kFull) v = (4 * lp + 2 <= D) ? f32x4{v[2], v[3], 0.f, 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
        float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        mx = fmaxf(mx, w3_dpp<0xB1>(mx));          // quad_perm [1, 0, 3, 2]
        mx = fmaxf(mx, w3_dpp<0x4E>(mx));          // quad_perm [2, 3, 0, 1]
        mx = fmaxf(mx, w3_dpp<0x141>(mx));         // row_half_mirror: lane i <-> 7 - i of its eight
        const float sc = pow2_scale(mx);
        unsigned h[2], l[2];
        split2_pair(v[0] * sc, v[1] * sc, h[0], l[0]);
        split2_pair(v[2] * sc, v[3] * sc, h[1], l[1]);
        const int row = lrow + 8 * i;
        unsigned char* dst = w3s + X * W3_IMG + w3_slot(row, lp >> 1) + 8 * (lp & 1);
        *(sr_u32x2*)dst = sr_u32x2{h[0], h[1]};
        *(sr_u32x2*)(dst + W3_PL) = sr_u32x2{l[0], l[1]};
        if (lp == 0) rs[64 * X + row] = pow2_inv(sc) * (X == 0 ? scale : 1.f);
      }
    // the next window's rows travel while this one is computed (32 registers: the kernel stays under the 168 of three
    // blocks per CU without a spill -- a spilled address is reloaded through the same counter as these loads)
    if (widx + 1 < nwin && wi + 1 < W3_WPB) fetch(widx + 1);
    SR_TS(1)
    sr_lds_barrier();
    SR_TS(2)
    const bool lane_masked = geo.last_col && (((c >> 2) & 1) != (g & 1));

    // ================= query side: query tile I = wv
    if (!W3_MODE(1)) {
      const int I = wv;
      u32x4 qh, ql, gh, gl;
      w3_rows(imQ, I, c, g, qh, ql);
      w3_rows(imG, I, c, g, gh, gl);
      const float rq = rs[16 * I + c], rg = rs[192 + 16 * I + c];
      f32x4 S[4], P[4];
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        u32x4 kh, kl, vh, vl;
        w3_rows(imK, J, c, g, kh, kl);
        w3_rows(imV, J, c, g, vh, vl);
        S[J] = mfma3(kh, kl, qh, ql, f32x4{0.f, 0.f, 0.f, 0.f});
        P[J] = mfma3(vh, vl, gh, gl, f32x4{0.f, 0.f, 0.f, 0.f});      // dP^T
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        const float* tb = tab + bq + 30 * (I - J);
        const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
        const f32x4 rkk = *(const f32x4*)(rs + 64 + 16 * J + 4 * g);
        const f32x4 rvv = *(const f32x4*)(rs + 128 + 16 * J + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sv = S[J][e] * (rq * rkk[e]) + tb[-e];
          sv += masked ? -100.f : 0.f;
          S[J][e] = sv;
          mx = fmaxf(mx, sv);
          P[J][e] *= rg * rvv[e];
        }
      }
      mx = w3_max4(mx);
      float sum = 0.f;
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __expf(S[J][e] - mx);
          S[J][e] = pe;
          sum += pe;
        }
      sum = w3_sum4(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
      float dl = 0.f;
#pragma unroll
      for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S[J][e] *= inv;
          dl += S[J][e] * P[J][e];
        }
      dl = w3_sum4(dl);
      if (g == 0) { lse[16 * I + c] = mx + __logf(sum); dlt[16 * I + c] = dl; }
      float dmx = 0.f;
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        const f32x4 rkk = *(const f32x4*)(rs + 64 + 16 * J + 4 * g);
        f32x4* const acc = (f32x4*)(dsl + w2_img_index(I, J, lane));
        f32x4 ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ds[e] = S[J][e] * (P[J][e] - dl);
          const float dk = ds[e] * rkk[e];           // the key row's 2^-s rides in dS: K^T below is read as stored
          S[J][e] = dk;
          dmx = fmaxf(dmx, fabsf(dk));
        }
        if (wi == 0) *acc = ds;                      // (the wave's own tiles: no other wave touches them)
        else { const f32x4 o = *acc; *acc = f32x4{o[0] + ds[0], o[1] + ds[1], o[2] + ds[2], o[3] + ds[3]}; }
      }
      dmx = w3_max4(dmx);
      const float dsc = pow2_scale(dmx), dri = pow2_inv(dsc) * scale;
      f32x4 O[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int JJ = 0; JJ < 2; ++JJ) {
        u32x4 dh, dlo;
        w2_pack_pair(S[2 * JJ], S[2 * JJ + 1], dsc, dh, dlo);
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          u32x4 kth, ktl;
          w3_cols(imK, JJ, jd, c, g, kth, ktl);
          O[jd] = mfma3(kth, ktl, dh, dlo, O[jd]);
        }
      }
      float* op = dqb + (long)w2_token(geo, 16 * I + c, H, W, shift) * C3 + 4 * g;
      if (!W3_MODE(2))
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) w3_store<D>(op + 16 * jd, 16 * jd + 4 * g, O[jd], dri);
    }
    SR_TS(3)
    sr_lds_barrier();                                // log-sum-exp and delta of every row
    SR_TS(4)

    // ================= key side: key tile J = wv
    if (!W3_MODE(1)) {
      const int J = wv;
      u32x4 kh, kl, vh, vl;
      w3_rows(imK, J, c, g, kh, kl);
      w3_rows(imV, J, c, g, vh, vl);
      const float rk = rs[64 + 16 * J + c], rv = rs[128 + 16 * J + c];
      f32x4 S[4], P[4];      // lane (c, g): key 16 J + c against queries 16 I + 4 g + e
      float dmx = 0.f, pmx = 0.f;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        u32x4 qh, ql, gh, gl;
        w3_rows(imQ, I, c, g, qh, ql);
        w3_rows(imG, I, c, g, gh, gl);
        S[I] = mfma3(qh, ql, kh, kl, f32x4{0.f, 0.f, 0.f, 0.f});
        P[I] = mfma3(gh, gl, vh, vl, f32x4{0.f, 0.f, 0.f, 0.f});
        const float* tb = tab + bk + 30 * (I - J);
        const f32x4 ls = *(const f32x4*)(lse + 16 * I + 4 * g), dl = *(const f32x4*)(dlt + 16 * I + 4 * g);
        const f32x4 rq = *(const f32x4*)(rs + 16 * I + 4 * g), rg = *(const f32x4*)(rs + 192 + 16 * I + 4 * g);
        const bool masked = lane_masked || (geo.last_row && ((I >> 1) != (J >> 1)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sv = S[I][e] * (rq[e] * rk) + tb[e];
          sv += masked ? -100.f : 0.f;
          const float pe = __expf(sv - ls[e]);
          const float dsv = pe * (P[I][e] * (rg[e] * rv) - dl[e]);
          // the query rows' 2^-s (dO's for P, q's -- times the logit scale -- for dS) ride in these operands
          const float pf = pe * rg[e], df = dsv * rq[e];
          P[I][e] = pf;
          S[I][e] = df;
          pmx = fmaxf(pmx, pf);
          dmx = fmaxf(dmx, fabsf(df));
        }
      }
      dmx = w3_max4(dmx);
      pmx = w3_max4(pmx);
      const float dsc = pow2_scale(dmx), dri = pow2_inv(dsc);
      const float psc = pow2_scale(pmx), pri = pow2_inv(psc);
      f32x4 OV[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 OK[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int II = 0; II < 2; ++II) {
        u32x4 ph, pl, dh, dlo;
        w2_pack_pair(P[2 * II], P[2 * II + 1], psc, ph, pl);
        w2_pack_pair(S[2 * II], S[2 * II + 1], dsc, dh, dlo);
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
          u32x4 th, tl;
          w3_cols(imG, II, jd, c, g, th, tl);
          OV[jd] = mfma3(th, tl, ph, pl, OV[jd]);
          w3_cols(imQ, II, jd, c, g, th, tl);
          OK[jd] = mfma3(th, tl, dh, dlo, OK[jd]);
        }
      }
      float* op = dqb + (long)w2_token(geo, 16 * J + c, H, W, shift) * C3 + 4 * g;
      if (!W3_MODE(2))
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) {
        w3_store<D>(op + C + 16 * jd, 16 * jd + 4 * g, OK[jd], dri);
        w3_store<D>(op + 2 * C + 16 * jd, 16 * jd + 4 * g, OV[jd], pri);
      }
    }
    SR_TS(5)
    sr_lds_barrier();                                // the images are free for the next window
    SR_TS(6)
  }
  if (part) {       // the block's partial bias-gradient tile: wave I owns the tiles (I, 0..3)
    float* dst = part + ((long)wg * heads + head) * W2_DS;
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const int o = w2_img_index(wv, J, lane);
      *(f32x4*)(dst + o) = *(const f32x4*)(dsl + o);
    }
  }
}

// partial tiles [nparts][heads][4096] (accumulator order of k_wattn2_bwd) -> the bias-gradient image in the order
// srhip_bias_grad reads (wa_dimg_index, wattn.hip: [a][b][q][lane] of the 32x32 tiles, key = mfma_row(q, lane) + 32 a,
// query = (lane & 31) + 32 b); fp64 sums in partial order
// (blockIdx.y = attention block of a batched call: its partials at part + y * part_stride, its image at dimg + y * img_stride)
__global__ void __launch_bounds__(256) k_dbias2_reduce(const float* __restrict__ part, int nparts, int heads,
                                                       float* __restrict__ dimg, long part_stride, long img_stride) {
  // block = (head, a, b, q >> 2): 64 lanes x the 4 registers q & 3, which are 4 CONSECUTIVE floats of a partial tile
  // (key & 3 == q & 3): one 16-byte load per partial; the partials dealt to the block's 4 waves, joined through LDS
  __shared__ double sm[4][64][4];
  const int ln = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int hd = blockIdx.x >> 4, a = (blockIdx.x >> 3) & 1, b = (blockIdx.x >> 2) & 1, qh = blockIdx.x & 3;
  part += (long)blockIdx.y * part_stride;
  dimg += (long)blockIdx.y * img_stride;
  const int key = mfma_row(4 * qh, ln) + 32 * a, query = (ln & 31) + 32 * b;
  const int mine = w2_img_index(query >> 4, key >> 4, 16 * ((key >> 2) & 3) + (query & 15));
  const float* p = part + (long)hd * W2_DS + mine;
  const long ps = (long)heads * W2_DS;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = pg;
  for (; i + 28 < nparts; i += 32) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(p + (long)(i + 4 * u) * ps);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += (double)v[u][0]; s1 += (double)v[u][1]; s2 += (double)v[u][2]; s3 += (double)v[u][3]; }
  }
  for (; i < nparts; i += 4) {
    const f32x4 v = *(const f32x4*)(p + (long)i * ps);
    s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
  }
  sm[pg][ln][0] = s0; sm[pg][ln][1] = s1; sm[pg][ln][2] = s2; sm[pg][ln][3] = s3;
  __syncthreads();
  // wave pg writes register q = 4 qh + pg of the 64 lanes: one 256-byte row of the image
  const double t = (sm[0][ln][pg] + sm[1][ln][pg]) + (sm[2][ln][pg] + sm[3][ln][pg]);
  dimg[(long)hd * 4096 + ((a * 2 + b) * 16 + 4 * qh + pg) * 64 + ln] = (float)t;
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
// [blocks][4 waves][4 windows][8] stamps of the 100 MHz wall clock
__attribute__((visibility("default"))) int srhip_wattn2_debug_buffer(long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w3_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
__attribute__((visibility("default"))) int srhip_wattn2_debug_mode(int mode) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w3_mode), &mode, sizeof(mode)) == hipSuccess ? 0 : -5;
}
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
  dim3 grid(nparts * heads), blk(256);
  float* part = workspace;
#define SR_WA(D_)                                                                                                       \
  if (D == D_)                                                                                                          \
    hipLaunchKernelGGL((k_wattn3_bwd<D_>), grid, blk, W3_LDS, st, qkv, dout, dqkv, biasF, part, nwin, H, W, C,        \
                       heads, shift, scale);
  SR_WA(30) SR_WA(10) SR_WA(16) SR_WA(32)
#undef SR_WA
  if (dbiasT)
    hipLaunchKernelGGL(k_dbias2_reduce, dim3(heads * 16), dim3(256), 0, st, part, nparts, heads, dbiasT, 0L, 0L);
  SR_LAUNCH_CHECK("window_attention_bwd_f16x2");
  return 0;
}

// The bias-gradient images of nblocks attention blocks (same B, H, W, heads) from the partial tiles their backward
// launches left at workspace + i * ws_stride, in one launch: dbiasT + i * img_stride.
int srhip_window_attention_dbias_reduce_f16x2(const float* workspace, long ws_stride, int nblocks, float* dbiasT,
                                              long img_stride, int B, int H, int W, int heads, void* stream) {
  SR_REQUIRE(workspace && dbiasT && nblocks > 0 && nblocks <= 65535, "window_attention_dbias_reduce_f16x2: bad arguments");
  const int nparts = sr_cdiv(B * (H / 8) * (W / 8), 4);
  hipLaunchKernelGGL(k_dbias2_reduce, dim3(heads * 16, nblocks), dim3(256), 0, (hipStream_t)stream,
                     workspace, nparts, heads, dbiasT, ws_stride, img_stride);
  SR_LAUNCH_CHECK("window_attention_dbias_reduce_f16x2");
  return 0;
}

}  // extern "C"
