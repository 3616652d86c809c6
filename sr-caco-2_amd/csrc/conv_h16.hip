// fp16 STORAGE inference path of the plain conv family (VDSR, DRRN, EDSR; --amp at evaluation time): activations live in
// HBM as fp16 (channels-last), weights are the LEADING fp16 plane of the fp16x2 conv operand (srhip_prep_table job kind 4:
// [Kp/16][9*Cout][16] fp16 under one power-of-two scale per output channel, followed by the second plane and the Cout
// inverse scales), ONE v_mfma_f32_16x16x32_f16 product, f32 accumulate, fp16 out.  Half the bytes of the f32-storage --amp
// path on both sides of every conv and no split arithmetic in the staging: the halo tile goes from global memory into the
// stage image as it is.
//
//   k_conv3x3_h16   3x3 conv, stride 1, zero padding, Cin a multiple of 32, Cout a multiple of 64: 2 RW x 16 pixel tiles x
//                   64-column slices (the shapes and wave layout of k_nhcw2, gemm_ntw.hip); the nine taps of a 32-channel
//                   chunk run without a barrier on weight fragments requested a chunk ahead; optional BatchNorm-ReLU input
//                   prologue (MemNet); epilogues bias / ReLU / LeakyReLU / residual / residual + ReLU on the row-major
//                   re-laid tile (16-byte accesses), optionally stored through PixelShuffle(2) (the EDSR upsampler,
//                   network_nlsn.py:100-118)
//   k_conv1x1_h16   a 1x1 conv held as the centre tap of a 3x3 operand (MemNet's gate units, SRCNN's layers): no halo,
//                   three 32-channel chunks per stage
//   k_cin1_h16      the 1 -> Cout conv at the head (f32 image in, fp16 features out; network_vdsr.py:57-60)
//   k_cout1_h16     the Cin -> 1 conv at the tail (fp16 features in, f32 image out, + the interpolated input)
#include "common.h"
#include "kernels.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HP = 80;                           // bytes per halo pixel of a 32-channel chunk (64 + 16 pad: conflict-free ds_read_b128)
constexpr int TPF = 68;                          // pitch of the f32 output tile (floats)

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

constexpr int h16_lds(int rw) {       // two halo tiles (chunks alternate) | three chunks of the tile's own pixels (1x1) | the f32 output tile
  const int halo = 2 * (2 * rw + 2) * 18 * HP, one = 3 * 2 * rw * 16 * HP, tile = 2 * rw * 16 * TPF * 4;
  return (halo > tile ? halo : tile) > one ? (halo > tile ? halo : tile) : one;
}

// The accumulators of a 2 RW x 16 pixel x 64 column tile (acc[i][j][e] = pixel 16 (RW wm + i) + 4 g + e, column 32 wn + 16 j + c)
// -> row-major in LDS -> bias / activation / residual on 8-column groups -> fp16 stores of 16 bytes.
template <int RW, bool LDS_ONLY = false>
__device__ __forceinline__ void h16_epilogue(const ConvH16Args& p, const f32x4 (&acc)[RW][2], unsigned char* smem, int tid, int wm,
                                             int wn, int c, int g, int n0, int img, int y0, int x0) {
  constexpr int NPX = 2 * RW * 16;
  // (LDS_ONLY: the persistent kernel has the next tile's halo loads in flight here; __syncthreads() would wait for them)
  if (LDS_ONLY) sr_lds_barrier(); else __syncthreads();     // the halo tile is dead from here on
  float* const T = (float*)smem;
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) T[(16 * (RW * wm + i) + 4 * g + e) * TPF + 32 * wn + 16 * j + c] = acc[i][j][e];
  if (LDS_ONLY) sr_lds_barrier(); else __syncthreads();
  const float* const winv = (const float*)((const char*)p.Wb + 2 * p.plane_bytes);
  // thread -> (pixel, 8-column group): NPX * 8 items
#pragma unroll
  for (int it = 0; it < (NPX * 8) / 256; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 3, c8 = idx & 7;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    if (y >= p.H || x >= p.Wd) continue;
    const int col = n0 + c8 * 8;
    const f32x4 t0 = *(const f32x4*)(T + px * TPF + c8 * 8), t1 = *(const f32x4*)(T + px * TPF + c8 * 8 + 4);
    const f32x4 w0 = ldg_f4(winv + col), w1 = ldg_f4(winv + col + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = t0[e] * w0[e]; v[4 + e] = t1[e] * w1[e]; }
    if (p.bias) {
      // conv + PixelShuffle(2): kernel column sp * (N / 4) + cc holds torch channel cc * 4 + sp
      if (p.ps) {
        const int fs = p.N >> 2, sp = col / fs, cc = col - sp * fs;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[(cc + e) * 4 + sp];
      } else {
        const f32x4 b0 = ldg_f4(p.bias + col), b1 = ldg_f4(p.bias + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
    }
    const long pix = ((long)img * p.H + y) * p.Wd + x;
    if (p.epi == 2 || p.epi == 8) {
      const h16x8 r = *(const h16x8*)(p.R + pix * p.ldr + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + (float)r[e];
    }
    if (p.epi == 1 || p.epi == 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.epi == 6) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
    if (p.ps) {
      const int fs = p.N >> 2, sp = col / fs, cc = col - sp * fs;
      const long opix = ((long)img * 2 * p.H + 2 * y + (sp >> 1)) * (2 * p.Wd) + 2 * x + (sp & 1);
      *(h16x8*)(p.Y + opix * p.ldy + cc) = o;
    } else {
      *(h16x8*)(p.Y + pix * p.ldy + col) = o;
    }
  }
}

template <int RW>
__global__ void __launch_bounds__(256, 3) k_conv3x3_h16(ConvH16Args p) {
  constexpr int AROWS = (2 * RW + 2) * 18;       // halo pixels of a 2 RW x 16 tile
  constexpr int AN = AROWS * 4;                  // 16-byte slots per chunk (8 channels each)
  constexpr int AIT = (AN + 255) / 256;
  constexpr int NPX = 2 * RW * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int t = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int ncol = p.N >> 6;
  const int n0 = (t % ncol) * 64; t /= ncol;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = p.K >> 5;

  unsigned offA[AIT];
  bool inA[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int idx = min(tid + it * 256, AN - 1);
    const int row = idx >> 2, c8 = idx & 3;
    const int hy = row / 18, hx = row - hy * 18;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    inA[it] = y >= 0 && y < p.H && x >= 0 && x < p.Wd && (AN % 256 == 0 || tid + it * 256 < AN);
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
    offA[it] = (unsigned)((((long)img * p.H + yc) * p.Wd + xc) * p.ldx + c8 * 8) * 2u;
  }
  // input prologue (MemNet's BN-ReLU-conv, network_memnet.py:27-34): relu((x - mean) k + beta) on the thread's eight
  // channels of the chunk (256 % 4 == 0: the same group in every iteration), before the padding zeros go in
  f32x4 bm[2], bk[2], bb[2];
  auto load_a = [&](int kc, u32x4 (&ra)[AIT]) {
    const char* base = (const char*)p.X + (long)kc * 64;
#pragma unroll
    for (int it = 0; it < AIT; ++it) ra[it] = inA[it] ? *(const u32x4*)(base + offA[it]) : u32x4{0u, 0u, 0u, 0u};
  };
  constexpr int IMG = AROWS * HP;                 // one chunk's halo image; two of them, used alternately
  auto store_a = [&](const u32x4 (&ra)[AIT], int kc) {
    unsigned char* const img_w = smem + (kc & 1) * IMG;
    if (p.in_bn) {                                // (L1 / L2 hits: every block reads the same 3 x 32 floats per chunk)
      const int ch = kc * 32 + (tid & 3) * 8;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        bm[h2] = ldg_f4(p.in_bn + ch + 4 * h2);
        bk[h2] = ldg_f4(p.in_bn + 2 * p.K + ch + 4 * h2);
        bb[h2] = ldg_f4(p.in_bn + 3 * p.K + ch + 4 * h2);
      }
    }
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      if (AN % 256 == 0 || tid + it * 256 < AN) {
        const int idx = tid + it * 256;
        u32x4 v = ra[it];
        if (p.in_bn) {
          h16x8 hv = __builtin_bit_cast(h16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = (_Float16)fmaxf(((float)hv[e] - bm[e >> 2][e & 3]) * bk[e >> 2][e & 3] + bb[e >> 2][e & 3], 0.f);
          v = __builtin_bit_cast(u32x4, hv);
        }
        *(u32x4*)(img_w + (idx >> 2) * HP + (idx & 3) * 16) = inA[it] ? v : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  const long wrows = 9L * p.N;
  unsigned boff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) boff[jt] = (unsigned)(((g >> 1) * wrows + n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
  auto load_b = [&](int kc, int tap, u32x4 (&fb)[2]) {
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) fb[jt] = *(const u32x4*)(base + boff[jt]);
  };
  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 18 + c) * HP + 16 * g;

  u32x4 ra[AIT];
  load_a(0, ra);
  // weight fragments three taps ahead, in three register sets (a whole chunk ahead took 72 registers: two blocks per CU;
  // with 24 the kernel fits three)
  u32x4 fb0[2], fb1[2], fb2[2];
  const int niter = nkc * 9;
  // every block walks the nine taps from another start (weight lines spread over the L2: gemm_ntw.hip, round 5)
  const int rot9 = (int)(((unsigned)blockIdx.x >> 3) % 9u);
  auto tr9 = [&](int tap) { const int x = tap + rot9; return x >= 9 ? x - 9 : x; };
  auto load_it = [&](int it, u32x4 (&fb)[2]) { load_b(it / 9, tr9(it % 9), fb); };
  auto mma = [&](const unsigned char* img_r, int tap, const u32x4 (&fb)[2]) {
    const int tp = tr9(tap), ty3 = (tp * 11) >> 5;          // tp / 3 for tp < 9
    const int toff = (ty3 * 18 + (tp - 3 * ty3)) * HP;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const u32x4 fa = *(const u32x4*)(img_r + a_off[i] + toff);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa, fb[j], acc[i][j]);
    }
  };
  load_it(0, fb0); load_it(1, fb1); load_it(2, fb2);
  for (int kc = 0; kc < nkc; ++kc) {
    // ONE barrier per chunk: chunk kc goes into image kc & 1 while slower waves may still read the other one (whoever
    // writes image kc & 1 has passed barrier kc - 1, which every wave reaches behind its taps of chunk kc - 2)
    store_a(ra, kc);
    __syncthreads();
    if (kc + 1 < nkc) load_a(kc + 1, ra);
    const unsigned char* const img_r = smem + (kc & 1) * IMG;
    const int it = kc * 9;
#pragma unroll
    for (int t3 = 0; t3 < 9; t3 += 3) {
      mma(img_r, t3, fb0);     if (it + t3 + 3 < niter) load_it(it + t3 + 3, fb0);
      mma(img_r, t3 + 1, fb1); if (it + t3 + 4 < niter) load_it(it + t3 + 4, fb1);
      mma(img_r, t3 + 2, fb2); if (it + t3 + 5 < niter) load_it(it + t3 + 5, fb2);
    }
  }

  h16_epilogue<RW>(p, acc, smem, tid, wm, wn, c, g, n0, img, y0, x0);
}

// The same conv as a PERSISTENT kernel for the launches that are many rounds of blocks (round 5).  A
// 64 -> 64 layer on 512 x 512 x 8 pixels is 16,384 tiles of 144 MFMAs per wave -- 1.1 us of matrix work each behind three
// dependent memory round trips (the first halo chunk, the weight fragments, the residual): with three blocks per CU the
// launch ran at a block's latency, 215 us against ~65 us of either floor.  Here a block walks tiles (grid = 3 blocks per CU,
// tiles b, b + G, ...: the same XCD-aware order, G a multiple of 8) and, when the last tap of a tile has been issued,
// the stream of (tile, chunk) stages simply runs on: behind the last chunk's barrier the NEXT tile's first halo chunk is
// requested, behind its last taps the next tile's first weight fragments; they travel during the epilogue (LDS-only
// barriers: nothing waits for them) and the next tile starts from registers.  Same arithmetic, same order per output.
template <bool BN>
__global__ void __launch_bounds__(256, 3) k_conv3x3_h16p(ConvH16Args p, int ntiles) {
  constexpr int RW = 4;
  constexpr int AROWS = (2 * RW + 2) * 18;
  constexpr int AN = AROWS * 4;
  constexpr int AIT = (AN + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int ncol = p.N >> 6;
  const int nkc = p.K >> 5;
  const long wrows = 9L * p.N;
  constexpr int IMG = AROWS * HP;
  struct Geo { int n0, img, y0, x0; unsigned offA[AIT]; unsigned in; };
  auto geom = [&](int tl, Geo& q) {
    int t = sr_xcd_block(tl, ntiles);
    q.n0 = (t % ncol) * 64; t /= ncol;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    q.img = t / p.tiles_y;
    q.y0 = ty * (2 * RW); q.x0 = tx * 16;
    q.in = 0u;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int idx = min(tid + it * 256, AN - 1);
      const int row = idx >> 2, c8 = idx & 3;
      const int hy = row / 18, hx = row - hy * 18;
      const int y = q.y0 + hy - 1, x = q.x0 + hx - 1;
      if (y >= 0 && y < p.H && x >= 0 && x < p.Wd && (AN % 256 == 0 || tid + it * 256 < AN)) q.in |= 1u << it;
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.Wd - 1);
      q.offA[it] = (unsigned)((((long)q.img * p.H + yc) * p.Wd + xc) * p.ldx + c8 * 8) * 2u;
    }
  };
  auto load_a = [&](const Geo& q, int kc, u32x4 (&ra)[AIT]) {
    const char* base = (const char*)p.X + (long)kc * 64;
#pragma unroll
    for (int it = 0; it < AIT; ++it) ra[it] = ((q.in >> it) & 1u) ? *(const u32x4*)(base + q.offA[it]) : u32x4{0u, 0u, 0u, 0u};
  };
  f32x4 bm[2], bk[2], bb[2];
  auto store_a = [&](const u32x4 (&ra)[AIT], int kc, unsigned in) {
    unsigned char* const img_w = smem + (kc & 1) * IMG;
    if (BN) {
      const int ch = kc * 32 + (tid & 3) * 8;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        bm[h2] = ldg_f4(p.in_bn + ch + 4 * h2);
        bk[h2] = ldg_f4(p.in_bn + 2 * p.K + ch + 4 * h2);
        bb[h2] = ldg_f4(p.in_bn + 3 * p.K + ch + 4 * h2);
      }
    }
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      if (AN % 256 == 0 || tid + it * 256 < AN) {
        const int idx = tid + it * 256;
        u32x4 v = ra[it];
        if (BN) {
          h16x8 hv = __builtin_bit_cast(h16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = (_Float16)fmaxf(((float)hv[e] - bm[e >> 2][e & 3]) * bk[e >> 2][e & 3] + bb[e >> 2][e & 3], 0.f);
          v = __builtin_bit_cast(u32x4, hv);
        }
        *(u32x4*)(img_w + (idx >> 2) * HP + (idx & 3) * 16) = ((in >> it) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  unsigned boff[2];
  auto set_boff = [&](int n0) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) boff[jt] = (unsigned)(((g >> 1) * wrows + n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
  };
  auto load_b = [&](int kc, int tap, u32x4 (&fb)[2]) {
    const char* base = (const char*)p.Wb + ((long)(2 * kc) * wrows + (long)tap * p.N) * 32;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) fb[jt] = *(const u32x4*)(base + boff[jt]);
  };
  f32x4 acc[RW][2];
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 18 + c) * HP + 16 * g;
  auto mma = [&](const unsigned char* img_r, int tap, const u32x4 (&fb)[2]) {
    const int toff = ((tap / 3) * 18 + (tap % 3)) * HP;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const u32x4 fa = *(const u32x4*)(img_r + a_off[i] + toff);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa, fb[j], acc[i][j]);
    }
  };
  const int niter = nkc * 9;
  auto load_it = [&](int it, u32x4 (&fb)[2]) { load_b(it / 9, it % 9, fb); };

  Geo gc, gn;
  int tl = blockIdx.x;
  geom(tl, gc);
  gn = gc;
  set_boff(gc.n0);
  unsigned boffn[2] = {boff[0], boff[1]};            // the NEXT tile's weight-fragment offsets (its column slice may differ)
  u32x4 ra[AIT];
  u32x4 fb0[2], fb1[2], fb2[2];
  load_it(0, fb0); load_it(1, fb1); load_it(2, fb2);
  load_a(gc, 0, ra);
  for (;;) {
    const int tn = tl + (int)gridDim.x;
    const bool more = tn < ntiles;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nkc; ++kc) {
      store_a(ra, kc, gc.in);
      sr_lds_barrier();                            // ONE barrier per chunk (as k_conv3x3_h16), LDS only: prefetches stay in flight
      const bool last = kc + 1 == nkc;
      if (!last) load_a(gc, kc + 1, ra);
      else if (more) {                             // the stream of chunks runs on into the next tile: its first halo chunk now,
        geom(tn, gn);                              // its first weight fragments behind this chunk's last taps
        load_a(gn, 0, ra);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          boffn[jt] = (unsigned)(((g >> 1) * wrows + gn.n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
      }
      const unsigned char* const img_r = smem + (kc & 1) * IMG;
      const int it = kc * 9;
      // fragments of tap it + 3 ..: this tile's while there are any, then taps 0, 1, 2 of the next tile
      auto next_frag = [&](int nx, u32x4 (&fb)[2]) {
        if (nx < niter) load_it(nx, fb);
        else if (more) {
          const char* base = (const char*)p.Wb + (long)(nx - niter) * p.N * 32;
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) fb[jt] = *(const u32x4*)(base + boffn[jt]);
        }
      };
#pragma unroll
      for (int t3 = 0; t3 < 9; t3 += 3) {
        mma(img_r, t3, fb0);     next_frag(it + t3 + 3, fb0);
        mma(img_r, t3 + 1, fb1); next_frag(it + t3 + 4, fb1);
        mma(img_r, t3 + 2, fb2); next_frag(it + t3 + 5, fb2);
      }
    }
    h16_epilogue<RW, true>(p, acc, smem, tid, wm, wn, c, g, gc.n0, gc.img, gc.y0, gc.x0);
    if (!more) break;
    gc = gn;
    boff[0] = boffn[0]; boff[1] = boffn[1];
    tl = tn;
    sr_lds_barrier();                              // every thread has read its part of the output tile: the images may be written
  }
}

// 1x1 conv (the weight's centre tap): no halo, so a stage is THREE 32-channel chunks of the tile's own pixels (one pair of
// barriers per 96 channels instead of per 32: at one tap a chunk is 8 MFMAs per wave, the barriers were the loop).
template <int RW>
__global__ void __launch_bounds__(256, 2) k_conv1x1_h16(ConvH16Args p) {
  constexpr int NPX = 2 * RW * 16;
  constexpr int CPS = 3;                         // chunks per stage
  constexpr int AN = NPX * 4;                    // 16-byte slots per chunk
  constexpr int AIT = AN / 256;                  // 2 (RW = 4) or 1 (RW = 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  int t = sr_xcd_block((int)blockIdx.x, gridDim.x);
  const int ncol = p.N >> 6;
  const int n0 = (t % ncol) * 64; t /= ncol;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int y0 = ty * (2 * RW), x0 = tx * 16;
  const int nkc = p.K >> 5;
  unsigned offA[AIT];
  bool inA[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int idx = tid + it * 256;
    const int px = idx >> 2, c8 = idx & 3;
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    inA[it] = y < p.H && x < p.Wd;
    offA[it] = (unsigned)((((long)img * p.H + min(y, p.H - 1)) * p.Wd + min(x, p.Wd - 1)) * p.ldx + c8 * 8) * 2u;
  }
  f32x4 bm[CPS][2], bk[CPS][2], bb[CPS][2];
  auto load_a = [&](int kc0, u32x4 (&ra)[CPS][AIT]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      const int kc = min(kc0 + q, nkc - 1);
      const char* base = (const char*)p.X + (long)kc * 64;
#pragma unroll
      for (int it = 0; it < AIT; ++it) ra[q][it] = inA[it] ? *(const u32x4*)(base + offA[it]) : u32x4{0u, 0u, 0u, 0u};
      if (p.in_bn) {
        const int ch = kc * 32 + (tid & 3) * 8;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          bm[q][h2] = ldg_f4(p.in_bn + ch + 4 * h2);
          bk[q][h2] = ldg_f4(p.in_bn + 2 * p.K + ch + 4 * h2);
          bb[q][h2] = ldg_f4(p.in_bn + 3 * p.K + ch + 4 * h2);
        }
      }
    }
  };
  auto store_a = [&](const u32x4 (&ra)[CPS][AIT]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q)
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int idx = tid + it * 256;
        u32x4 v = ra[q][it];
        if (p.in_bn) {
          h16x8 hv = __builtin_bit_cast(h16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            hv[e] = (_Float16)fmaxf(((float)hv[e] - bm[q][e >> 2][e & 3]) * bk[q][e >> 2][e & 3] + bb[q][e >> 2][e & 3], 0.f);
          v = inA[it] ? __builtin_bit_cast(u32x4, hv) : u32x4{0u, 0u, 0u, 0u};
        }
        *(u32x4*)(smem + q * NPX * HP + (idx >> 2) * HP + (idx & 3) * 16) = v;
      }
  };
  const long wrows = 9L * p.N;
  unsigned boff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) boff[jt] = (unsigned)(((g >> 1) * wrows + 4L * p.N + n0 + wn * 32 + jt * 16 + c) * 32 + (g & 1) * 16);
  auto load_b = [&](int kc0, u32x4 (&fb)[CPS][2]) {
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      const char* base = (const char*)p.Wb + (long)(2 * min(kc0 + q, nkc - 1)) * wrows * 32;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) fb[q][jt] = *(const u32x4*)(base + boff[jt]);
    }
  };
  f32x4 acc[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int a_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) a_off[i] = ((RW * wm + i) * 16 + c) * HP + 16 * g;
  u32x4 ra[CPS][AIT];
  u32x4 fb[CPS][2];
  load_a(0, ra);
  load_b(0, fb);
  for (int kc0 = 0; kc0 < nkc; kc0 += CPS) {
    if (kc0) __syncthreads();
    store_a(ra);
    __syncthreads();
    if (kc0 + CPS < nkc) load_a(kc0 + CPS, ra);
#pragma unroll
    for (int q = 0; q < CPS; ++q) {
      if (kc0 + q < nkc) {                          // block-uniform
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const u32x4 fa = *(const u32x4*)(smem + q * NPX * HP + a_off[i]);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(fa, fb[q][j], acc[i][j]);
        }
      }
    }
    if (kc0 + CPS < nkc) load_b(kc0 + CPS, fb);
  }
  h16_epilogue<RW>(p, acc, smem, tid, wm, wn, c, g, n0, img, y0, x0);
}

// SRCNN under --amp, all three layers in ONE kernel (network_srcnn.py:23-69: features 5x5 1 -> 1024 + ReLU as its 32-wide patch
// matrix, map 1x1 1024 -> 128 + ReLU, reconstruction 1x1 128 -> 1): a block owns 128 pixels and walks the 1024 hidden channels
// in 16 chunks of 64 -- layer 1 of a chunk TRANSPOSED (weights on the MFMA's row side, so that a lane ends with four consecutive
// hidden units of one pixel: half an 8-k unit of layer 2's A operand, one 8-byte LDS write), bias + ReLU + fp16 in registers,
// layer 2 accumulating over the chunks from a double-buffered stage image; the 128-wide result meets its 128 -> 1 dot product
// in the epilogue.  The 1024-channel map (8.6 GB at B = 8, 512 x 512 in f32, written and read back by the layer-wise path)
// never exists.  Weights: the centre taps of the fp16x2 conv operands srhip_prep_table builds (leading plane).
struct SrcnnH16Args {
  const _Float16* A0;                // [T][32] patch matrix (25 taps + 7 zeros), or null: built here from the image
  const float* img; int B, H, W;     // the f32 image [B][H][W] (A0 == null)
  const unsigned short* W1; long plane1; const float* b1;     // planes of [1024][32] at tap 4 of a [9*1024] row set
  const unsigned short* W2; long plane2; const float* b2;     // planes of [128][1024] at tap 4 of a [9*128] row set
  const float* w3; const float* b3;  // [128], [1]
  float* y;                          // [T]
  long T;
};
constexpr int SC_N1 = 1024, SC_N2 = 128, SC_NPX = 128;
constexpr int SC_LDS = SC_NPX * HP + 2 * 2 * SC_NPX * HP + 2 * SC_NPX * 4;      // A0 image + two h1 images (2 sub-chunks each) + row sums

__global__ void __launch_bounds__(256, 3) k_srcnn_h16(SrcnnH16Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const a0img = smem;
  unsigned char* const h1img = smem + SC_NPX * HP;                 // [buf][sub][px][HP]
  float* const rsum = (float*)(h1img + 4 * SC_NPX * HP);           // [2 column halves][128 pixels]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const long t0 = (long)blockIdx.x * SC_NPX;
  // ---- the block's 128 rows of the patch matrix: 512 16-byte slots
  if (p.A0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256, px = idx >> 2, c8 = idx & 3;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (t0 + px < p.T) v = *(const u32x4*)((const char*)p.A0 + ((t0 + px) * 32 + c8 * 8) * 2);
      *(u32x4*)(a0img + px * HP + c8 * 16) = v;
    }
  } else {
    // ... built from the image: column j = tap (j / 5, j % 5) of the 5 x 5 neighbourhood, zero outside the image
    // (srhip_im2col_c1's matrix, data.hip); a thread = (pixel, taps 16 half .. + 15)
    const int px = tid >> 1, hf = tid & 1;
    const long t = t0 + px;
    h16x8 v0, v1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v0[e] = (_Float16)0.f; v1[e] = (_Float16)0.f; }
    if (t < p.T) {
      const int xx = (int)(t % p.W), yy = (int)((t / p.W) % p.H);
      const float* ib = p.img + (t / ((long)p.W * p.H)) * p.H * p.W;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = 16 * hf + e;
        float v = 0.f;
        const int sy = yy + j / 5 - 2, sx = xx + j % 5 - 2;
        if (j < 25 && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v = ib[(long)sy * p.W + sx];
        if (e < 8) v0[e] = (_Float16)v; else v1[e - 8] = (_Float16)v;
      }
    }
    *(h16x8*)(a0img + px * HP + hf * 32) = v0;
    *(h16x8*)(a0img + px * HP + hf * 32 + 16) = v1;
  }
  const float* const winv1 = (const float*)((const char*)p.W1 + 2 * p.plane1);
  const float* const winv2 = (const float*)((const char*)p.W2 + 2 * p.plane2);
  const long wrows1 = 9L * SC_N1, wrows2 = 9L * SC_N2;
  // layer 1, transposed: A = weight rows (hidden units 64 hc + 32 wn + 16 j + c), B = pixel rows 16 (4 wm + i) + c
  auto load_w1 = [&](int hc, u32x4 (&f)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      f[j] = *(const u32x4*)((const char*)p.W1 + ((long)(g >> 1) * wrows1 + 4L * SC_N1 + 64 * hc + 32 * wn + 16 * j + c) * 32 + (g & 1) * 16);
  };
  // layer 2: B = weight rows (outputs 64 wn + 16 j2 + c), k = hidden units 64 hc + 32 s + 8 g ..
  auto load_w2 = [&](int hc, u32x4 (&f)[2][4]) {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        f[sb][j] = *(const u32x4*)((const char*)p.W2 + ((long)(4 * hc + 2 * sb + (g >> 1)) * wrows2 + 4L * SC_N2 + 64 * wn + 16 * j + c) * 32 + (g & 1) * 16);
  };
  f32x4 acc2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fw1[2], fw2[2][4];
  load_w1(0, fw1);
  load_w2(0, fw2);
  __syncthreads();
  for (int hc = 0; hc < SC_N1 / 64; ++hc) {
    unsigned char* const img = h1img + (hc & 1) * 2 * SC_NPX * HP;
    // ---- layer 1 of the chunk, one 16-pixel row tile at a time: acc1[j][e] = hidden unit 64 hc + 32 wn + 16 j + 4 g + e of
    // pixel 16 (4 wm + i) + c
    f32x4 wi[2], bi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u0 = 64 * hc + 32 * wn + 16 * j + 4 * g;
      wi[j] = ldg_f4(winv1 + u0);
      bi[j] = ldg_f4(p.b1 + u0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 fa0 = *(const u32x4*)(a0img + (16 * (4 * wm + i) + c) * HP + 16 * g);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 a1 = mfma16(fw1[j], fa0, f32x4{0.f, 0.f, 0.f, 0.f});
        _Float16 h4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h4[e] = (_Float16)fmaxf(a1[e] * wi[j][e] + bi[j][e], 0.f);
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        // sub-chunk wn (k = 32 wn + ..), k offset 16 j + 4 g inside it: 8 bytes
        *(h16x4*)(img + wn * SC_NPX * HP + (16 * (4 * wm + i) + c) * HP + (16 * j + 4 * g) * 2) = h16x4{h4[0], h4[1], h4[2], h4[3]};
      }
    }
    if (hc + 1 < SC_N1 / 64) load_w1(hc + 1, fw1);
    __syncthreads();                                  // the chunk's image is complete (the other buffer is free to be rewritten)
    // ---- layer 2 over the chunk's 64 hidden units
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 fa = *(const u32x4*)(img + sb * SC_NPX * HP + (16 * (4 * wm + i) + c) * HP + 16 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = mfma16(fa, fw2[sb][j], acc2[i][j]);
      }
    if (hc + 1 < SC_N1 / 64) load_w2(hc + 1, fw2);
  }
  // ---- layer 2's epilogue and layer 3: y = sum_c relu(acc2 * winv2 + b2)[c] * w3[c] + b3
  float part[4][4];                                   // [i][e]: pixel 16 (4 wm + i) + 4 g + e, over this lane's four columns
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[i][e] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = 64 * wn + 16 * j + c;
    const float wv = winv2[col], bv = p.b2[col], w3 = p.w3[col];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[i][e] += fmaxf(acc2[i][j][e] * wv + bv, 0.f) * w3;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = part[i][e];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (c == 0) rsum[wn * SC_NPX + 16 * (4 * wm + i) + 4 * g + e] = v;
    }
  __syncthreads();
  if (tid < SC_NPX && t0 + tid < p.T) p.y[t0 + tid] = rsum[tid] + rsum[SC_NPX + tid] + p.b3[0];
}

// 1 -> Co conv (f32 image in, fp16 features out), act 0 none | 1 ReLU | 2 LeakyReLU(alpha).  A thread owns 8 output channels of one pixel.
__global__ void __launch_bounds__(256) k_cin1_h16(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, _Float16* __restrict__ y, long ldy, int B,
                                                  int H, int W, int Co, int act, float alpha) {
  extern __shared__ float wl[];                  // [9][Co] + [Co]
  for (int i = threadIdx.x; i < 9 * Co; i += 256) wl[(i % 9) * Co + i / 9] = w[i];
  for (int i = threadIdx.x; i < Co; i += 256) wl[9 * Co + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int G = Co >> 3;
  const long n = (long)B * H * W * G;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int gq = (int)(i % G);
    const long pix = i / G;
    const int xx = (int)(pix % W);
    const long r = pix / W;
    const int yy = (int)(r % H);
    const long b = r / H;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wl[9 * Co + gq * 8 + e];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      const float xv = x[(b * H + sy) * W + sx];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += xv * wl[t * Co + gq * 8 + e];
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(act == 1 ? fmaxf(v[e], 0.f) : (act == 2 && v[e] < 0.f ? v[e] * alpha : v[e]));
    *(h16x8*)(y + pix * ldy + gq * 8) = o;
  }
}

// Ci -> 1 conv (fp16 features in, f32 image out) + bias + optional f32 addend image.  Eight lanes per pixel, each over an
// eighth of the channels (16-byte loads), summed with three shuffles.
__global__ void __launch_bounds__(256) k_cout1_h16(const _Float16* __restrict__ x, long ldx, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ add,
                                                   const float* __restrict__ in_bn, float* __restrict__ y, int B, int H, int W,
                                                   int Ci) {
  extern __shared__ float wl[];                  // [9][Ci] weights, [3][Ci] BatchNorm mean / k / beta of the input prologue
  float* const cf = wl + 9 * Ci;
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  h16x2* const wh = (h16x2*)(cf + 3 * Ci);       // [9][Ci / 2] the weights as fp16 pairs (no prologue: v_dot2_f32_f16)
  for (int i = threadIdx.x; i < 9 * Ci; i += 256) wl[(i % 9) * Ci + i / 9] = w[i];
  if (!in_bn)
    for (int i = threadIdx.x; i < 9 * Ci / 2; i += 256) {
      const int t = i / (Ci / 2), c2 = i - t * (Ci / 2);
      wh[i] = h16x2{(_Float16)w[(2 * c2) * 9 + t], (_Float16)w[(2 * c2 + 1) * 9 + t]};
    }
  if (in_bn)
    for (int i = threadIdx.x; i < Ci; i += 256) { cf[i] = in_bn[i]; cf[Ci + i] = in_bn[2 * Ci + i]; cf[2 * Ci + i] = in_bn[3 * Ci + i]; }
  __syncthreads();
  const long n = (long)B * H * W;
  const int sub = threadIdx.x & 7;
  for (long pix = blockIdx.x * 32L + (threadIdx.x >> 3); pix < n + 31; pix += (long)gridDim.x * 32) {   // whole groups of 8 lanes stay together
    const long pc = min(pix, n - 1);
    const int xx = (int)(pc % W);
    const long r = pc / W;
    const int yy = (int)(r % H);
    const long b = r / H;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      const _Float16* row = x + ((b * H + sy) * W + sx) * ldx;
      for (int c0 = sub * 8; c0 < Ci; c0 += 64) {
        const h16x8 xv = *(const h16x8*)(row + c0);
        if (in_bn) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xe = fmaxf(((float)xv[e] - cf[c0 + e]) * cf[Ci + c0 + e] + cf[2 * Ci + c0 + e], 0.f);
            s += xe * wl[t * Ci + c0 + e];
          }
        } else {                                 // four packed dot products (fp16 pairs, f32 accumulate) per 16 bytes
          const h16x2* wp = wh + t * (Ci / 2) + (c0 >> 1);
#pragma unroll
          for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2(h16x2{xv[2 * e], xv[2 * e + 1]}, wp[e], s, false);
        }
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0 && pix < n) y[pix] = s + (bias ? bias[0] : 0.f) + (add ? add[pix] : 0.f);
  }
}

}  // namespace

int sr_conv3x3_h16(ConvH16Args& p, hipStream_t st) {
  SR_REQUIRE(p.X && p.Wb && p.Y, "conv3x3_h16: null operand");
  SR_REQUIRE(p.K % 32 == 0 && p.K >= 32 && p.K <= 4096 && p.N % 64 == 0 && p.N <= 4096,
             "conv3x3_h16: Cin = %d (a multiple of 32), Cout = %d (a multiple of 64), both <= 4096", p.K, p.N);
  SR_REQUIRE(p.ldx % 8 == 0 && p.ldy % 8 == 0 && (!p.R || p.ldr % 8 == 0), "conv3x3_h16: pixel pitches must be multiples of 8 halves");
  SR_REQUIRE(p.epi == 0 || p.epi == 1 || p.epi == 6 || ((p.epi == 2 || p.epi == 8) && p.R),
             "conv3x3_h16: epilogue %d (0, 1, 6, 2 / 8 with R)", p.epi);
  SR_REQUIRE(!p.ps || (p.N % 256 == 0 && !p.R), "conv3x3_h16 + PixelShuffle(2): Cout %% 256 == 0 and no residual (Cout=%d)", p.N);
  SR_REQUIRE(p.B > 0 && p.H > 0 && p.Wd > 0, "conv3x3_h16: empty image");
  SR_REQUIRE((long)p.B * p.H * p.Wd * p.ldx < (1L << 31), "conv3x3_h16: input larger than 4 GiB (32-bit staging offsets)");
  p.Kp = (p.K + 31) / 32 * 32;
  p.plane_bytes = 9L * p.N * p.Kp * 2;
  const long blocks128 = (long)sr_cdiv(p.Wd, 16) * sr_cdiv(p.H, 8) * p.B * (p.N / 64);
  const int rw = blocks128 >= 1024 ? 4 : 2;
  p.tiles_x = sr_cdiv(p.Wd, 16);
  p.tiles_y = sr_cdiv(p.H, 2 * rw);
  dim3 grid((unsigned)((long)p.tiles_x * p.tiles_y * p.B * (p.N / 64)));
  if (p.center_only) {
    if (rw == 4) hipLaunchKernelGGL(k_conv1x1_h16<4>, grid, dim3(256), h16_lds(4), st, p);
    else hipLaunchKernelGGL(k_conv1x1_h16<2>, grid, dim3(256), h16_lds(2), st, p);
  } else if (rw == 4) {
    // more than four rounds of blocks: the persistent form (768 blocks = three per CU, a multiple of 8 for the XCD order) --
    // an experiment, OFF: measured 3-5 % SLOWER than the one-tile blocks (VDSR --amp 1921 -> 1836 patches/s, DRRN 212 -> 205,
    // MemNet 65.2 -> 62.7).  The launch is not waiting for its first halo chunk: per 128-pixel tile the four waves pull
    // 147 KB of weight fragments through the CU's 64-B/clk vector-memory path (70 us of a 215-us VDSR layer), read 290 KB of
    // A fragments from LDS and issue 1.1 us of MFMAs -- three throughput limits of the same size that overlap only partly.
    static const int pers = [] { const char* e = sr_getenv("SRHIP_H16_PERSISTENT"); return e ? atoi(e) : 0; }();
    const long ntiles = (long)grid.x;
    if (pers && ntiles >= 4 * 768 && ntiles < (1L << 30)) {
      if (p.in_bn) hipLaunchKernelGGL(k_conv3x3_h16p<true>, dim3(768), dim3(256), h16_lds(4), st, p, (int)ntiles);
      else hipLaunchKernelGGL(k_conv3x3_h16p<false>, dim3(768), dim3(256), h16_lds(4), st, p, (int)ntiles);
    }
    else hipLaunchKernelGGL(k_conv3x3_h16<4>, grid, dim3(256), h16_lds(4), st, p);
  } else hipLaunchKernelGGL(k_conv3x3_h16<2>, grid, dim3(256), h16_lds(2), st, p);
  SR_LAUNCH_CHECK("k_conv3x3_h16");
  return 0;
}

int sr_conv_cin1_h16(const float* x, const float* w, const float* bias, void* y, long ldy, int B, int H, int W, int Co, int act,
                     float alpha, hipStream_t st) {
  SR_REQUIRE(x && w && y, "conv_cin1_h16: null operand");
  SR_REQUIRE(Co % 8 == 0 && Co <= 1024 && ldy % 8 == 0, "conv_cin1_h16: Cout = %d (a multiple of 8, <= 1024)", Co);
  const long n = (long)B * H * W * (Co / 8);
  if (n <= 0) return 0;
  const int grid = (int)(n / 256 + 1 < 8192 ? n / 256 + 1 : 8192);
  hipLaunchKernelGGL(k_cin1_h16, dim3(grid), dim3(256), (size_t)10 * Co * 4, st, x, w, bias, (_Float16*)y, ldy, B, H, W, Co, act, alpha);
  SR_LAUNCH_CHECK("k_cin1_h16");
  return 0;
}

int sr_conv_cout1_h16(const void* x, long ldx, const float* w, const float* bias, const float* add, const float* in_bn, float* y,
                      int B, int H, int W, int Ci, hipStream_t st) {
  SR_REQUIRE(x && w && y, "conv_cout1_h16: null operand");
  SR_REQUIRE(Ci % 64 == 0 && Ci <= 1024 && ldx % 8 == 0, "conv_cout1_h16: Cin = %d (a multiple of 64, <= 1024)", Ci);
  const long n = (long)B * H * W;
  if (n <= 0) return 0;
  const int grid = (int)(n / 32 + 1 < 16384 ? n / 32 + 1 : 16384);
  hipLaunchKernelGGL(k_cout1_h16, dim3(grid), dim3(256), (size_t)12 * Ci * 4 + (size_t)9 * Ci * 2, st, (const _Float16*)x, ldx, w, bias, add, in_bn, y, B, H, W, Ci);
  SR_LAUNCH_CHECK("k_cout1_h16");
  return 0;
}

int sr_srcnn_h16(const void* a0, const float* img, int B, int H, int W, const void* W1h, const float* b1, const void* W2h,
                 const float* b2, const float* w3, const float* b3, float* y, long T, hipStream_t st) {
  SR_REQUIRE((a0 || img) && W1h && b1 && W2h && b2 && w3 && b3 && y && T > 0, "srcnn_h16: null operand");
  SR_REQUIRE(a0 || (B > 0 && H > 0 && W > 0 && (long)B * H * W == T), "srcnn_h16: image %d x %d x %d for %ld pixels", B, H, W, T);
  SrcnnH16Args p;
  p.A0 = (const _Float16*)a0; p.img = img; p.B = B; p.H = H; p.W = W; p.W1 = (const unsigned short*)W1h; p.plane1 = 9L * SC_N1 * 32 * 2; p.b1 = b1;
  p.W2 = (const unsigned short*)W2h; p.plane2 = 9L * SC_N2 * 1024 * 2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.y = y; p.T = T;
  hipLaunchKernelGGL(k_srcnn_h16, dim3((unsigned)((T + SC_NPX - 1) / SC_NPX)), dim3(256), SC_LDS, st, p);
  SR_LAUNCH_CHECK("k_srcnn_h16");
  return 0;
}
