// How fast can a kernel stream [T][180]-float token rows (720 B: no multiple of a 128-B line) against [T][192] rows (768 B) and
// against a flat buffer?  The fused SwinIR kernels and the grouped weight gradient read 180-float rows with 12-byte (W = 3 floats)
// or 16-byte lane loads and sit at 3.3 - 3.9 TB/s; k_axpby (flat dwordx4) reads 5.6.  One block per CU-slot, 16 rows per
// iteration and wave, every lane sums what it reads.
//   hipcc -O3 --offload-arch=gfx950 rowread_calib.hip -o rowread_calib && ./rowread_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f3 __attribute__((ext_vector_type(3)));
typedef f3 __attribute__((aligned(4))) f3u;
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: rows of 180 floats, 60 lanes x 12 B | 1: rows of 180 floats, 45 lanes x 16 B | 2: rows of 192 floats, 48 lanes x 16 B
// 3: flat, 64 lanes x 16 B
template <int MODE>
__global__ void __launch_bounds__(256) k_rows(const float* __restrict__ x, long T, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nw = (long)gridDim.x * 4, w = (long)blockIdx.x * 4 + wave;
  float s = 0.f;
  if (MODE == 3) {
    const long n4 = T * 180 / 4;
    for (long i = w * 64 + lane; i < n4; i += nw * 64 * 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = i + (long)u * nw * 64;
        if (j < n4) { const f4 v = ((const f4*)x)[j]; s += v.x + v.y + v.z + v.w; }
      }
    }
  } else {
    const int ld = MODE == 2 ? 192 : 180;
    for (long r0 = w * 16; r0 < T; r0 += nw * 16) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const long r = r0 + t;
        if (r >= T) break;
        if (MODE == 0) { if (lane < 60) { const f3 v = *(const f3u*)(x + r * ld + lane * 3); s += v.x + v.y + v.z; } }
        else if (MODE == 1) { if (lane < 45) { const f4 v = *(const f4*)(x + r * ld + lane * 4); s += v.x + v.y + v.z + v.w; } }
        else { if (lane < 48) { const f4 v = *(const f4*)(x + r * ld + lane * 4); s += v.x + v.y + v.z + v.w; } }
      }
    }
  }
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, const float* x, long T, float* out, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_rows<MODE>, dim3(blocks), dim3(256), 0, 0, x, T, out);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_rows<MODE>, dim3(blocks), dim3(256), 0, 0, x, T, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)T * (MODE == 2 ? 192 : 180) * 4.0;
  printf("%-44s blocks %5d: %7.1f us  %6.2f TB/s\n", name, blocks, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
}

int main() {
  const long T = 1 << 20;                       // 1 M tokens: 755 / 805 MB
  float *x, *out;
  hipMalloc(&x, T * 192 * 4);
  hipMalloc(&out, 4096);
  hipMemset(x, 0, T * 192 * 4);
  for (int blocks : {512, 2048}) {
    run<0>("rows of 180 floats, 60 lanes x 12 B", x, T, out, blocks);
    run<1>("rows of 180 floats, 45 lanes x 16 B", x, T, out, blocks);
    run<2>("rows of 192 floats, 48 lanes x 16 B", x, T, out, blocks);
    run<3>("flat, 64 lanes x 16 B", x, T, out, blocks);
  }
  return 0;
}
