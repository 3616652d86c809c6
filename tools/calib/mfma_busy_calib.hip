// Calibration of SQ_VALU_MFMA_BUSY_CYCLES for the two MFMA shapes this library uses: a kernel that does nothing but N
// back-to-back MFMAs per wave, one wave per SIMD on every CU.  Run under
//   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d out -- ./mfma_busy_calib
// and divide the counter by (duration x clock x 1024 SIMDs): what a SATURATED matrix pipe reads for each shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256, 1) k_calib_16x16x32(float* out, int n) {
  f16x8 a, b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int q = 0; q < 8; ++q) b[q][i] = (_Float16)(0.5f - i * 0.01f + q * 0.125f);      // eight different operands: eight chains
  }
  f32x4 c[8];
  for (int q = 0; q < 8; ++q) c[q] = f32x4{(float)q, 0, 0, 0};
  for (int i = 0; i < n; ++i) {          // inline asm: the compiler rotates / merges accumulator chains of the builtin form
#pragma unroll
    for (int q = 0; q < 8; ++q)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a), "v"(b[q]));
  }
  float s = 0.f;
  for (int q = 0; q < 8; ++q) s += c[q][q & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256, 1) k_calib_32x32x16(float* out, int n) {
  f16x8 a, b[4];
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int q = 0; q < 4; ++q) b[q][i] = (_Float16)(0.5f - i * 0.01f + q * 0.125f);
  }
  f32x16 c[4];
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 16; ++i) c[q][i] = (float)q;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a), "v"(b[q]));
  }
  out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    k_calib_16x16x32<<<256, 256>>>(d, 10000);      // 80,000 MFMAs per wave
    k_calib_32x32x16<<<256, 256>>>(d, 10000);      // 40,000 MFMAs per wave
  }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  hipEventRecord(e0); k_calib_16x16x32<<<256, 256>>>(d, 10000); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("16x16x32: 80000 MFMAs per wave in %.3f ms = %.1f ns per MFMA\n", ms, ms * 1e6 / 80000);
  hipEventRecord(e0); k_calib_32x32x16<<<256, 256>>>(d, 10000); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("32x32x16: 40000 MFMAs per wave in %.3f ms = %.1f ns per MFMA\n", ms, ms * 1e6 / 40000);
  return 0;
}
