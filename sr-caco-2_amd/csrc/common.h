// Shared device/host helpers for libsrhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/srhip.h"   // every entry point is defined against its public prototype (and takes its default visibility from it)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SR_WAVE 64

// ---- error plumbing (include/srhip.h: every entry point returns 0 or <0) ----
extern "C" const char* srhip_last_error(void);
int sr_fail(int code, const char* fmt, ...);
#define SR_REQUIRE(cond, ...) \
  do { if (!(cond)) return sr_fail(-22, __VA_ARGS__); } while (0)
#define SR_LAUNCH_CHECK(name) \
  do { hipError_t e_ = hipGetLastError(); \
       if (e_ != hipSuccess) return sr_fail(-5, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// Tuning and ablation switches are read from the environment only by builds made with `make EXPERIMENTS=1`
// (-DSRHIP_EXPERIMENTS; tools/ab_*.sh load such a build through SRHIP_LIB): the shipped library takes its defaults.
#ifdef SRHIP_EXPERIMENTS
#include <stdlib.h>
static inline const char* sr_getenv(const char* name) { return getenv(name); }
// the library is built with -fvisibility=hidden: include/srhip.h's prototypes are the only exports (the header pushes default
// visibility around them).  The stamp / ablation hooks of tools/mb_*_phases.py are exported by the experiments build only.
#define SR_DEBUG_EXPORT extern "C" __attribute__((visibility("default")))
#else
static inline const char* sr_getenv(const char*) { return nullptr; }
#define SR_DEBUG_EXPORT extern "C"
#endif

// Stores of tensors nobody reads again before they have left every cache (saved activations of a training step: h, dh,
// gelu(h), qkv): with -DSRHIP_NT they carry the `nt` (streaming) hint, so that they do not push the rows a kernel re-reads
// (residual rows, its own attention output) out of the L2.  A build variant (make EXTRA=-DSRHIP_NT=1): measured, see DESIGN.
#ifdef SRHIP_NT
#define SR_ST_STREAM(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define SR_ST_STREAM(PTR, VAL) (*(PTR) = (VAL))
#endif

static inline int sr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Neutral operands for optional per-row prologue data.  A conditional load whose
// result merges with a constant forces the compiler to wait for the load at the
// branch join (s_waitcnt vmcnt(0) in the middle of a prefetch); selecting the
// ADDRESS instead keeps the load unconditional and the prefetch asynchronous.
//   [0..1] = {mean 0, rstd 1}   [2..3] = {0, 0}   [1] = scale 1
__device__ __attribute__((aligned(16))) const float k_sr_neutral[4] = {0.f, 1.f, 0.f, 0.f};

// Loads through a pointer the compiler cannot prove global (a select between a
// kernel-argument pointer and a __device__ constant, a pointer read from a struct
// passed by reference) compile to FLAT loads, which also count against the LDS
// counter and serialise with ds_* traffic.  These force the global path.
typedef const __attribute__((address_space(1))) float* sr_gptr_f;
typedef float sr_f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) sr_f32x2* sr_gptr_f2;
typedef const __attribute__((address_space(1))) f32x4* sr_gptr_f4;
__device__ __forceinline__ float ldg_f(const float* p) { return *(sr_gptr_f)p; }
__device__ __forceinline__ float2 ldg_f2(const float* p) {
  const sr_f32x2 v = *(sr_gptr_f2)p;
  return float2{v.x, v.y};
}
__device__ __forceinline__ f32x4 ldg_f4(const void* p) { return *(sr_gptr_f4)p; }

// ---- 3-way bf16 split of f32 pairs (the operand form of the bf16x3 MFMA kernels) ----
//   x = h + m + l, each part the bf16 rounding (RNE) of the remaining residual; the
//   residuals are exact in f32.  Packed pairs: element 0 in the low half.
typedef __bf16 sr_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned sr_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  // SCALAR subtractions on purpose, and the library is built with -fno-slp-vectorize: packed f32 VALU
  // (v_pk_add_f32 / v_pk_mul_f32) beside MFMAs is an anti-lever on this part (MI355X_MICROARCH.md, cycle table:
  // "2 v_pk_add_f32 per gap +26 cyc vs 2 v_fma_f32"), and hipcc packs adjacent scalar adds by itself under plain -O3.
  // Same box, whole training step: +2.4 % (SwinIR), +2.7 % (EDSR x8) against the packed form.
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(sr_f32x2{x0, x1}, sr_bf16x2));
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(sr_f32x2{r0, r1}, sr_bf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(sr_f32x2{s0, s1}, sr_bf16x2));
}

// ---- XCD-aware block order ----
// Hardware deals consecutive block indices to the 8 XCDs (each with its own L2) round robin.
// Blocks that share data -- neighbouring image tiles and their halos, the heads of one
// attention window -- have consecutive LOGICAL indices: give XCD x the x-th contiguous run of
// logical indices, so that they meet in one L2.  Bijection for any grid size n.
__device__ __forceinline__ int sr_xcd_block(int B, int n) {
  const int q = n >> 3, rem = n & 7, xcd = B & 7;
  return xcd * q + min(xcd, rem) + (B >> 3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global loads and stores
// (s_waitcnt vmcnt(0)): rows prefetched for the next tile would be waited for at the first barrier after their issue.
__device__ __forceinline__ void sr_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- wave-level reductions (wave = 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU and its derivative (nn.GELU default, network_swinir.py:30)
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// x Phi(x) on PAIRS with one exp per element (Abramowitz & Stegun 7.1.26, |Phi error| <= 1.5e-7): the activation of the fused
// MLP kernels (mlp_f16.hip) and of the weight-gradient prologue that recomputes gelu(h) from the saved pre-activation
// (gemm_tnb.hip, b_mode 2) -- one definition, so both see the same bits.
// The same two functions on PAIRS (v_pk_mul / v_pk_fma / v_pk_add_f32: two lanes' worth of f32 arithmetic per issue slot).
// The activation phase runs no MFMA and is bound by vector issue (96 hidden units per lane x ~17 instructions): packed,
// the non-transcendental part is half the slots.  (Beside MFMAs packed f32 is an anti-lever -- the library is built with
// -fno-slp-vectorize for that -- so the packing is explicit and only here.)
__device__ __forceinline__ sr_f32x2 gelu_poly2(sr_f32x2 x, sr_f32x2& e1) {
  const sr_f32x2 ax = sr_f32x2{fabsf(x.x), fabsf(x.y)};
  const sr_f32x2 q = (x * x) * (-0.5f * 1.44269504088896340736f);        // exp(-x^2 / 2) = 2^q
  e1 = sr_f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const sr_f32x2 d = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  const sr_f32x2 t = sr_f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const sr_f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const sr_f32x2 om = 1.0f - poly * e1;
  return sr_f32x2{copysignf(om.x, x.x), copysignf(om.y, x.y)} * 0.5f + 0.5f;        // Phi(x)
}
__device__ __forceinline__ sr_f32x2 gelu_fast2(sr_f32x2 x) {
  sr_f32x2 e1;
  return x * gelu_poly2(x, e1);
}

// 32x32x2 f32 MFMA: exact f32 fma chain (MI355X_MICROARCH "Matrix cores").
// lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
