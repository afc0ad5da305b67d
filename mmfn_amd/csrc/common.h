// Shared device/host helpers for the MMFN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmfn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- element-type-generic accessors (fp32 path: T = float; bf16 training mode: T = bf16_t, stored as raw 16-bit words).
// Every streaming kernel moves 4 consecutive elements per access: 16 bytes of fp32 or 8 bytes of bf16; arithmetic is fp32.
typedef __bf16 bf16_t;
__device__ __forceinline__ unsigned short mmfn_f2bf(float f) {   // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ f32x4 ldx4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldx4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 r;
  r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xFFFF0000u);
  r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ void stx4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void stx4(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = (unsigned)mmfn_f2bf(v[0]) | ((unsigned)mmfn_f2bf(v[1]) << 16);
  u.y = (unsigned)mmfn_f2bf(v[2]) | ((unsigned)mmfn_f2bf(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float ldx1(const float* p) { return *p; }
__device__ __forceinline__ float ldx1(const bf16_t* p) { return __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short*>(p)) << 16); }
__device__ __forceinline__ void stx1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stx1(bf16_t* p, float v) { *reinterpret_cast<unsigned short*>(p) = mmfn_f2bf(v); }
// N consecutive elements (N = 1, 2, 4) as floats
template <int N, typename T> struct VecIO;
template <typename T> struct VecIO<4, T> {
  typedef f32x4 vec;
  static __device__ __forceinline__ vec ld(const T* p) { return ldx4(p); }
  static __device__ __forceinline__ void st(T* p, vec v) { stx4(p, v); }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <> struct VecIO<2, float> {
  typedef f32x2 vec;
  static __device__ __forceinline__ vec ld(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
  static __device__ __forceinline__ void st(float* p, vec v) { *reinterpret_cast<f32x2*>(p) = v; }
};
template <> struct VecIO<2, bf16_t> {
  typedef f32x2 vec;
  static __device__ __forceinline__ vec ld(const bf16_t* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    vec r; r[0] = __uint_as_float(u << 16); r[1] = __uint_as_float(u & 0xFFFF0000u); return r;
  }
  static __device__ __forceinline__ void st(bf16_t* p, vec v) {
    *reinterpret_cast<unsigned*>(p) = (unsigned)mmfn_f2bf(v[0]) | ((unsigned)mmfn_f2bf(v[1]) << 16);
  }
};
typedef float f32x1 __attribute__((ext_vector_type(1)));
template <typename T> struct VecIO<1, T> {
  typedef f32x1 vec;
  static __device__ __forceinline__ vec ld(const T* p) { vec r; r[0] = ldx1(p); return r; }
  static __device__ __forceinline__ void st(T* p, vec v) { stx1(p, v[0]); }
};

// BatchNorm apply y = x * alpha + beta with alpha = w * rstd, beta = b - mean * alpha: ONE spelling (explicit fma) for every
// kernel that evaluates it - the apply pass (norm.hip), the Winograd input transform that applies the producer's BatchNorm on the
// fly (winograd.hip) and the backward kernels that recompute the ReLU mask of an output that was never written to HBM - so
// that a recomputed sign is the forward's sign bit for bit.
__device__ __forceinline__ float mmfn_bn_alpha(float w, float rstd) { return w * rstd; }
__device__ __forceinline__ float mmfn_bn_beta(float b, float mean, float alpha) { return fmaf(-mean, alpha, b); }
__device__ __forceinline__ float mmfn_bn_affine(float x, float alpha, float beta) { return fmaf(x, alpha, beta); }

#define MMFN_LAUNCH_CHECK()                        \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Counter-based RNG: one 32-bit draw per (seed, step, stream, index); the backward pass regenerates the same mask instead of
// storing it.  Strength: a dropout mask, not a general-purpose generator - the (seed, step, stream) key reaches a draw only through
// a 32-bit XOR salt, so two streams / steps whose salts agree in the bits above log2(numel) get masks that are XOR-index
// permutations of each other with equal keep counts (probability ~ numel / 2^32 per pair: ~1/256 for a 16 M-element tensor, i.e. it
// happens over a long run).  Statistically harmless for dropout (each mask is still Bernoulli(p) per element); the masks of different
// streams are NOT independent in the cryptographic sense and the tests only claim the Bernoulli rate and pairwise decorrelation.  The draw is a 32-bit mixer (lowbias32: two 32-bit multiplies) of the index, salted with a hash of the 64-bit
// (seed, step, stream) key - the salt is uniform over a launch, so it costs scalar instructions once.  Rounds 1-4 ran the 64-bit
// splitmix finaliser per element (three 64-bit multiplies = a dozen quarter-rate v_mul_lo/hi_u32 + ~15 more instructions): with
// 36-96 elements per lane that was 9-10 k cycles of every attention kernel and the longest part of a dropout epilogue.
__device__ __forceinline__ uint32_t mmfn_hash32(uint32_t x) {   // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t mmfn_rng_salt(uint64_t key) { return mmfn_hash32((uint32_t)key ^ mmfn_hash32((uint32_t)(key >> 32))); }
__device__ __forceinline__ uint32_t mmfn_rng_u32(uint64_t key, uint64_t idx) {
  // (the high index word only matters for tensors of more than 2^32 elements; it is folded in rather than ignored)
  return mmfn_hash32(((uint32_t)idx ^ mmfn_rng_salt(key)) + (uint32_t)(idx >> 32) * 0x9E3779B1u);
}
__device__ __forceinline__ uint64_t mmfn_rng_key(const uint64_t* state, uint32_t stream) {
  // state[0] = seed, state[1] = step counter
  uint64_t k = state[0] * 0xD1342543DE82EF95ull + state[1] * 0xA24BAED4963EE407ull;
  return k ^ ((uint64_t)stream << 40) ^ (uint64_t)stream * 0x9E3779B1ull;
}
// keep-mask scale: 0 if dropped, 1/(1-p) if kept
__device__ __forceinline__ float mmfn_dropout_scale(uint64_t key, uint64_t idx, float p, float inv_keep) {
  float u = (float)(mmfn_rng_u32(key, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// exp(x) for the softmax kernels: branch-free, ~1-2 ulp.  expf() compiles to a call-like sequence with range branches;
// wrapped in a per-element mask it becomes 36 divergent blocks per lane (~65 instructions each: 12 k cycles of a 35 k-cycle
// attention kernel).  Here: x * log2(e) with the rounding error of the product (and of the constant) carried into a
// first-order correction, then one v_exp_f32.  exp(-inf) = 0, no NaN for finite or -inf input.
__device__ __forceinline__ float mmfn_exp(float x) {
  const float kHi = 1.44269502162933349609375f;   // float(log2 e)
  const float kLo = 1.925963033500347e-08f;       // log2 e - kHi
  const float t = x * kHi;
  float r = fmaf(x, kHi, -t);                     // exact remainder of the rounded product
  r = fmaf(x, kLo, r);
  const float e = __builtin_amdgcn_exp2f(t);
  // x = -inf: t = -inf, the fmaf above is -inf - (-inf) = NaN, so take the correction only where t is finite
  return t > -3.0e38f ? fmaf(e, r * 0.693147180559945f, e) : 0.0f;
}

__device__ __forceinline__ float mmfn_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float mmfn_gelu_grad(float x) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  return cdf + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
