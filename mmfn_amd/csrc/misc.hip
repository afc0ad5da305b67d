// Library info + small utility kernels (fill, scale-add) used by the host orchestration.
#include <algorithm>

#include "common.h"

namespace {
__global__ void noop_kernel() {}

__global__ void fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// y = a*x + b*y
__global__ void axpby_kernel(float* y, const float* x, float a, float b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0.0f ? 0.0f : b * y[i]);
}
__global__ void rng_advance_kernel(uint64_t* state) { state[1] += 1; }
}  // namespace

extern "C" int mmfn_abi_version(void) { return 1; }
extern "C" int mmfn_sizeof_gemm_desc(void) { return (int)sizeof(mmfn_gemm_desc); }

extern "C" int mmfn_device_selftest(void* stream) {
  hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_fill_f32(float* p, float v, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, v, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_axpby_f32(float* y, const float* x, float a, float b, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, x, a, b, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_rng_advance(uint64_t* state, void* stream) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  MMFN_LAUNCH_CHECK();
  return 0;
}
