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
// out = y > 0 ? g : 0
__global__ void relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = y[i] > 0.0f ? g[i] : 0.0f;
}
__global__ void rng_advance_kernel(uint64_t* state) { state[1] += 1; }
// out[i] = in[i] * keepscale(i): the backward of an epilogue dropout (same index = row*N + col)
__global__ void dropout_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float p,
                                     const uint64_t* __restrict__ state, uint32_t stream_id) {
  const uint64_t key = mmfn_rng_key(state, stream_id);
  const float inv_keep = 1.0f / (1.0f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] * mmfn_dropout_scale(key, (uint64_t)i, p, inv_keep);
}
}  // namespace

extern "C" int mmfn_abi_version(void) { return 1; }
extern "C" int mmfn_sizeof_gemm_desc(void) { return (int)sizeof(mmfn_gemm_desc); }

extern "C" int mmfn_device_selftest(void* stream) {
  hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}

// w[Co][T][Ci] -> wt[Ci][T][Co] with the taps reversed (t -> T-1-t): the filter of the transposed convolution.
// One 32x32 (co, ci) tile per block through LDS so both the read (ci contiguous) and the write (co contiguous)
// are 128-byte runs.
__global__ __launch_bounds__(256) void conv_weight_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co,
                                                               int T, int Ci) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Co && ci < Ci) ? w[((size_t)co * T + t) * Ci + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Ci && co < Co) wt[((size_t)ci * T + (T - 1 - t)) * Co + co] = tile[tx][r];
  }
}

extern "C" int mmfn_conv_weight_flip_f32(const float* w, float* wt, int Co, int T, int Ci, void* stream) {
  if (Co <= 0 || T <= 0 || Ci <= 0 || !w || !wt) return MMFN_EINVAL;
  hipLaunchKernelGGL(conv_weight_flip_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32, T), dim3(256), 0, (hipStream_t)stream, w, wt,
                     Co, T, Ci);
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

extern "C" int mmfn_dropout_apply_f32(const float* in, float* out, int64_t n, float p, const uint64_t* rng_state,
                                      uint32_t rng_stream, void* stream) {
  if (n <= 0) return 0;
  if (!rng_state || p < 0.f || p >= 1.f) return MMFN_EINVAL;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, p, rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_relu_mask_f32(const float* g, const float* y, float* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, y, out, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}
