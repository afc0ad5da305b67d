// Fused AdamW over the flat parameter buffer (run_steps/phase2_train_net.py:110,256:
// torch.optim.AdamW defaults lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2).
// HBM-bound: 28 B/param (read p,g,m,v; write p,m,v) in one pass with 16-byte accesses.
// The step count lives in device memory so the launch is hipGraph-replayable.
#include <algorithm>

#include "common.h"

namespace {
__global__ void step_advance_kernel(int64_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                                                    float eps, float wd, const int64_t* __restrict__ step, float grad_scale) {
  const double t = (double)*step;
  // scalar prep in fp64 like torch's Python-side arithmetic, then rounded once to fp32
  const double bc1 = 1.0 - pow((double)beta1, t);
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
  const float step_size = (float)((double)lr / bc1);
  const float decay = (float)(1.0 - (double)lr * (double)wd);
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * grad_scale;
      pv[e] *= decay;
      mv[e] = mv[e] + (gg - mv[e]) * (1.0f - beta1);
      vv[e] = vv[e] * beta2 + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] = pv[e] - step_size * (mv[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  // tail
  const int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float gg = g[i] * grad_scale;
    float pv = p[i] * decay;
    const float mv = m[i] + (gg - m[i]) * (1.0f - beta1);
    const float vv = v[i] * beta2 + (1.0f - beta2) * gg * gg;
    pv = pv - step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}
}  // namespace

extern "C" int mmfn_step_advance(int64_t* step, void* stream) {
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, const int64_t* step, float grad_scale, void* stream) {
  if (n <= 0) return 0;
  if (!step || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return MMFN_EINVAL;
  const int blocks = (int)std::min<int64_t>(ceil_div64(n / 4 + 1, 256), 4096);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, step, grad_scale);
  MMFN_LAUNCH_CHECK();
  return 0;
}

// ---- grouped AdamW: hyper-parameters in DEVICE memory -------------------------------------------------------------
// torch.optim.AdamW param_groups (the reference defines decay / no-decay groups, model_vec.py:179-209) + a learning rate
// that may change every step without re-capturing a hipGraph: each group's (lr, beta1, beta2, eps, weight_decay,
// grad_scale) is a row of `hyper` [n_groups][8] in HBM, read by the kernel; `group_of` holds one group id per FOUR
// consecutive parameters (every tensor of the flat layout starts 16-byte aligned, so a float4 never straddles tensors);
// NULL = everything in group 0.
#define MMFN_ADAMW_MAX_GROUPS 16
namespace {
struct GroupScalars { float step_size, decay, bc2_sqrt, beta1, beta2, eps, grad_scale, pad; };

__global__ __launch_bounds__(256) void adamw_groups_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n, const uint8_t* __restrict__ group_of,
                                                           const float* __restrict__ hyper, int n_groups,
                                                           const int64_t* __restrict__ step) {
  __shared__ GroupScalars gs[MMFN_ADAMW_MAX_GROUPS];
  if ((int)threadIdx.x < n_groups) {
    const float* h = hyper + threadIdx.x * 8;
    const double t = (double)*step;
    const double lr = (double)h[0], beta1 = (double)h[1], beta2 = (double)h[2];
    GroupScalars s;
    s.step_size = (float)(lr / (1.0 - pow(beta1, t)));
    s.decay = (float)(1.0 - lr * (double)h[4]);
    s.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
    s.beta1 = h[1]; s.beta2 = h[2]; s.eps = h[3]; s.grad_scale = h[5]; s.pad = 0.f;
    gs[threadIdx.x] = s;
  }
  __syncthreads();
  const int64_t n4 = (n + 3) >> 2;   // the flat buffers are padded to a multiple of 4 floats
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const GroupScalars s = gs[group_of ? group_of[i] : 0];
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * s.grad_scale;
      pv[e] *= s.decay;
      mv[e] = mv[e] + (gg - mv[e]) * (1.0f - s.beta1);
      vv[e] = vv[e] * s.beta2 + (1.0f - s.beta2) * gg * gg;
      const float denom = sqrtf(vv[e]) / s.bc2_sqrt + s.eps;
      pv[e] = pv[e] - s.step_size * (mv[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
}
}  // namespace

extern "C" int mmfn_adamw_groups_f32(float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of,
                                     const float* hyper, int n_groups, const int64_t* step, void* stream) {
  if (n <= 0) return 0;
  if (!step || !hyper || n_groups < 1 || n_groups > MMFN_ADAMW_MAX_GROUPS || (n & 3) ||
      (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15))
    return MMFN_EINVAL;
  const int blocks = (int)std::min<int64_t>(ceil_div64(n / 4, 256), 4096);
  hipLaunchKernelGGL(adamw_groups_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, group_of, hyper, n_groups,
                     step);
  MMFN_LAUNCH_CHECK();
  return 0;
}
