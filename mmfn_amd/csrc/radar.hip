// Radar graph-attention branch (model_rad.py:778-884, SpGraphAttentionLayer / SpGAT) — the small
// element-wise and row-wise pieces; the matrix products run on the batched fp32-MFMA GEMM.
//   GAT attention   e = LeakyReLU_alpha(Wh a);  logits = adj > 0 ? e : -9e15;  softmax; dropout
//   ELU (alpha 1), row-wise log_softmax over the 512 channels with the (8,8) spatial swap that
//   view(B,8,8,512).transpose(1,3) performs (model_rad.py:883-884).
#include "common.h"

namespace {
constexpr int NT = 256;

__global__ void elu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v > 0.f ? v : expm1f(v);
  }
}
// dx = g * (y > 0 ? 1 : y + 1)    (y = elu(x))
__global__ void elu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = y[i];
    dx[i] = g[i] * (v > 0.f ? 1.f : v + 1.f);
  }
}

// one wave per row of N <= 128 entries
__global__ __launch_bounds__(NT) void gat_softmax_fwd_kernel(const float* __restrict__ e_pre, const float* __restrict__ adj,
                                                             float alpha, float* __restrict__ p_out, float* __restrict__ att,
                                                             int R, int N, float drop_p, const uint64_t* __restrict__ rng_state,
                                                             uint32_t rng_stream) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  float v[2];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    v[i] = -INFINITY;
    if (c < N) {
      const float e = e_pre[(size_t)row * N + c];
      const float le = e > 0.f ? e : alpha * e;
      v[i] = adj[(size_t)row * N + c] > 0.f ? le : -9e15f;
      mx = fmaxf(mx, v[i]);
    }
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < N) { v[i] = expf(v[i] - mx); s += v[i]; }
  }
  s = wave_sum(s);
  uint64_t key = 0;
  if (drop_p > 0.f) key = mmfn_rng_key(rng_state, rng_stream);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      const float p = v[i] / s;
      p_out[(size_t)row * N + c] = p;
      float a = p;
      if (drop_p > 0.f) a *= mmfn_dropout_scale(key, (uint64_t)row * N + c, drop_p, 1.0f / (1.0f - drop_p));
      att[(size_t)row * N + c] = a;
    }
  }
}

__global__ __launch_bounds__(NT) void gat_softmax_bwd_kernel(const float* __restrict__ g_att, const float* __restrict__ p,
                                                             const float* __restrict__ e_pre, const float* __restrict__ adj,
                                                             float alpha, float* __restrict__ g_epre, int R, int N, float drop_p,
                                                             const uint64_t* __restrict__ rng_state, uint32_t rng_stream) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  uint64_t key = 0;
  if (drop_p > 0.f) key = mmfn_rng_key(rng_state, rng_stream);
  float gp[2], pv[2];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    gp[i] = pv[i] = 0.f;
    if (c < N) {
      float g = g_att[(size_t)row * N + c];
      if (drop_p > 0.f) g *= mmfn_dropout_scale(key, (uint64_t)row * N + c, drop_p, 1.0f / (1.0f - drop_p));
      gp[i] = g;
      pv[i] = p[(size_t)row * N + c];
      dot += g * pv[i];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < N) {
      float gl = pv[i] * (gp[i] - dot);
      if (!(adj[(size_t)row * N + c] > 0.f)) gl = 0.f;
      const float e = e_pre[(size_t)row * N + c];
      g_epre[(size_t)row * N + c] = gl * (e > 0.f ? 1.f : alpha);
    }
  }
}

// rows of C (<= 512, multiple of 64); in row (b, i*8+j) -> out row (b, j*8+i) when swap != 0
__global__ __launch_bounds__(NT) void log_softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C,
                                                             int swap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int pl = C >> 6;
  float v[8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pl) { v[i] = x[(size_t)row * C + lane + 64 * i]; mx = fmaxf(mx, v[i]); }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pl) s += expf(v[i] - mx);
  const float lse = mx + logf(wave_sum(s));
  int orow = row;
  if (swap) { const int b = row >> 6, a = row & 63; orow = (b << 6) + ((a & 7) << 3) + (a >> 3); }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pl) y[(size_t)orow * C + lane + 64 * i] = v[i] - lse;
}

// dx[row] = g[orow] - exp(y[orow]) * sum(g[orow])
__global__ __launch_bounds__(NT) void log_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                             float* __restrict__ dx, int R, int C, int swap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int pl = C >> 6;
  int orow = row;
  if (swap) { const int b = row >> 6, a = row & 63; orow = (b << 6) + ((a & 7) << 3) + (a >> 3); }
  float gv[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pl) { gv[i] = g[(size_t)orow * C + lane + 64 * i]; s += gv[i]; }
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pl) dx[(size_t)row * C + lane + 64 * i] = gv[i] - expf(y[(size_t)orow * C + lane + 64 * i]) * s;
}
int grid_for(int64_t n) { return (int)(ceil_div64(n, NT) < 4096 ? ceil_div64(n, NT) : 4096); }
}  // namespace

extern "C" int mmfn_elu_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(elu_fwd_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, x, y, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}
extern "C" int mmfn_elu_bwd_f32(const float* g, const float* y, float* dx, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(elu_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, g, y, dx, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}
extern "C" int mmfn_gat_softmax_fwd_f32(const float* e_pre, const float* adj, float alpha, float* p, float* att, int R, int N,
                                        float drop_p, const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  if (R <= 0 || N <= 0 || N > 128) return MMFN_EINVAL;
  hipLaunchKernelGGL(gat_softmax_fwd_kernel, dim3(ceil_div(R, NT / 64)), dim3(NT), 0, (hipStream_t)stream, e_pre, adj, alpha, p, att,
                     R, N, drop_p, rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}
extern "C" int mmfn_gat_softmax_bwd_f32(const float* g_att, const float* p, const float* e_pre, const float* adj, float alpha,
                                        float* g_epre, int R, int N, float drop_p, const uint64_t* rng_state,
                                        uint32_t rng_stream, void* stream) {
  if (R <= 0 || N <= 0 || N > 128) return MMFN_EINVAL;
  hipLaunchKernelGGL(gat_softmax_bwd_kernel, dim3(ceil_div(R, NT / 64)), dim3(NT), 0, (hipStream_t)stream, g_att, p, e_pre, adj,
                     alpha, g_epre, R, N, drop_p, rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}
extern "C" int mmfn_log_softmax_fwd_f32(const float* x, float* y, int R, int C, int swap, void* stream) {
  if (R <= 0 || C % 64 || C > 512 || (swap && (R % 64))) return MMFN_EINVAL;
  hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3(ceil_div(R, NT / 64)), dim3(NT), 0, (hipStream_t)stream, x, y, R, C, swap);
  MMFN_LAUNCH_CHECK();
  return 0;
}
extern "C" int mmfn_log_softmax_bwd_f32(const float* g, const float* y, float* dx, int R, int C, int swap, void* stream) {
  if (R <= 0 || C % 64 || C > 512 || (swap && (R % 64))) return MMFN_EINVAL;
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(ceil_div(R, NT / 64)), dim3(NT), 0, (hipStream_t)stream, g, y, dx, R, C, swap);
  MMFN_LAUNCH_CHECK();
  return 0;
}
