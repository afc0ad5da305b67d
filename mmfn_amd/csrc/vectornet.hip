// VectorNet polyline sub-graph pooling (model_vec.py:269-282 Subgraph.forward):
//   y = MLP(x)                  [R, V, H]   (Linear + LayerNorm + ReLU: GEMM + norm kernels)
//   pooled = max_v y            [R, H]      first-max index kept for the backward
//   out = cat(y, pooled bcast)  [R, V, 2H]  (or, for the last layer, max_v of that = [pooled, pooled])
// One workgroup stages a group of polylines ([G, V, H] floats) in LDS, reduces the V sub-nodes
// of each polyline there, and streams the concatenated rows out with 16-byte stores.
// Padded (all-zero) nodes/lanes are NOT masked, exactly like the reference.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int HMAX = 64;

template <bool LAST>
__global__ __launch_bounds__(NT) void poly_pool_fwd_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                           uint8_t* __restrict__ arg, int R, int V, int H, int G) {
  extern __shared__ float tile[];  // [G][V][H]
  const int r0 = blockIdx.x * G;
  const int g_n = min(G, R - r0);
  const int n = g_n * V * H;
  const float* src = y + (size_t)r0 * V * H;
  for (int i = threadIdx.x * 4; i < n; i += NT * 4) *reinterpret_cast<f32x4*>(&tile[i]) = *reinterpret_cast<const f32x4*>(src + i);
  __syncthreads();
  __shared__ float pooled[NT];
  // thread -> (polyline g, channel c)
  for (int idx = threadIdx.x; idx < g_n * H; idx += NT) {
    const int g = idx / H, c = idx % H;
    float best = tile[(g * V) * H + c];
    int bi = 0;
    for (int v = 1; v < V; ++v) {
      const float t = tile[(g * V + v) * H + c];
      if (t > best || t != t) { best = t; bi = v; }
    }
    arg[(size_t)(r0 + g) * H + c] = (uint8_t)bi;
    if (LAST) {
      out[(size_t)(r0 + g) * 2 * H + c] = best;
      out[(size_t)(r0 + g) * 2 * H + H + c] = best;
    } else {
      pooled[idx] = best;
    }
  }
  if (!LAST) {
    __syncthreads();
    // out[r, v, 0:H] = y, out[r, v, H:2H] = pooled  (requires G*H <= NT)
    const int rows = g_n * V;
    const int hq = H / 4;
    for (int i = threadIdx.x; i < rows * 2 * hq; i += NT) {
      const int row = i / (2 * hq), q = i % (2 * hq);
      f32x4 v;
      if (q < hq) v = *reinterpret_cast<f32x4*>(&tile[row * H + q * 4]);
      else v = *reinterpret_cast<f32x4*>(&pooled[(row / V) * H + (q - hq) * 4]);
      *reinterpret_cast<f32x4*>(out + ((size_t)r0 * V + row) * 2 * H + q * 4) = v;
    }
  }
}

// gy[R,V,H] = gout[..., :H] + onehot(arg) * sum_v gout[..., v, H:]      (LAST: gout is [R, 2H])
template <bool LAST>
__global__ __launch_bounds__(NT) void poly_pool_bwd_kernel(const float* __restrict__ gout, const uint8_t* __restrict__ arg,
                                                           float* __restrict__ gy, int R, int V, int H) {
  const int64_t total = (int64_t)R * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % H);
    const int64_t r = i / H;
    const int a = arg[i];
    float gp = 0.f;
    if (LAST) {
      gp = gout[r * 2 * H + c] + gout[r * 2 * H + H + c];
      for (int v = 0; v < V; ++v) gy[(r * V + v) * H + c] = (v == a) ? gp : 0.f;
    } else {
      for (int v = 0; v < V; ++v) gp += gout[(r * V + v) * 2 * H + H + c];
      for (int v = 0; v < V; ++v) gy[(r * V + v) * H + c] = gout[(r * V + v) * 2 * H + c] + ((v == a) ? gp : 0.f);
    }
  }
}
}  // namespace

extern "C" int mmfn_polyline_pool_fwd_f32(const float* y, float* out, uint8_t* arg, int R, int V, int H, int last, void* stream) {
  if (H % 4 || H > HMAX || V < 1 || V > 255 || R <= 0) return MMFN_EINVAL;
  const int G = NT / H;  // polylines per workgroup (4 for H = 64)
  const size_t lds = (size_t)G * V * H * sizeof(float);
  if (lds > 96 * 1024) return MMFN_EINVAL;
  dim3 grid(ceil_div(R, G));
  if (last) hipLaunchKernelGGL(poly_pool_fwd_kernel<true>, grid, dim3(NT), lds, (hipStream_t)stream, y, out, arg, R, V, H, G);
  else hipLaunchKernelGGL(poly_pool_fwd_kernel<false>, grid, dim3(NT), lds, (hipStream_t)stream, y, out, arg, R, V, H, G);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_polyline_pool_bwd_f32(const float* gout, const uint8_t* arg, float* gy, int R, int V, int H, int last,
                                          void* stream) {
  if (R <= 0) return MMFN_EINVAL;
  const int64_t total = (int64_t)R * H;
  const int blocks = (int)(ceil_div64(total, NT) < 4096 ? ceil_div64(total, NT) : 4096);
  if (last) hipLaunchKernelGGL(poly_pool_bwd_kernel<true>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, gout, arg, gy, R, V, H);
  else hipLaunchKernelGGL(poly_pool_bwd_kernel<false>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, gout, arg, gy, R, V, H);
  MMFN_LAUNCH_CHECK();
  return 0;
}
