// Winograd F(2x2, 3x3) transforms for the stride-1 3x3 convolutions of the deep ResNet stages (layer3 / layer4:
// 256 and 512 channels at 16x16 and 8x8 pixels; torchvision BasicBlock convs, model_vec.py:509-593).
//
// y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 input patch d -> 2x2 output patch: the element-wise product summed over
// input channels is 16 independent [tiles x Cin] x [Cin x Cout] GEMMs (one batched launch of gemm_f32, 2.25x fewer
// MFMA FLOPs than the implicit GEMM and 16x more tiles in flight, so no split-K at 2048 / 512 output pixels per
// channel).  The three transforms here are plain HBM-bound streaming kernels (16-byte accesses, channels innermost).
// At 64 / 128 channels the 4x expansion of the transformed activations costs more than the FLOPs saved, so the
// host (ops.conv2d_fwd) only takes this path from 256 channels up.
#include <algorithm>

#include "common.h"

namespace {

constexpr int NT = 256;

// U[t][co][ci] = (G g G^T)[t], g = w[co][:, :, ci];  w is [Co][3][3][Ci], U is [16][Co][Ci]
__global__ __launch_bounds__(NT) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci) {
  const int64_t n = (int64_t)Co * Ci;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / Ci), ci = (int)(i % Ci);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = w[(((size_t)co * 3 + a) * 3 + b) * Ci + ci];
    float t[4][3];  // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0];
      const float u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
      const float u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
      const float u3 = t[a][2];
      U[((size_t)(a * 4 + 0) * Co + co) * Ci + ci] = u0;
      U[((size_t)(a * 4 + 1) * Co + co) * Ci + ci] = u1;
      U[((size_t)(a * 4 + 2) * Co + co) * Ci + ci] = u2;
      U[((size_t)(a * 4 + 3) * Co + co) * Ci + ci] = u3;
    }
  }
}

// V[t][tile][c] = (B^T d B)[t]; x is NHWC [B][H][W][C]; tile (b, i, j) reads rows 2i-1..2i+2, cols 2j-1..2j+2
__global__ __launch_bounds__(NT) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W,
                                                        int C) {
  const int cq = C >> 2, th = H >> 1, tw = W >> 1;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 2 * ii - 1 + a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xx = 2 * j - 1 + e;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        d[a][e] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + y) * W + xx) * C + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 r[4][4];  // B^T d
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[0][e] = d[0][e] - d[2][e];
      r[1][e] = d[1][e] + d[2][e];
      r[2][e] = d[2][e] - d[1][e];
      r[3][e] = d[1][e] - d[3][e];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 v0 = r[a][0] - r[a][2], v1 = r[a][1] + r[a][2], v2 = r[a][2] - r[a][1], v3 = r[a][1] - r[a][3];
      float* p = V + ((size_t)(a * 4) * T + tile) * C + c4;
      *reinterpret_cast<f32x4*>(p) = v0;
      *reinterpret_cast<f32x4*>(p + (size_t)T * C) = v1;
      *reinterpret_cast<f32x4*>(p + (size_t)2 * T * C) = v2;
      *reinterpret_cast<f32x4*>(p + (size_t)3 * T * C) = v3;
    }
  }
}

// y[b][2i+p][2j+q][c] = (A^T m A)[p][q] (+ res), m = Mt[:][tile][c];  Mt is [16][T][C]
__global__ __launch_bounds__(NT) void wino_output_kernel(const float* __restrict__ Mt, const float* __restrict__ res,
                                                         float* __restrict__ y, int B, int H, int W, int C) {
  const int cq = C >> 2, th = H >> 1, tw = W >> 1;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 s[2][4];  // A^T m
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* p = Mt + ((size_t)e * T + tile) * C + c4;
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(p), m1 = *reinterpret_cast<const f32x4*>(p + (size_t)4 * T * C);
      const f32x4 m2 = *reinterpret_cast<const f32x4*>(p + (size_t)8 * T * C), m3 = *reinterpret_cast<const f32x4*>(p + (size_t)12 * T * C);
      s[0][e] = m0 + m1 + m2;
      s[1][e] = m1 - m2 - m3;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x4 o0 = s[p][0] + s[p][1] + s[p][2], o1 = s[p][1] - s[p][2] - s[p][3];
      const size_t off = (((size_t)b * H + 2 * ii + p) * W + 2 * j) * C + c4;
      if (res) {
        o0 += *reinterpret_cast<const f32x4*>(res + off);
        o1 += *reinterpret_cast<const f32x4*>(res + off + C);
      }
      *reinterpret_cast<f32x4*>(y + off) = o0;
      *reinterpret_cast<f32x4*>(y + off + C) = o1;
    }
  }
}

inline int grid_for(int64_t n) { return (int)std::min<int64_t>((n + NT - 1) / NT, 65535 * 4); }

}  // namespace

extern "C" int mmfn_wino_weight_f32(const float* w, float* U, int Co, int Ci, void* stream) {
  if (!w || !U || Co <= 0 || Ci <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((int64_t)Co * Ci)), dim3(NT), 0, (hipStream_t)stream, w, U, Co, Ci);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_input_f32(const float* x, float* V, int B, int H, int W, int C, void* stream) {
  if (!x || !V || (H & 1) || (W & 1) || (C & 3) || B <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, x, V,
                     B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_output_f32(const float* Mt, const float* res, float* y, int B, int H, int W, int C, void* stream) {
  if (!Mt || !y || (H & 1) || (W & 1) || (C & 3) || B <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, Mt,
                     res, y, B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}
