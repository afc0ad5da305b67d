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

// ---------------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): 6x6 input patches -> 4x4 outputs, 36 element-wise products (4x fewer MFMA FLOPs than the implicit GEMM,
// 2.25x expansion of the transformed activations).  Cook-Toom construction over the interpolation points 0, +-a, +-b, inf.
// Lavin & Gray's a = 1, b = 2 (rounds 1-3) has transform entries up to 8 and a fp32 error ~12x that of a direct convolution; the
// points a = 3/4, b = 3/2 (every entry of A^T and B^T still exact in binary) halve that (tools/experiments/winograd_points.py:
// 6.0x; Barabasz et al., "Error analysis and improving the accuracy of Winograd convolution"), measured on the training step
// as the per-stage cosine of the gradient to fp64 (tools/grad_cosine.py).  With
//   f_a = 2 a^2 (a^2 - b^2),  f_b = 2 b^2 (b^2 - a^2):
//   A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1]
//   G   = [1/(a^2 b^2) 0 0; (1, +-a, a^2)/f_a; (1, +-b, b^2)/f_b; 0 0 1]
//   B^T = [a^2b^2 0 -(a^2+b^2) 0 1 0; 0 -+ab^2 -b^2 +-a 1 0; 0 -+a^2b -a^2 +-b 1 0; 0 a^2b^2 0 -(a^2+b^2) 0 1]
#ifndef MMFN_WINO_A   // (ablation builds: -DMMFN_WINO_A=1.0f -DMMFN_WINO_B=2.0f is Lavin & Gray's set, tools/experiments/r4_points_ab.sh)
#define MMFN_WINO_A 0.75f
#define MMFN_WINO_B 1.5f
#endif
constexpr float WA = MMFN_WINO_A, WB = MMFN_WINO_B;
constexpr float WA2 = WA * WA, WB2 = WB * WB, WA3 = WA2 * WA, WB3 = WB2 * WB;
constexpr float WK0 = WA2 * WB2;                       // a^2 b^2 (= B^T[0][0]): 81/64
constexpr float WS2 = WA2 + WB2;                       // a^2 + b^2: 45/16
constexpr float WFA = 1.0f / (2.0f * WA2 * (WA2 - WB2));   // 1 / f_a = -128/243
constexpr float WFB = 1.0f / (2.0f * WB2 * (WB2 - WA2));   // 1 / f_b =   32/243
template <typename T>
__device__ __forceinline__ void f4_bt(const T* d, T* r) {  // r = B^T d, 6 -> 6
  const T ea = d[4] - WB2 * d[2], oa = d[3] - WB2 * d[1];   // rows +-a: (d4 - b^2 d2) +- a (d3 - b^2 d1)
  const T eb = d[4] - WA2 * d[2], ob = d[3] - WA2 * d[1];   // rows +-b: (d4 - a^2 d2) +- b (d3 - a^2 d1)
  r[0] = WK0 * d[0] - WS2 * d[2] + d[4];
  r[1] = ea + WA * oa;
  r[2] = ea - WA * oa;
  r[3] = eb + WB * ob;
  r[4] = eb - WB * ob;
  r[5] = WK0 * d[1] - WS2 * d[3] + d[5];
}
template <typename T>
__device__ __forceinline__ void f4_at(const T* m, T* o) {  // o = A^T m, 6 -> 4
  const T p = m[1] + m[2], q = m[1] - m[2], u = m[3] + m[4], v = m[3] - m[4];
  o[0] = m[0] + p + u;
  o[1] = WA * q + WB * v;
  o[2] = WA2 * p + WB2 * u;
  o[3] = WA3 * q + WB3 * v + m[5];
}
__device__ __forceinline__ void f4_g(const float* g, float* u) {  // u = G g, 3 -> 6
  const float ea = g[0] + WA2 * g[2], eb = g[0] + WB2 * g[2];
  u[0] = (1.0f / WK0) * g[0];
  u[1] = WFA * (ea + WA * g[1]);
  u[2] = WFA * (ea - WA * g[1]);
  u[3] = WFB * (eb + WB * g[1]);
  u[4] = WFB * (eb - WB * g[1]);
  u[5] = g[2];
}

__global__ __launch_bounds__(NT) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci) {
  const int64_t n = (int64_t)Co * Ci;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / Ci), ci = (int)(i % Ci);
    float t[6][3];  // G g (columns of g transformed)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float g[3], u[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) g[a] = w[(((size_t)co * 3 + a) * 3 + b) * Ci + ci];
      f4_g(g, u);
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a][b] = u[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float u[6];
      f4_g(t[a], u);
#pragma unroll
      for (int b = 0; b < 6; ++b) U[((size_t)(a * 6 + b) * Co + co) * Ci + ci] = u[b];
    }
  }
}

__global__ __launch_bounds__(NT) void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W,
                                                         int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 r[6][6];  // B^T d, built column by column so only one 6-vector of raw pixels is live
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int xx = 4 * j - 1 + e;
      f32x4 d[6], c[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int y = 4 * ii - 1 + a;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        d[a] = ok ? *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + y) * W + xx) * C + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      f4_bt(d, c);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][e] = c[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      f32x4 v[6];
      f4_bt(r[a], v);
#pragma unroll
      for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(V + ((size_t)(a * 6 + e) * T + tile) * C + c4) = v[e];
    }
  }
}

// The same transform reading the PRODUCER's convolution output and applying its BatchNorm (+ residual) (+ ReLU) on the fly:
//   d = [relu]( bn(x) [+ res] ) inside the image, 0 in the padding ring (the padding pads the activation, not bn(0)),
// so the BatchNorm apply pass between two convolutions of a BasicBlock chain (model_vec.py:509-593 -> torchvision BasicBlock:
// bn1 -> relu -> conv2, bn2 -> +identity -> relu -> next conv1) is no kernel of its own and its output makes no round trip through
// HBM.  y != NULL: the activation is needed as a tensor as well (the block output: the next block's skip connection); every
// tile then writes the 4x4 interior of its 6x6 patch, which tiles the image exactly.  y == NULL: it is never written; the
// backward recomputes the ReLU sign from x (col_partial_kernel / wino4_outgrad_bn_kernel, zmask).
// Loads are unconditional (coordinates clamped into the image, the padding ring selected to 0 afterwards): a branch around each
// pixel would serialise the 36 (72 with a residual) loads of a thread.  The 16 interior pixels of a patch are always inside.
// HAS_RES / WRITE_Y / RELU are template parameters: a run-time branch inside the unrolled pixel loop - even a wave-uniform one -
// splits it into basic blocks and the loads of a thread are no longer issued back to back.
template <bool HAS_RES, bool WRITE_Y, bool RELU>
__global__ __launch_bounds__(NT) void wino4_input_bn_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ bw, const float* __restrict__ bb,
                                                            float* __restrict__ y, float* __restrict__ V, int B, int H, int W, int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 al, be;
#pragma unroll
    for (int e = 0; e < 4; ++e) { al[e] = mmfn_bn_alpha(bw[c4 + e], rstd[c4 + e]); be[e] = mmfn_bn_beta(bb[c4 + e], mean[c4 + e], al[e]); }
    const size_t img = (size_t)b * H * W;
    f32x4 r[6][6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int xx = 4 * j - 1 + e;
      const bool okx = (e >= 1 && e <= 4) || (unsigned)xx < (unsigned)W;
      const int xc = okx ? xx : (xx < 0 ? 0 : W - 1);
      f32x4 d[6], c[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int yy = 4 * ii - 1 + a;
        const bool oky = (a >= 1 && a <= 4) || (unsigned)yy < (unsigned)H;
        const int yc = oky ? yy : (yy < 0 ? 0 : H - 1);
        const size_t off = (img + (size_t)yc * W + xc) * C + c4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = mmfn_bn_affine(xv[q], al[q], be[q]);
        if (HAS_RES) o += *reinterpret_cast<const f32x4*>(res + off);
        if (RELU) {   // fmaxf, as bn_apply_kernel
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.0f);
        }
        if (a >= 1 && a <= 4 && e >= 1 && e <= 4) {
          if (WRITE_Y) *reinterpret_cast<f32x4*>(y + off) = o;
        } else if (!(okx && oky)) {
          o = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        d[a] = o;
      }
      f4_bt(d, c);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][e] = c[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      f32x4 v[6];
      f4_bt(r[a], v);
#pragma unroll
      for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(V + ((size_t)(a * 6 + e) * T + tile) * C + c4) = v[e];
    }
  }
}

__global__ __launch_bounds__(NT) void wino4_output_kernel(const float* __restrict__ Mt, const float* __restrict__ res,
                                                          float* __restrict__ y, int B, int H, int W, int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 s[4][6];  // A^T m, column by column
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      f32x4 m[6], o[4];
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a] = *reinterpret_cast<const f32x4*>(Mt + ((size_t)(a * 6 + e) * T + tile) * C + c4);
      f4_at(m, o);
#pragma unroll
      for (int p = 0; p < 4; ++p) s[p][e] = o[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 o[4];
      f4_at(s[p], o);
      const size_t off = (((size_t)b * H + 4 * ii + p) * W + 4 * j) * C + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = o[q];
        if (res) v += *reinterpret_cast<const f32x4*>(res + off + (size_t)q * C);
        *reinterpret_cast<f32x4*>(y + off + (size_t)q * C) = v;
      }
    }
  }
}

// All filter transforms of a training step in ONE launch: the filters only change at the optimizer step, so the forward
// transforms every Winograd layer's filter up front (55 layers in the vec model: 55 launches of ~7 us each become one
// ~0.2 ms streaming kernel that runs beside the stem / layer1 convolutions).  table[l] = {w, U, Co, Ci, first element}.
struct WinoGroupEntry { const float* w; float* U; int Co; int Ci; long long start; };

__global__ __launch_bounds__(NT) void wino4_weight_group_kernel(const WinoGroupEntry* __restrict__ table, int n_layers,
                                                                long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_layers - 1;   // last entry with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const WinoGroupEntry e = table[lo];
    const long long j = i - e.start;
    const int Co = e.Co, Ci = e.Ci;
    const int co = (int)(j / Ci), ci = (int)(j % Ci);
    float t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float g[3], u[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) g[a] = e.w[(((size_t)co * 3 + a) * 3 + b) * Ci + ci];
      f4_g(g, u);
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a][b] = u[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float u[6];
      f4_g(t[a], u);
#pragma unroll
      for (int b = 0; b < 6; ++b) e.U[((size_t)(a * 6 + b) * Co + co) * Ci + ci] = u[b];
    }
  }
}

// ---- weight gradient in the F(4x4,3x3) domain:  dw = G^T [ sum_tiles (A dY A^T) . (B^T x B) ] G
// dMt[t][tile][c] = (A dy A^T)[t] for the 4x4 output-gradient patch of the tile (A = transpose of A^T above, 6x4)
template <typename T>
__device__ __forceinline__ void f4_a(const T* y, T* m) {  // m = A y, 4 -> 6
  const T ea = y[0] + WA2 * y[2], oa = WA * y[1] + WA3 * y[3];
  const T eb = y[0] + WB2 * y[2], ob = WB * y[1] + WB3 * y[3];
  m[0] = y[0];
  m[1] = ea + oa;
  m[2] = ea - oa;
  m[3] = eb + ob;
  m[4] = eb - ob;
  m[5] = y[3];
}
__device__ __forceinline__ void f4_gt(const float* u, float* g) {  // g = G^T u, 6 -> 3
  const float pa = WFA * (u[1] + u[2]), qa = WFA * (u[1] - u[2]), pb = WFB * (u[3] + u[4]), qb = WFB * (u[3] - u[4]);
  g[0] = (1.0f / WK0) * u[0] + pa + pb;
  g[1] = WA * qa + WB * qb;
  g[2] = WA2 * pa + WB2 * pb + u[5];
}

__global__ __launch_bounds__(NT) void wino4_outgrad_kernel(const float* __restrict__ dy, float* __restrict__ dMt, int B, int H, int W,
                                                           int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 r[6][4];  // A dy, column by column
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 y[4], m[6];
#pragma unroll
      for (int p = 0; p < 4; ++p) y[p] = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * H + 4 * ii + p) * W + 4 * j + q) * C + c4);
      f4_a(y, m);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][q] = m[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      f32x4 m[6];
      f4_a(r[a], m);
#pragma unroll
      for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(dMt + ((size_t)(a * 6 + e) * T + tile) * C + c4) = m[e];
    }
  }
}

// The same transform with the BatchNorm backward applied on the fly: the layer's output gradient
//   dy = w * rstd * (ge - mean(ge) - xhat * mean(ge * xhat)),  ge = g * (y > 0),  xhat = (x - mean) * rstd
// (bn_bwd_apply_kernel, norm.hip) is only ever consumed in the Winograd domain by these layers, so it is formed in registers
// from g, y and the convolution output x instead of being written to HBM by one pass and read back by the next
// (one launch and 2 x |dy| bytes less per convolution).  means = [2][C]: mean(ge), mean(ge * xhat) from the reduction
// kernels.  ge_out (optional) receives ge, the gradient of the residual branch.
// MASK: 0 no ReLU, 1 mask from y, 2 mask recomputed from x (zb).  Template parameters, not run-time tests: see wino4_input_bn_kernel.
template <int MASK, bool HAS_GE>
__global__ __launch_bounds__(NT) void wino4_outgrad_bn_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                              const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ w,
                                                              const float* __restrict__ zb, const float* __restrict__ means,
                                                              float* __restrict__ ge_out, float* __restrict__ dMt, int B, int H, int W,
                                                              int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  constexpr bool zmask = MASK == 2;   // the ReLU output was never written: its sign is recomputed from x (norm.hip col_partial_kernel)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t tile = i / cq;
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c4);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(means + c4), m2 = *reinterpret_cast<const f32x4*>(means + C + c4);
    f32x4 al = {0.f, 0.f, 0.f, 0.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (zmask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { al[e] = mmfn_bn_alpha(wv[e], rs[e]); be[e] = mmfn_bn_beta(zb[c4 + e], mu[e], al[e]); }
    }
    f32x4 r[6][4];  // A dy, column by column
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 d[4], m[6];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t off = (((size_t)b * H + 4 * ii + p) * W + 4 * j + q) * C + c4;
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        if (MASK == 1) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(y + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.0f ? gv[e] : 0.0f;
        } else if (MASK == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[e] = mmfn_bn_affine(xv[e], al[e], be[e]) > 0.0f ? gv[e] : 0.0f;
        }
        if (HAS_GE) *reinterpret_cast<f32x4*>(ge_out + off) = gv;
        {
          // no floating-point contraction in here: the product below feeds the sums of f4_a, and whether the compiler fuses
          // it into them has differed between the MASK / HAS_GE instantiations (1 ulp in 0.2 % of dM) - the mask recomputed
          // from x must give the bits of the mask read from y
#pragma clang fp contract(off)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            d[p][e] = (gv[e] - m1[e] - xh * m2[e]) * (wv[e] * rs[e]);
          }
        }
      }
      f4_a(d, m);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][q] = m[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      f32x4 m[6];
      f4_a(r[a], m);
#pragma unroll
      for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(dMt + ((size_t)(a * 6 + e) * T + tile) * C + c4) = m[e];
    }
  }
}

// ---- data gradient in the F(4x4,3x3) domain: the ADJOINT of the forward pipeline instead of a second convolution with
// the flipped filter.  With dM = A dY A^T (wino4_outgrad_kernel - the weight gradient needs it anyway) and the forward's
// own U = G w G^T:   dV[t] = dM[t] . U[t]   ([tiles x Co] x [Co x Ci], one 36-batch GEMM),   dx = sum over tiles of
// B dV_tile B^T put back onto the tile's 6x6 input patch.  Saves, per convolution, the filter flip, the second filter
// transform and the input transform of dY.  Patches of neighbouring tiles overlap by two pixels, so this is an overlap-ADD,
// done as a gather (below): no atomics, fixed order.
template <typename T>
__device__ __forceinline__ void f4_b(const T* v, T* o) {  // o = B v, 6 -> 6 (B = transpose of the B^T in f4_bt)
  o[0] = WK0 * v[0];
  o[1] = (WA * WB2) * (v[2] - v[1]) + (WA2 * WB) * (v[4] - v[3]) + WK0 * v[5];
  o[2] = -WS2 * v[0] - WB2 * (v[1] + v[2]) - WA2 * (v[3] + v[4]);
  o[3] = WA * (v[1] - v[2]) + WB * (v[3] - v[4]) - WS2 * v[5];
  o[4] = v[0] + v[1] + v[2] + v[3] + v[4];
  o[5] = v[5];
}

// Gather form, no LDS and no phases.  A thread owns one 4x4 output block (= the interior of its own tile's 6x6 patch) and adds
// what the eight neighbouring tiles' patches put on it.  Because rows 0 and 5 of B are a^2b^2*e0 and e5 (f4_b: o[0] = WK0 v[0],
// o[5] = v[5]), a neighbour's halo row / column needs only ONE row / column of its dV: 36 + 4*6 + 4 = 64 loads per thread
// instead of 9*36, every addition in a fixed order.
// The 4x4 block of dx a thread owns: P[a][e], rows 4*ti + a, columns 4*tj + e, four channels from c4 (res not yet added).
__device__ __forceinline__ void wino4_adjoint_block(const float* __restrict__ dV, int64_t T, int C, int th, int tw, int64_t tile, int c4,
                                                    f32x4 (*P)[4]) {
    const int tj = (int)(tile % tw), ti = (int)((tile / tw) % th);
    auto at = [&](int64_t tl, int a, int e) { return *reinterpret_cast<const f32x4*>(dV + ((size_t)(a * 6 + e) * T + tl) * C + c4); };
    f32x4 t[4][6];  // rows 1..4 of B dV, built column by column
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      f32x4 v[6], o[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) v[a] = at(tile, a, e);
      f4_b(v, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) t[a][e] = o[a + 1];
    }
    // interior of B dV B^T
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 o[6];
      f4_b(t[a], o);
#pragma unroll
      for (int e = 0; e < 4; ++e) P[a][e] = o[e + 1];
    }
    const bool up = ti > 0, down = ti < th - 1, left = tj > 0, right = tj < tw - 1;
    if (up) {      // tile above: its patch row 5 lies on our first row
      f32x4 v[6], o[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) v[e] = at(tile - tw, 5, e);
      f4_b(v, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) P[0][e] += o[e + 1];
    }
    if (down) {    // tile below: its patch row 0 (= a^2b^2 * dV row 0) lies on our last row
      f32x4 v[6], o[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) v[e] = at(tile + tw, 0, e);
      f4_b(v, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) P[3][e] += WK0 * o[e + 1];
    }
    if (left) {
      f32x4 v[6], o[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) v[a] = at(tile - 1, a, 5);
      f4_b(v, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a][0] += o[a + 1];
    }
    if (right) {
      f32x4 v[6], o[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) v[a] = at(tile + 1, a, 0);
      f4_b(v, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a][3] += WK0 * o[a + 1];
    }
    if (up && left) P[0][0] += at(tile - tw - 1, 5, 5);
    if (up && right) P[0][3] += WK0 * at(tile - tw + 1, 5, 0);
    if (down && left) P[3][0] += WK0 * at(tile + tw - 1, 0, 5);
    if (down && right) P[3][3] += (WK0 * WK0) * at(tile + tw + 1, 0, 0);
}

template <bool HAS_RES>
__global__ __launch_bounds__(NT) void wino4_input_adjoint_kernel(const float* __restrict__ dV, const float* __restrict__ res,
                                                                 float* __restrict__ dx, int B, int H, int W, int C) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw, n = T * cq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % cq) * 4;
    const int64_t tile = idx / cq;
    const int tj = (int)(tile % tw), ti = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 P[4][4];
    wino4_adjoint_block(dV, T, C, th, tw, tile, c4, P);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const size_t off = (((size_t)b * H + 4 * ti + a) * W + 4 * tj) * C + c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 v = P[a][e];
        if (HAS_RES) v += *reinterpret_cast<const f32x4*>(res + off + (size_t)e * C);
        *reinterpret_cast<f32x4*>(dx + off + (size_t)e * C) = v;
      }
    }
  }
}

// dw[co][3][3][ci] = G^T dU[:, co, ci] G;  dU is [36][Co][Ci]
__global__ __launch_bounds__(NT) void wino4_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Co, int Ci) {
  const int64_t n = (int64_t)Co * Ci;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / Ci), ci = (int)(i % Ci);
    float t[3][6];  // G^T dU
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      float u[6], g[3];
#pragma unroll
      for (int a = 0; a < 6; ++a) u[a] = dU[((size_t)(a * 6 + e) * Co + co) * Ci + ci];
      f4_gt(u, g);
#pragma unroll
      for (int a = 0; a < 3; ++a) t[a][e] = g[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float g[3];
      f4_gt(t[a], g);
#pragma unroll
      for (int b = 0; b < 3; ++b) dw[(((size_t)co * 3 + a) * 3 + b) * Ci + ci] = g[b];
    }
  }
}

// The same over the un-combined slabs [splits][36][Co][Ci] of a split-K batched GEMM (MMFN_EPI_KEEP_SLABS): a block owns 64
// consecutive ci of one co; all 256 threads first sum the slabs of its 36 x 64 frequency values in slice order (coalesced 256-byte
// rows, two partial sums so the loads pipeline) into LDS, then 64 threads transform.
__global__ __launch_bounds__(NT) void wino4_wgrad_out_slabs_kernel(const float* __restrict__ slabs, int splits, float* __restrict__ dw,
                                                                   int Co, int Ci) {
  __shared__ float du[36][64];
  const int blocks_ci = Ci >> 6;
  const int co = blockIdx.x / blocks_ci, ci0 = (blockIdx.x - co * blocks_ci) << 6;
  const int j = threadIdx.x & 63, fg = threadIdx.x >> 6;
  const size_t slab = (size_t)36 * Co * Ci, plane = (size_t)Co * Ci;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int f = fg + 4 * q;
    const float* p = slabs + (size_t)f * plane + (size_t)co * Ci + ci0 + j;
    float s0 = 0.f, s1 = 0.f;
    int z = 0;
    for (; z + 1 < splits; z += 2) {
      s0 += p[(size_t)z * slab];
      s1 += p[(size_t)(z + 1) * slab];
    }
    if (z < splits) s0 += p[(size_t)z * slab];
    du[f][j] = s0 + s1;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int ci = ci0 + j;
  float t[3][6];
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    float u[6], g[3];
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = du[a * 6 + e][j];
    f4_gt(u, g);
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a][e] = g[a];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float g[3];
    f4_gt(t[a], g);
#pragma unroll
    for (int b = 0; b < 3; ++b) dw[(((size_t)co * 3 + a) * 3 + b) * Ci + ci] = g[b];
  }
}

// Output transform that also emits the BatchNorm batch statistics of what it writes: block b owns a contiguous run of
// tiles, thread (tile lane, channel quad) accumulates sum / sum of squares of its 16 pixels x 4 channels in fp64, the tile
// lanes are combined through LDS and row b of partials [gridDim][2][C] is written (same format as col_partial_kernel<0>,
// finished by mmfn_bn_finalize_stats_f32).  Saves the separate statistics pass over the convolution output.
__global__ __launch_bounds__(NT) void wino4_output_stats_kernel(const float* __restrict__ Mt, float* __restrict__ y,
                                                                double* __restrict__ partials, int B, int H, int W, int C,
                                                                int tiles_per_block) {
  const int cq = C >> 2, th = H >> 2, tw = W >> 2;
  const int64_t T = (int64_t)B * th * tw;
  const int TL = NT / cq;  // tile lanes per block
  const int cqi = threadIdx.x % cq, tl = threadIdx.x / cq;
  const int c4 = cqi * 4;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = (t0 + tiles_per_block < T) ? t0 + tiles_per_block : T;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int64_t tile = t0 + tl; tile < t1; tile += TL) {
    const int j = (int)(tile % tw), ii = (int)((tile / tw) % th), b = (int)(tile / ((int64_t)tw * th));
    f32x4 s[4][6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      f32x4 m[6], o[4];
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a] = *reinterpret_cast<const f32x4*>(Mt + ((size_t)(a * 6 + e) * T + tile) * C + c4);
      f4_at(m, o);
#pragma unroll
      for (int p = 0; p < 4; ++p) s[p][e] = o[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 o[4];
      f4_at(s[p], o);
      const size_t off = (((size_t)b * H + 4 * ii + p) * W + 4 * j) * C + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4*>(y + off + (size_t)q * C) = o[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double v = o[q][e]; s1[e] += v; s2[e] += v * v; }
      }
    }
  }
  __shared__ double red[2][NT][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  if (tl == 0) {
    for (int k = 1; k < TL; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += red[0][k * cq + cqi][e]; s2[e] += red[1][k * cq + cqi][e]; }
    double* p = partials + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[c4 + e] = s1[e]; p[C + c4 + e] = s2[e]; }
  }
}

inline int grid_for(int64_t n) { return (int)std::min<int64_t>((n + NT - 1) / NT, 65535 * 4); }

}  // namespace

extern "C" int mmfn_wino_weight_f32(const float* w, float* U, int Co, int Ci, int m, void* stream) {
  if (!w || !U || Co <= 0 || Ci <= 0 || (m != 2 && m != 4)) return MMFN_EINVAL;
  if (m == 4) {
    hipLaunchKernelGGL(wino4_weight_kernel, dim3(grid_for((int64_t)Co * Ci)), dim3(NT), 0, (hipStream_t)stream, w, U, Co, Ci);
    MMFN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(wino_weight_kernel, dim3(grid_for((int64_t)Co * Ci)), dim3(NT), 0, (hipStream_t)stream, w, U, Co, Ci);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_weight_group_f32(const void* table, int n_layers, int64_t total, void* stream) {
  if (!table || n_layers <= 0 || total <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino4_weight_group_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                     (const WinoGroupEntry*)table, n_layers, (long long)total);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_input_f32(const float* x, float* V, int B, int H, int W, int C, int m, void* stream) {
  if (!x || !V || (m != 2 && m != 4) || (H % m) || (W % m) || (C & 3) || B <= 0) return MMFN_EINVAL;
  if (m == 4) {
    hipLaunchKernelGGL(wino4_input_kernel, dim3(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, x,
                       V, B, H, W, C);
    MMFN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, x, V,
                     B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_input_bn_f32(const float* x, const float* res, const float* mean, const float* rstd, const float* weight,
                                      const float* bias, int relu, float* y, float* V, int B, int H, int W, int C, void* stream) {
  if (!x || !V || !mean || !rstd || !weight || !bias || (H & 3) || (W & 3) || (C & 3) || B <= 0) return MMFN_EINVAL;
  const dim3 grid(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4)));
#define MMFN_WINO_IN_BN(R, Y, L)                                                                                                  \
  hipLaunchKernelGGL((wino4_input_bn_kernel<R, Y, L>), grid, dim3(NT), 0, (hipStream_t)stream, x, res, mean, rstd, weight, bias, y, V, B, \
                     H, W, C)
#define MMFN_WINO_IN_BN2(R, Y) do { if (relu) MMFN_WINO_IN_BN(R, Y, true); else MMFN_WINO_IN_BN(R, Y, false); } while (0)
  if (res) { if (y) MMFN_WINO_IN_BN2(true, true); else MMFN_WINO_IN_BN2(true, false); }
  else     { if (y) MMFN_WINO_IN_BN2(false, true); else MMFN_WINO_IN_BN2(false, false); }
#undef MMFN_WINO_IN_BN2
#undef MMFN_WINO_IN_BN
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_output_f32(const float* Mt, const float* res, float* y, int B, int H, int W, int C, int m,
                                    void* stream) {
  if (!Mt || !y || (m != 2 && m != 4) || (H % m) || (W % m) || (C & 3) || B <= 0) return MMFN_EINVAL;
  if (m == 4) {
    hipLaunchKernelGGL(wino4_output_kernel, dim3(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4))), dim3(NT), 0, (hipStream_t)stream,
                       Mt, res, y, B, H, W, C);
    MMFN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, Mt,
                     res, y, B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_outgrad_f32(const float* dy, float* dMt, int B, int H, int W, int C, void* stream) {
  if (!dy || !dMt || (H & 3) || (W & 3) || (C & 3) || B <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino4_outgrad_kernel, dim3(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4))), dim3(NT), 0, (hipStream_t)stream, dy,
                     dMt, B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_outgrad_bn_f32(const float* g, const float* y, const float* x, const float* mean, const float* rstd,
                                        const float* weight, const float* relu_bias, const float* means, float* ge_out, float* dMt,
                                        int B, int H, int W, int C, void* stream) {
  if (!g || !x || !mean || !rstd || !weight || !means || !dMt || (H & 3) || (W & 3) || (C & 3) || B <= 0) return MMFN_EINVAL;
  const dim3 grid(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4)));
#define MMFN_OG(M, G)                                                                                                              \
  hipLaunchKernelGGL((wino4_outgrad_bn_kernel<M, G>), grid, dim3(NT), 0, (hipStream_t)stream, g, y, x, mean, rstd, weight, relu_bias, \
                     means, ge_out, dMt, B, H, W, C)
#define MMFN_OG2(M) do { if (ge_out) MMFN_OG(M, true); else MMFN_OG(M, false); } while (0)
  if (y) MMFN_OG2(1); else if (relu_bias) MMFN_OG2(2); else MMFN_OG2(0);
#undef MMFN_OG2
#undef MMFN_OG
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_input_adjoint_f32(const float* dV, const float* res, float* dx, int B, int H, int W, int C, void* stream) {
  if (!dV || !dx || (H & 3) || (W & 3) || (C & 3) || B <= 0) return MMFN_EINVAL;
  const dim3 grid(grid_for((int64_t)B * (H / 4) * (W / 4) * (C / 4)));
  if (res) hipLaunchKernelGGL(wino4_input_adjoint_kernel<true>, grid, dim3(NT), 0, (hipStream_t)stream, dV, res, dx, B, H, W, C);
  else hipLaunchKernelGGL(wino4_input_adjoint_kernel<false>, grid, dim3(NT), 0, (hipStream_t)stream, dV, res, dx, B, H, W, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_wgrad_out_f32(const float* dU, float* dw, int Co, int Ci, void* stream) {
  if (!dU || !dw || Co <= 0 || Ci <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino4_wgrad_out_kernel, dim3(grid_for((int64_t)Co * Ci)), dim3(NT), 0, (hipStream_t)stream, dU, dw, Co, Ci);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_wgrad_out_slabs_f32(const float* slabs, int splits, float* dw, int Co, int Ci, void* stream) {
  if (!slabs || !dw || splits < 1 || Co <= 0 || Ci <= 0 || (Ci & 63)) return MMFN_EINVAL;
  hipLaunchKernelGGL(wino4_wgrad_out_slabs_kernel, dim3(Co * (Ci >> 6)), dim3(NT), 0, (hipStream_t)stream, slabs, splits, dw, Co, Ci);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_wino_output_stats_f32(const float* Mt, float* y, double* partials, int* nblk_out, int B, int H, int W, int C,
                                          void* stream) {
  if (!Mt || !y || !partials || !nblk_out || (H & 3) || (W & 3) || (C & 3) || B <= 0) return MMFN_EINVAL;
  const int cq = C / 4;
  if (cq > NT || NT % cq) return MMFN_EINVAL;
  const int64_t T = (int64_t)B * (H / 4) * (W / 4);
  const int TL = NT / cq;
  const int tpb = (int)std::max<int64_t>(TL, (T + 511) / 512);
  const int nblk = (int)((T + tpb - 1) / tpb);
  *nblk_out = nblk;
  hipLaunchKernelGGL(wino4_output_stats_kernel, dim3(nblk), dim3(NT), 0, (hipStream_t)stream, Mt, y, partials, B, H, W, C, tpb);
  MMFN_LAUNCH_CHECK();
  return 0;
}
