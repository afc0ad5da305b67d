// Shared pieces of the fused GPT-block kernels (gpt_block.hip: the row-block kernels; attention_wg.hip: the per-(sample, head, half)
// kernel with the projection prologue).  model_vec.py:112-133 (Block), :73-109 (SelfAttention), :120-125 (mlp).
#pragma once
#include "common.h"

typedef mmfn_gpt_block_desc GptArgs;

#ifdef __HIPCC__
__device__ __forceinline__ f32x4 gpt_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void gpt_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 gpt_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[i][j] += W[16 i + .][:] . A[16 j + .][:]^T over K columns ("NT": both operands contract along their rows' columns).
//   W: global (L2-resident weights), row pitch ldw floats, pointing at the first of this wave's NWT 16-row tiles;
//   sA: LDS activations, row pitch K + 4 floats, pointing at the first of this wave's NTT 16-row tiles.
// MFMA roles: A operand = weight rows (accumulator rows 4*l4 + r = output column n), B operand = activation rows (accumulator
// column l15 = token), so a lane ends up with 4 consecutive output columns of one token: 16-byte stores to LDS / HBM.
// Lane (l15, l4) reads the float4 at columns 16c + 4*l4 of its row: element e feeds MFMA step e, whose four k slots are the
// columns {16c + 4*slot + e} - the same permutation of the chunk on both operands (attention_wg.hip product_phase).
// The weight fragments run D chunks ahead of the MFMAs in a register ring (L2 latency ~ 1-2 chunks of MFMA time).
template <int K, int NWT, int NTT, int D>
__device__ __forceinline__ void gpt_rows_gemm_nt(const float* __restrict__ W, int ldw, const float* sA, int l15, int l4,
                                                 f32x4 (*acc)[NTT]) {
  constexpr int NCH = K / 16, P = K + 4;
  f32x4 wf[D][NWT];
  const float* wp = W + (size_t)l15 * ldw + 4 * l4;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < NCH) {
#pragma unroll
      for (int i = 0; i < NWT; ++i) wf[d][i] = gpt_ld4(wp + (size_t)(16 * i) * ldw + 16 * d);
    }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 af[NTT], wv[NWT];
#pragma unroll
    for (int j = 0; j < NTT; ++j) af[j] = *reinterpret_cast<const f32x4*>(sA + (16 * j + l15) * P + 16 * c + 4 * l4);
#pragma unroll
    for (int i = 0; i < NWT; ++i) wv[i] = wf[c % D][i];
    if (c + D < NCH) {
#pragma unroll
      for (int i = 0; i < NWT; ++i) wf[c % D][i] = gpt_ld4(wp + (size_t)(16 * i) * ldw + 16 * (c + D));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < NWT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j) acc[i][j] = gpt_mfma16(wv[i][e], af[j][e], acc[i][j]);
  }
}

// acc[i][j] += sum_n W[n][16 i + .] * A[16 j + .][n] over N rows of W ("NN": the data gradient dx = g . W, W [N][Kout]).
//   W: global, row pitch ldw, pointing at column 16 * (first tile) of row 0;  sA: LDS [rows][N + 4] gradient rows.
// The weight fragment of a lane is column 16 i + l15 of rows {16c + 4*l4 + e}: four 4-byte loads, 64 contiguous bytes per
// 16 lanes.
template <int N, int NWT, int NTT, int D>
__device__ __forceinline__ void gpt_rows_gemm_nn(const float* __restrict__ W, int ldw, const float* sA, int l15, int l4,
                                                 f32x4 (*acc)[NTT]) {
  constexpr int NCH = N / 16, P = N + 4;
  f32x4 wf[D][NWT];
  const float* wp = W + (size_t)(4 * l4) * ldw + l15;
  auto fetch = [&](int c, f32x4* dst) {
#pragma unroll
    for (int i = 0; i < NWT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[i][e] = wp[(size_t)(16 * c + e) * ldw + 16 * i];
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < NCH) fetch(d, wf[d]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 af[NTT], wv[NWT];
#pragma unroll
    for (int j = 0; j < NTT; ++j) af[j] = *reinterpret_cast<const f32x4*>(sA + (16 * j + l15) * P + 16 * c + 4 * l4);
#pragma unroll
    for (int i = 0; i < NWT; ++i) wv[i] = wf[c % D][i];
    if (c + D < NCH) fetch(c + D, wf[c % D]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < NWT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j) acc[i][j] = gpt_mfma16(wv[i][e], af[j][e], acc[i][j]);
  }
}

// sum over the 16 lanes that share l4 (lanes l15 = 0..15 of a row group)
__device__ __forceinline__ float gpt_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v + __shfl_xor(v, 8, 64);
}
#endif
