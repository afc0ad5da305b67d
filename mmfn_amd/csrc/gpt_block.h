// Shared pieces of the fused GPT-block kernels (gpt_block.hip: the row-block kernels; attention_wg.hip: the per-(sample, head, half)
// kernel with the projection prologue).  model_vec.py:112-133 (Block), :73-109 (SelfAttention), :120-125 (mlp).
#pragma once
#include "common.h"

typedef mmfn_gpt_block_desc GptArgs;

#ifdef __HIPCC__
__device__ __forceinline__ f32x4 gpt_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void gpt_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 gpt_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() is a workgroup-scope release fence + s_barrier, and
// the fence drains the wave's global STORES too (s_waitcnt vmcnt(0)): every phase of the fused kernels ends in an epilogue that
// stores a saved tensor (x1, a2, h, gh, ...), and waiting for those stores to be acknowledged cost 2-4 us per phase (s_memtime
// stamps, tools/experiments/gpt_phases.sh: 13 of the 31 us of gpt_mlp_fwd_kernel<128>).  Nothing behind these barriers reads what
// the workgroup stored to HBM; loads feeding an LDS write are ordered by their data dependency.
__device__ __forceinline__ void gpt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Precision of the GEMM operands.  fp32 path: everything fp32, v_mfma_f32_16x16x4_f32, 16 contraction indices per chunk.  bf16 mode
// (GlobalConfig(act_dtype="bf16")): activations / activation gradients that are GEMM operands and the weight shadows are bf16 in HBM
// and LDS, v_mfma_f32_16x16x32_bf16, 32 indices per chunk; accumulation, LayerNorm, the residual stream and its gradient stay fp32.
typedef __bf16 gpt_bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF> struct GptPrec;
template <> struct GptPrec<false> { typedef float act; typedef f32x4 frag; static constexpr int KCH = 16, EPU = 4; };
template <> struct GptPrec<true> { typedef bf16_t act; typedef gpt_bf16x8 frag; static constexpr int KCH = 32, EPU = 8; };
// (EPU = elements per 16-byte unit = the padding of an LDS operand row: row pitch K + EPU elements, so the 16-byte fragment reads
// of 16 consecutive rows fall into distinct bank quads)

// Weight fragments of a row-block product, D chunks (of KCH contraction indices) ahead of the MFMAs in a register ring.  start()
// is called BEFORE the barrier / epilogue / LayerNorm phase that precedes the product, so the first chunks' L2 latency (1-2 us on
// a cold matrix) is not exposed at every phase boundary of the fused kernels.
//   NN = false ("NT"): out[t][n] = sum_k A[t][k] W[n][k]  (forward Linear; W [N][K], row pitch ldw, this wave's first row)
//        lane (l15, l4) reads the 16 bytes at columns KCH*c + EPU*l4 of weight row 16 i + l15.  fp32: element e feeds MFMA step e,
//        whose four k slots are the columns {16c + 4*slot + e} - the same permutation of the chunk on both operands
//        (attention_wg.hip); bf16: the 8 values ARE the lane's k slots of one 16x16x32 MFMA.
//   NN = true (fp32 only): out[t][k] = sum_n A[t][n] W[n][k]  (data gradient dx = g . W; W [N][Kout], this wave's first COLUMN)
//        the lane's fragment is column 16 i + l15 of rows {16c + 4*l4 + e}: four 4-byte loads, 64 contiguous bytes per 16 lanes.
//        (The bf16 mode has transposed weight shadows, so its data gradients are NT products too.)
// MFMA roles: A operand = weights (accumulator rows 4*l4 + r = output column), B operand = activation rows from LDS (accumulator
// column l15 = token), so a lane ends up with 4 consecutive output columns of one token: 16- / 8-byte stores to LDS / HBM.
template <int K, int NWT, int D, bool NN, bool BF = false>
struct GptWRing {
  typedef GptPrec<BF> PR;
  typedef typename PR::act elem;
  typedef typename PR::frag frag;
  static_assert(!(NN && BF), "bf16 data gradients contract over the transposed shadows (NT)");
  static constexpr int NCH = K / PR::KCH;
  frag wf[D][NWT];
  const elem* wp;
  int ldw;
  __device__ __forceinline__ void fetch(int c, frag* dst) const {
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
      if constexpr (NN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[i][e] = wp[(size_t)(16 * c + e) * ldw + 16 * i];
      } else {
        dst[i] = *reinterpret_cast<const frag*>(wp + (size_t)(16 * i) * ldw + PR::KCH * c);
      }
    }
  }
  __device__ __forceinline__ void start(const elem* __restrict__ W, int ldw_, int l15, int l4) {
    ldw = ldw_;
    wp = NN ? W + (size_t)(4 * l4) * ldw + l15 : W + (size_t)l15 * ldw + PR::EPU * l4;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < NCH) fetch(d, wf[d]);
  }
  // acc[i][j] += (weight tile i) x (activation tile j); sA: LDS rows of this wave's first 16-row tile, row pitch K + EPU elements
  template <int NTT>
  __device__ __forceinline__ void run(const elem* sA, int l15, int l4, f32x4 (*acc)[NTT]) {
    constexpr int P = K + PR::EPU;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      frag af[NTT], wv[NWT];
#pragma unroll
      for (int j = 0; j < NTT; ++j) af[j] = *reinterpret_cast<const frag*>(sA + (16 * j + l15) * P + PR::KCH * c + PR::EPU * l4);
#pragma unroll
      for (int i = 0; i < NWT; ++i) wv[i] = wf[c % D][i];
      if (c + D < NCH) fetch(c + D, wf[c % D]);
      if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < NWT; ++i)
#pragma unroll
          for (int j = 0; j < NTT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[i], af[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < NWT; ++i)
#pragma unroll
            for (int j = 0; j < NTT; ++j) acc[i][j] = gpt_mfma16(wv[i][e], af[j][e], acc[i][j]);
      }
    }
  }
};

// sum over the 16 lanes that share l4 (lanes l15 = 0..15 of a row group)
__device__ __forceinline__ float gpt_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v + __shfl_xor(v, 8, 64);
}
#endif
