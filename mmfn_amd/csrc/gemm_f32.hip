// fp32 GEMM / implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32.
//
// One kernel template covers the six contraction forms of the MMFN training step
// (SURVEY.md section 2.2 K3,K5,K9,K11 and their backward twins):
//   Linear fwd      C[M,N]  = A[M,K]          * W[N,K]^T          (A_ROWMAJOR, B_NK)
//   Linear dX       dX[M,K] = dY[M,N]         * W[N,K]            (A_ROWMAJOR, B_KN)
//   Linear dW       dW[N,K] = dY[M,N]^T       * X[M,K]            (A_COLMAJOR, B_KN)
//   conv fwd        Y[M,Co] = im2col(X)       * W[Co,KhKwCi]^T    (A_IM2COL,   B_NK)
//   conv dgrad      dX      = gatherT(dY)     * W as [(kh,kw,co), ci]   (A_DGRAD, B_DGRADW)
//   conv wgrad      dW[Co,KhKwCi] = dY^T      * im2col(X)         (A_COLMAJOR, B_IM2COL)
// Feature maps are NHWC, conv weights are stored [Cout][KH][KW][Cin], so every form reads
// 16-byte vectors along the contiguous axis.
//
// Tiling: BMxBN block tile (128x128 or 64x64), BK = 16, 4 waves as 2x2, each wave a grid of
// 32x32 MFMA tiles.  Operands that are contiguous along k are staged in LDS as [rows][BK+4]
// (stride 20 dwords: the 16-lane groups of ds_read_b128 land on 16 distinct 4-bank slots) and
// fetched as one float4 per 8-k chunk; the MFMA k-order is permuted so that lane-half h consumes
// k = 8c+4h+j in step j — A and B use the same permutation, so the sum is unchanged.  Operands
// contiguous along m/n are staged as [BK][rows] and read conflict-free with ds_read_b32.
// Global->register->LDS staging is double buffered: the loads of tile t+1 are in flight while
// tile t runs on the matrix pipe (fp32 MFMA issues every 64 cycles, so one barrier per 16-k
// tile is far off the critical path).
#include <algorithm>
#include <cstdlib>

#include <stdlib.h>

#include "common.h"

namespace {

#ifndef MMFN_GEMM_BK
#define MMFN_GEMM_BK 16
#endif
constexpr int BK = MMFN_GEMM_BK;
constexpr int KQ = BK / 4;  // float4 slots per staged row
constexpr int NT = 256;

struct ConvPos {  // decoded position of a GEMM row in conv space
  int b, y0, x0;
  bool ok;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// LDS image of a k-contiguous operand tile: row r holds BK floats = KQ 16-byte slots, slot q stored at
// q ^ ((r / (64/BK)) & (KQ-1)).  ds_write_b128 (8-lane groups = 2 rows x 4 slots... ) and the MFMA
// fragment ds_read_b128 (16-lane groups) are both bank-conflict free with no padding.
__device__ __forceinline__ int kc_slot(int row, int q) { return q ^ ((row / (64 / BK)) & (KQ - 1)); }

__device__ __forceinline__ void epilogue_store(const mmfn_gemm_desc& d, uint64_t key, int row, int col, float v) {
  const int f = d.flags;
  if (f & MMFN_EPI_BIAS) v += d.bias[col];
  if (f & MMFN_EPI_RELU) v = fmaxf(v, 0.0f);
  if (f & MMFN_EPI_GELU) v = mmfn_gelu(v);
  if (f & MMFN_EPI_MASK_AUX) v = d.aux[(size_t)row * d.ldaux + col] > 0.0f ? v : 0.0f;
  if (f & MMFN_EPI_DROPOUT)
    v *= mmfn_dropout_scale(key, (uint64_t)row * (uint64_t)d.N + (uint64_t)col, d.drop_p, 1.0f / (1.0f - d.drop_p));
  if (f & MMFN_EPI_RESIDUAL) v += d.res[(size_t)row * d.ldr + col];
  float* c = d.C + (size_t)row * d.ldc + col;
  if (f & MMFN_EPI_ACCUM) v += *c;
  if (f & MMFN_EPI_RELU_LAST) v = fmaxf(v, 0.0f);
  *c = v;
}

__device__ __forceinline__ mmfn_gemm_desc batch_view(const mmfn_gemm_desc& in, int zb = -1) {
  mmfn_gemm_desc d = in;
  if (in.batch > 1) {
    const size_t z = zb >= 0 ? (size_t)zb : (size_t)blockIdx.z;
    d.A += z * in.strideA;
    d.B += z * in.strideB;
    d.C += z * in.strideC;
    if (d.res) d.res += z * in.strideC;
    if (d.aux) d.aux += z * in.strideC;
  }
  return d;
}

template <int AM, int BMODE, int BM, int BN>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(const mmfn_gemm_desc d_in, const int kt_per_split,
                                                      const int tiles_n) {
  const mmfn_gemm_desc d = batch_view(d_in);
  constexpr bool A_KC = (AM != MMFN_A_COLMAJOR);
  constexpr bool B_KC = (BMODE == MMFN_B_NK);
  constexpr int WAVES_M = 2, WAVES_N = 2;
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int LDK = BK;  // unpadded rows; 16-byte slots are XOR-swizzled by row (see kc_slot)
  constexpr int A_ELEMS = A_KC ? BM * LDK : BK * BM;
  constexpr int B_ELEMS = B_KC ? BN * LDK : BK * BN;
  constexpr int UA = BM * BK / 4 / NT, UB = BN * BK / 4 / NT;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int bid = blockIdx.x;
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int nkt = (d.K + BK - 1) / BK;
  const int kt_begin = blockIdx.y * kt_per_split;
  const int kt_end = min(nkt, kt_begin + kt_per_split);

  const int KHW = d.KH * d.KW;
  // ---------------- per-thread loader state ----------------
  ConvPos apos[UA];
  bool a_vec;
  if (AM == MMFN_A_ROWMAJOR) a_vec = ((d.lda & 3) == 0) && ((((uintptr_t)d.A) & 15) == 0);
  else if (AM == MMFN_A_COLMAJOR) a_vec = ((d.lda & 3) == 0) && ((((uintptr_t)d.A) & 15) == 0);
  else if (AM == MMFN_A_IM2COL) a_vec = ((d.Cin & 3) == 0) && ((((uintptr_t)d.A) & 15) == 0);
  else a_vec = ((d.Cout & 3) == 0) && ((((uintptr_t)d.A) & 15) == 0);
#pragma unroll
  for (int i = 0; i < UA; ++i) {
    apos[i].ok = false; apos[i].b = 0; apos[i].y0 = 0; apos[i].x0 = 0;
    if (AM == MMFN_A_IM2COL || AM == MMFN_A_DGRAD) {
      const int m = m0 + ((tid + i * NT) / KQ);
      if (m < d.M) {
        if (AM == MMFN_A_IM2COL) {
          const int ohw = d.OH * d.OW;
          const int b = m / ohw, rem = m - b * ohw;
          const int oh = rem / d.OW, ow = rem - oh * d.OW;
          apos[i] = {b, oh * d.stride - d.pad, ow * d.stride - d.pad, true};
        } else {
          const int hw = d.H * d.W;
          const int b = m / hw, rem = m - b * hw;
          const int ih = rem / d.W, iw = rem - ih * d.W;
          apos[i] = {b, ih + d.pad, iw + d.pad, true};
        }
      }
    }
  }
  bool b_vec;
  if (BMODE == MMFN_B_NK || BMODE == MMFN_B_KN) b_vec = ((d.ldb & 3) == 0) && ((((uintptr_t)d.B) & 15) == 0);
  else b_vec = ((d.Cin & 3) == 0) && ((((uintptr_t)d.B) & 15) == 0);
  // B_IM2COL: the column (kh,kw,ci) of each unit is fixed for the whole k loop
  int bcol_kh[UB], bcol_kw[UB], bcol_ci[UB];
#pragma unroll
  for (int i = 0; i < UB; ++i) {
    bcol_kh[i] = bcol_kw[i] = bcol_ci[i] = 0;
    if (BMODE == MMFN_B_IM2COL) {
      const int n = n0 + ((tid + i * NT) % (BN / 4)) * 4;
      const int khw = n / d.Cin;
      bcol_ci[i] = n - khw * d.Cin;
      bcol_kh[i] = khw / d.KW;
      bcol_kw[i] = khw - bcol_kh[i] * d.KW;
    }
  }

  auto load_a = [&](int i, int kt) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int u = tid + i * NT;
    if (A_KC) {
      const int m = m0 + (u / KQ);
      const int k = kt * BK + (u % KQ) * 4;
      if (m >= d.M || k >= d.K) return v;
      if (AM == MMFN_A_ROWMAJOR) {
        const float* p = d.A + (size_t)m * d.lda + k;
        if (a_vec && k + 3 < d.K) return ld4(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < d.K) v[e] = p[e];
      } else if (AM == MMFN_A_IM2COL) {
        if (!apos[i].ok) return v;
        if (a_vec) {
          const int khw = k / d.Cin, ci = k - khw * d.Cin;
          const int kh = khw / d.KW, kw = khw - kh * d.KW;
          const int ih = apos[i].y0 + kh, iw = apos[i].x0 + kw;
          if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
            return ld4(d.A + ((size_t)(apos[i].b * d.H + ih) * d.W + iw) * d.Cin + ci);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            if (kk < d.K) {
              const int khw = kk / d.Cin, ci = kk - khw * d.Cin;
              const int kh = khw / d.KW, kw = khw - kh * d.KW;
              const int ih = apos[i].y0 + kh, iw = apos[i].x0 + kw;
              if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
                v[e] = d.A[((size_t)(apos[i].b * d.H + ih) * d.W + iw) * d.Cin + ci];
            }
          }
        }
      } else {  // MMFN_A_DGRAD: k = (kh,kw,co); contributing output pixel oh = (ih + pad - kh)/stride
        if (!apos[i].ok) return v;
        const int khw = k / d.Cout, co = k - khw * d.Cout;
        const int kh = khw / d.KW, kw = khw - kh * d.KW;
        const int th = apos[i].y0 - kh, tw = apos[i].x0 - kw;
        if (th < 0 || tw < 0) return v;
        int oh = th, ow = tw;
        if (d.stride == 2) {
          if ((th | tw) & 1) return v;
          oh = th >> 1; ow = tw >> 1;
        } else if (d.stride != 1) {
          if (th % d.stride || tw % d.stride) return v;
          oh = th / d.stride; ow = tw / d.stride;
        }
        if (oh >= d.OH || ow >= d.OW) return v;
        const float* p = d.A + ((size_t)(apos[i].b * d.OH + oh) * d.OW + ow) * d.Cout + co;
        if (a_vec) return ld4(p);  // Cout % 4 == 0: the 4 k's share one tap
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = k + q;
          if (kk < d.K && kk / d.Cout == khw) v[q] = p[q];
        }
      }
    } else {  // MMFN_A_COLMAJOR: A[k*lda + m]
      const int kk = kt * BK + u / (BM / 4);
      const int m = m0 + (u % (BM / 4)) * 4;
      if (kk >= d.K || m >= d.M) return v;
      const float* p = d.A + (size_t)kk * d.lda + m;
      if (a_vec && m + 3 < d.M) return ld4(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (m + e < d.M) v[e] = p[e];
    }
    return v;
  };

  auto load_b = [&](int i, int kt) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int u = tid + i * NT;
    if (B_KC) {
      const int n = n0 + (u / KQ);
      const int k = kt * BK + (u % KQ) * 4;
      if (n >= d.N || k >= d.K) return v;
      const float* p = d.B + (size_t)n * d.ldb + k;
      if (b_vec && k + 3 < d.K) return ld4(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (k + e < d.K) v[e] = p[e];
    } else {
      const int kk = kt * BK + u / (BN / 4);
      const int n = n0 + (u % (BN / 4)) * 4;
      if (kk >= d.K || n >= d.N) return v;
      if (BMODE == MMFN_B_KN) {
        const float* p = d.B + (size_t)kk * d.ldb + n;
        if (b_vec && n + 3 < d.N) return ld4(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] = p[e];
      } else if (BMODE == MMFN_B_DGRADW) {
        const int khw = kk / d.Cout, co = kk - khw * d.Cout;
        const float* p = d.B + ((size_t)co * KHW + khw) * d.Cin + n;
        if (b_vec && n + 3 < d.N) return ld4(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] = p[e];
      } else {  // MMFN_B_IM2COL: row kk = output pixel (b,oh,ow); column = (kh,kw,ci)
        const int ohw = d.OH * d.OW;
        const int b = kk / ohw, rem = kk - b * ohw;
        const int oh = rem / d.OW, ow = rem - oh * d.OW;
        if (b_vec) {
          const int ih = oh * d.stride - d.pad + bcol_kh[i], iw = ow * d.stride - d.pad + bcol_kw[i];
          if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
            return ld4(d.B + ((size_t)(b * d.H + ih) * d.W + iw) * d.Cin + bcol_ci[i]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int nn = n + e;
            if (nn < d.N) {
              const int khw = nn / d.Cin, ci = nn - khw * d.Cin;
              const int kh = khw / d.KW, kw = khw - kh * d.KW;
              const int ih = oh * d.stride - d.pad + kh, iw = ow * d.stride - d.pad + kw;
              if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
                v[e] = d.B[((size_t)(b * d.H + ih) * d.W + iw) * d.Cin + ci];
            }
          }
        }
      }
    }
    return v;
  };

  auto store_a = [&](float* As, int i, f32x4 v) {
    const int u = tid + i * NT;
    if (A_KC) *reinterpret_cast<f32x4*>(&As[(u / KQ) * LDK + kc_slot(u / KQ, u % KQ) * 4]) = v;
    else *reinterpret_cast<f32x4*>(&As[(u / (BM / 4)) * BM + (u % (BM / 4)) * 4]) = v;
  };
  auto store_b = [&](float* Bs, int i, f32x4 v) {
    const int u = tid + i * NT;
    if (B_KC) *reinterpret_cast<f32x4*>(&Bs[(u / KQ) * LDK + kc_slot(u / KQ, u % KQ) * 4]) = v;
    else *reinterpret_cast<f32x4*>(&Bs[(u / (BN / 4)) * BN + (u % (BN / 4)) * 4]) = v;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 ra[UA], rb[UB];
  if (kt_begin < kt_end) {
#pragma unroll
    for (int i = 0; i < UA; ++i) ra[i] = load_a(i, kt_begin);
#pragma unroll
    for (int i = 0; i < UB; ++i) rb[i] = load_b(i, kt_begin);
#pragma unroll
    for (int i = 0; i < UA; ++i) store_a(smem, i, ra[i]);
#pragma unroll
    for (int i = 0; i < UB; ++i) store_b(smem + A_ELEMS, i, rb[i]);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) {
#pragma unroll
      for (int i = 0; i < UA; ++i) ra[i] = load_a(i, kt + 1);
#pragma unroll
      for (int i = 0; i < UB; ++i) rb[i] = load_b(i, kt + 1);
    }
    const float* As = smem + cur * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * TM * 32 + i * 32 + l31;
        if (A_KC) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&As[row * LDK + kc_slot(row, c * 2 + h) * 4]);
          a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[i][j] = As[(c * 8 + h * 4 + j) * BM + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int col = wn * TN * 32 + i * 32 + l31;
        if (B_KC) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&Bs[col * LDK + kc_slot(col, c * 2 + h) * 4]);
          b[i][0] = t[0]; b[i][1] = t[1]; b[i][2] = t[2]; b[i][3] = t[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[i][j] = Bs[(c * 8 + h * 4 + j) * BN + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < TN; ++q)
            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[q][j], acc[i][q], 0, 0, 0);
    }
    if (more) {
      float* An = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
#pragma unroll
      for (int i = 0; i < UA; ++i) store_a(An, i, ra[i]);
#pragma unroll
      for (int i = 0; i < UB; ++i) store_b(An + A_ELEMS, i, rb[i]);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---------------- epilogue ----------------
  uint64_t key = 0;
  if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
  const bool to_slab = d.splitk > 1;
  float* slab = to_slab ? d.workspace + ((size_t)blockIdx.y * max(1, d_in.batch) + (d_in.batch > 1 ? blockIdx.z : 0)) * d.M * d.N : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      const int col = n0 + wn * TN * 32 + q * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        if (to_slab) slab[(size_t)row * d.N + col] = acc[i][q][r];
        else epilogue_store(d, key, row, col, acc[i][q][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------
// Fast path: same tiling and LDS image, but every global access is an unconditional 16-byte load.
//   * rows/columns beyond M/N are clamped to the last valid row (their products are never stored),
//   * convolution padding and invalid transposed-conv taps read a 16-byte zero page instead of
//     branching,
//   * Cin % BK == 0 (Cout for dgrad) so a k-tile never straddles a filter tap: the (kh, kw, c0) of
//     the tile is wave-uniform scalar state advanced once per tile — no per-thread division,
//   * wgrad decodes its pixel rows with shifts (OH*OW and OW are powers of two).
// Eligibility is checked on the host (fast_ok); everything else runs the generic kernel above.
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

// MMFN_GEMM_TIMELINE (experiment builds only, tools/experiments/gemm64_timeline.sh): every wave of the first g_tl_blocks blocks
// stamps s_memtime at its start, after the prologue's loads are issued, after each k-tile's barrier (operands of that tile
// landed everywhere), after the k-loop and after its last store - the per-k-tile issue timeline the thread trace would give,
// where no trace decoder is installed.  Slots: 0 start, 1 HW_ID, 2 prologue issued, 3 + kt tile kt ready (kt < 36), 40 loop
// done, 41 stores issued, 42 XCC_ID, 43 s_memrealtime at start, 44 at end.
#ifdef MMFN_GEMM_TIMELINE
constexpr int TL_SLOTS = 48;
__device__ unsigned long long* g_tl_buf = nullptr;
__device__ int g_tl_blocks = 0;
#define TL_MARK(slot) do { if (tl && lane == 0) tl[(slot)] = __builtin_readcyclecounter(); } while (0)
#define TL_END() do { MMFN_WAIT_VMCNT(0); if (tl && lane == 0) { tl[41] = __builtin_readcyclecounter(); tl[44] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TL_MARK(slot) do { } while (0)
#define TL_END() do { } while (0)
#endif

constexpr bool USE_GLDS = true;   // +5-10 % over register staging on every shape measured (tools/gemm_bench.py)
// source-side slot of unit u (k-contiguous operands): the physical LDS slot u % KQ holds logical slot
// (u % KQ) ^ swizzle(row) when the tile is written lane-linearly by global_load_lds
#define SRCQ(u) (USE_GLDS ? kc_slot((u) / KQ, (u) % KQ) : (u) % KQ)
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

#ifndef MMFN_F32_STAGES
#define MMFN_F32_STAGES 3         // 128-row / 128-column tiles
#endif
#ifndef MMFN_F32_STAGES_SMALL
#define MMFN_F32_STAGES_SMALL 3   // 64x64 tiles (8 KB per stage)
#endif
// vmcnt(n) only (expcnt / lgkmcnt untouched): gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define MMFN_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x70 | 0xF00)

// LNF (NT form only): MMFN_EPI_LN_FOLD - LayerNorm of the A rows folded into the product (mmfn_gemm_desc.ln_c1): every lane
// sums the A fragment values it feeds to the MFMAs (lane (l31, h) of a wave row sees row l31's k values of its half h: the two
// halves together see the whole row once), the halves are combined with one shuffle, the row statistics go through 2 * BM
// floats of LDS to the lanes that hold that row's accumulators, and the interior-tile epilogue starts from
// rstd * (acc - mean * c1[n]) + c2[n] instead of acc + bias[n].
// (amdgpu_waves_per_eu: the LNF 128x128 instantiation took 192 registers = two blocks per CU; held to three like the plain form)
// WM_ x WN_: the block's waves as a grid over the tile; each wave owns (BM / WM_) x (BN / WN_) outputs as TM x TN MFMA tiles of
// 32 x 32.  2 x 2 waves everywhere but: 192 x 64 / 64 x 192 (wave tile 96 x 32 / 32 x 96: the M = 6144 transformer GEMMs divide
// into exactly 256 / 768 / 1024 blocks - whole rounds of the 256 CUs - where 128 x 128 leaves 192 or 576), and the two-wave
// 64 x 64 block (wave tile 32 x 64: three fragment reads per eight MFMAs instead of four).
template <int AM, int BMODE, int BM, int BN, bool LNF = false, int WM_ = 2, int WN_ = 2>
__global__ __launch_bounds__(64 * WM_ * WN_) __attribute__((amdgpu_waves_per_eu(LNF && BM == 128 && BN == 128 ? 3 : 1)))
void gemm_f32_fast_kernel(const mmfn_gemm_desc d_in, const int kt_per_split, const int tiles_n,
                                                           const int log2_ow, const int log2_ohw) {
  // Batched launches (the 36 frequency GEMMs of a Winograd convolution): a batch entry's tiles all on ONE XCD.  With the plain order
  // (x = tile fastest, z = entry) the tiles of an entry are dealt round-robin over the eight XCDs, so every L2 fetches that entry's
  // filter panel U_f (and, across column tiles, its activation panel) for itself: 58 MB fetched per layer3 launch against 20 MB of
  // distinct operands (profiles/r04a_pmc.txt).  Here XCD x (= linear block id % 8, the observed dispatch policy) takes the entries
  // x, x + 8, x + 16, ... whole, and the batch % 8 left-over entries are split over 8 / (batch % 8) XCDs each (36 entries: four whole
  // entries and half of a fifth per XCD).  bz / by / bxr: the (entry, k-split, tile) this block computes.
  int bz = blockIdx.z, by = blockIdx.y, bxr = blockIdx.x;
  bool xcd_batch = false;
  if (d_in.batch > 1 && !d_in.dg_parity) {
    const int U = gridDim.x * gridDim.y, full = d_in.batch >> 3, rem = d_in.batch & 7;
    const int shares = rem ? 8 / rem : 1;
    if (rem == 0 || ((8 % rem) == 0 && (U % shares) == 0)) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = lin & 7, s = lin >> 3;
      int r;
      if (s < full * U) { bz = xcd + 8 * (s / U); r = s % U; }
      else { bz = 8 * full + xcd / shares; r = (s - full * U) + (xcd % shares) * (U / shares); }
      by = r / (int)gridDim.x;
      bxr = r - by * (int)gridDim.x;
      xcd_batch = true;
    }
  }
  const mmfn_gemm_desc d = d_in.dg_parity ? d_in : batch_view(d_in, bz);
  // Stride-2 transposed convolution, decomposed by output-pixel parity (blockIdx.z = 2*py + px): an input
  // pixel (ih, iw) only receives taps with kh == (ih + pad) mod 2, kw == (iw + pad) mod 2, so each of the
  // four parity classes is a dense GEMM over its own 1/2/2/4 (3x3) taps instead of 9 taps of which 3/4
  // multiply zeros.
  const bool dgp = (AM == MMFN_A_DGRAD) && d.dg_parity;
  int py = 0, px = 0, kh0 = 0, kw0 = 0, nkw = d.KW, Mloc = d.M, Kloc = d.K;
  if (dgp) {
    // heaviest class first: with pad 1 (every 3x3 here) class (py, px) = (1, 1) has four taps and (0, 0) one; blocks are
    // dispatched z-major, and the long blocks must not be the ones left for the tail
    py = 1 - (int)(blockIdx.z >> 1); px = 1 - (int)(blockIdx.z & 1);
    kh0 = (py + d.pad) & 1; kw0 = (px + d.pad) & 1;
    const int nkh = (d.KH - kh0 + 1) >> 1;
    nkw = (d.KW - kw0 + 1) >> 1;
    Mloc = d.M >> 2;
    Kloc = nkh * nkw * d.Cout;
  }
  constexpr bool A_KC = (AM != MMFN_A_COLMAJOR);
  constexpr bool B_KC = (BMODE == MMFN_B_NK);
  constexpr int WAVES_N = WN_;
  constexpr int NTH = 64 * WM_ * WN_;
  constexpr int TM = BM / (32 * WM_), TN = BN / (32 * WN_);
  static_assert(TM * 32 * WM_ == BM && TN * 32 * WN_ == BN, "tile must divide into 32 x 32 MFMA tiles per wave");
  static_assert((BM * BK / 4) % NTH == 0 && (BN * BK / 4) % NTH == 0, "whole 16-byte staging units per thread");
  constexpr int LDK = BK;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;
  constexpr int UA = BM * BK / 4 / NTH, UB = BN * BK / 4 / NTH;
  // LDS stages of the operand pipeline (global_load_lds path).  NS >= 3: tiles kt+1 .. kt+NS-2 are in flight while tile kt is
  // multiplied, retired with COUNTED vmcnt waits, one raw s_barrier per k-tile - the step's GEMMs are a few hundred 64x64
  // tiles each (1-3 blocks per CU) with 36-288 k-tiles, so it is the depth of each block's operand stream, not occupancy,
  // that hides the L2 / HBM latency (wait_inst 0.6 of the wave cycles on the double-buffered form, profiles/r02d_pmc.txt).
  // The stage overwritten in iteration kt was last read in iteration kt-2, two barriers back.  NS = 2: the double buffer.
  constexpr int NS = USE_GLDS ? ((BM == 64 && BN == 64) ? MMFN_F32_STAGES_SMALL : MMFN_F32_STAGES) : 2;
  constexpr int D = NS == 2 ? 1 : NS - 2;
  __shared__ __attribute__((aligned(16))) float smem[NS * (A_ELEMS + B_ELEMS)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
#ifdef MMFN_GEMM_TIMELINE
  unsigned long long* tl = nullptr;
  {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (g_tl_buf && (int)lin < g_tl_blocks) {
      tl = g_tl_buf + ((size_t)lin * (NTH / 64) + wave) * TL_SLOTS;
      if (lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl[0] = __builtin_readcyclecounter();
        tl[1] = hwid; tl[42] = xcc;
        tl[43] = __builtin_amdgcn_s_memrealtime();
      }
    }
  }
#endif
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch policy; speed only).  Give each
  // XCD a contiguous run of tiles so the blocks that share an A row-panel / B column-panel hit the same L2.
  int bid;
  if (xcd_batch) {
    bid = bxr;   // the entry's tiles already share an XCD
  } else {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, r = nb & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int nkt = Kloc / BK;
  const int kt_begin = by * kt_per_split;
  const int kt_end = min(nkt, kt_begin + kt_per_split);
  const int KHW = d.KH * d.KW;
  const float* zero = g_zero_page;

  // ---- A loader state
  const float* pa[UA];
  int ay0[UA], ax0[UA];
#pragma unroll
  for (int i = 0; i < UA; ++i) {
    const int u = tid + i * NTH;
    ay0[i] = ax0[i] = 0;
    if (AM == MMFN_A_ROWMAJOR) {
      pa[i] = d.A + (size_t)min(m0 + u / KQ, d.M - 1) * d.lda + SRCQ(u) * 4;
    } else if (AM == MMFN_A_COLMAJOR) {
      pa[i] = d.A + (size_t)(u / (BM / 4)) * d.lda + min(m0 + (u % (BM / 4)) * 4, d.M - 4);
    } else if (AM == MMFN_A_IM2COL) {
      const int m = min(m0 + u / KQ, d.M - 1);
      const int ohw = d.OH * d.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oh = rem / d.OW, ow = rem - oh * d.OW;
      ay0[i] = oh * d.stride - d.pad;
      ax0[i] = ow * d.stride - d.pad;
      pa[i] = d.A + (size_t)b * d.H * d.W * d.Cin + SRCQ(u) * 4;
    } else {
      const int m = min(m0 + u / KQ, Mloc - 1);
      int b, ih, iw;
      if (dgp) {
        const int w2 = d.W >> 1, hw2 = (d.H >> 1) * w2;
        b = m / hw2;
        const int rem = m - b * hw2;
        ih = 2 * (rem / w2) + py;
        iw = 2 * (rem % w2) + px;
      } else {
        const int hw = d.H * d.W;
        b = m / hw;
        const int rem = m - b * hw;
        ih = rem / d.W;
        iw = rem - ih * d.W;
      }
      ay0[i] = ih + d.pad;
      ax0[i] = iw + d.pad;
      pa[i] = d.A + (size_t)b * d.OH * d.OW * d.Cout + SRCQ(u) * 4;
    }
  }
  // ---- B loader state
  const float* pb[UB];
  int bkh[UB], bkw[UB];
#pragma unroll
  for (int i = 0; i < UB; ++i) {
    const int u = tid + i * NTH;
    bkh[i] = bkw[i] = 0;
    if (BMODE == MMFN_B_NK) {
      pb[i] = d.B + (size_t)min(n0 + u / KQ, d.N - 1) * d.ldb + SRCQ(u) * 4;
    } else if (BMODE == MMFN_B_KN) {
      pb[i] = d.B + (size_t)(u / (BN / 4)) * d.ldb + min(n0 + (u % (BN / 4)) * 4, d.N - 4);
    } else if (BMODE == MMFN_B_DGRADW) {
      pb[i] = d.B + (size_t)(u / (BN / 4)) * KHW * d.Cin + min(n0 + (u % (BN / 4)) * 4, d.N - 4);
    } else {  // B_IM2COL: column (kh,kw,ci) fixed per unit
      const int n = min(n0 + (u % (BN / 4)) * 4, d.N - 4);
      const int khw = n / d.Cin;
      bkh[i] = khw / d.KW;
      bkw[i] = khw - bkh[i] * d.KW;
      pb[i] = d.B + (n - khw * d.Cin);
    }
  }
  // ---- wave-uniform tap state of the NEXT tile to load (conv modes)
  int t_kh = 0, t_kw = 0, t_c0 = 0;
  {
    const int chan = (AM == MMFN_A_IM2COL) ? d.Cin : d.Cout;
    if (AM == MMFN_A_IM2COL || AM == MMFN_A_DGRAD) {
      const int k0 = kt_begin * BK;
      const int tap = k0 / chan;
      t_c0 = k0 - tap * chan;
      t_kh = tap / nkw;
      t_kw = tap - t_kh * nkw;
    }
  }
  auto advance_tap = [&]() {
    const int chan = (AM == MMFN_A_IM2COL) ? d.Cin : d.Cout;
    t_c0 += BK;
    if (t_c0 == chan) { t_c0 = 0; if (++t_kw == nkw) { t_kw = 0; ++t_kh; } }
  };

  auto src_a = [&](int i, int kt) -> const float* {
    if (AM == MMFN_A_ROWMAJOR) return pa[i] + (size_t)kt * BK;
    if (AM == MMFN_A_COLMAJOR) return pa[i] + (size_t)kt * BK * d.lda;
    if (AM == MMFN_A_IM2COL) {
      const int ih = ay0[i] + t_kh, iw = ax0[i] + t_kw;
      const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      return ok ? pa[i] + ((size_t)ih * d.W + iw) * d.Cin + t_c0 : zero;
    }
    int oh = ay0[i] - (dgp ? kh0 + 2 * t_kh : t_kh), ow = ax0[i] - (dgp ? kw0 + 2 * t_kw : t_kw);
    bool ok = oh >= 0 && ow >= 0;
    if (d.stride == 2) { ok = ok && !((oh | ow) & 1); oh >>= 1; ow >>= 1; }
    else if (d.stride != 1) { ok = ok && (oh % d.stride == 0) && (ow % d.stride == 0); oh /= d.stride; ow /= d.stride; }
    ok = ok && oh < d.OH && ow < d.OW;
    return ok ? pa[i] + ((size_t)oh * d.OW + ow) * d.Cout + t_c0 : zero;
  };
  auto src_b = [&](int i, int kt) -> const float* {
    if (BMODE == MMFN_B_NK) return pb[i] + (size_t)kt * BK;
    if (BMODE == MMFN_B_KN) return pb[i] + (size_t)kt * BK * d.ldb;
    if (BMODE == MMFN_B_DGRADW)
      return pb[i] + ((size_t)t_c0 * KHW + (dgp ? (kh0 + 2 * t_kh) * d.KW + (kw0 + 2 * t_kw) : t_kh * d.KW + t_kw)) * d.Cin;
    const int kk = kt * BK + (tid + i * NTH) / (BN / 4);
    const int b = kk >> log2_ohw, rem = kk & ((1 << log2_ohw) - 1);
    const int oh = rem >> log2_ow, ow = rem & ((1 << log2_ow) - 1);
    const int ih = oh * d.stride - d.pad + bkh[i], iw = ow * d.stride - d.pad + bkw[i];
    const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
    return ok ? pb[i] + ((size_t)(b * d.H + ih) * d.W + iw) * d.Cin : zero;
  };
  auto store_a = [&](float* As, int i, f32x4 v) {
    const int u = tid + i * NTH;
    if (USE_GLDS) *reinterpret_cast<f32x4*>(&As[u * 4]) = v;
    else if (A_KC) *reinterpret_cast<f32x4*>(&As[(u / KQ) * LDK + kc_slot(u / KQ, u % KQ) * 4]) = v;
    else *reinterpret_cast<f32x4*>(&As[(u / (BM / 4)) * BM + (u % (BM / 4)) * 4]) = v;
  };
  auto store_b = [&](float* Bs, int i, f32x4 v) {
    const int u = tid + i * NTH;
    if (USE_GLDS) *reinterpret_cast<f32x4*>(&Bs[u * 4]) = v;
    else if (B_KC) *reinterpret_cast<f32x4*>(&Bs[(u / KQ) * LDK + kc_slot(u / KQ, u % KQ) * 4]) = v;
    else *reinterpret_cast<f32x4*>(&Bs[(u / (BN / 4)) * BN + (u % (BN / 4)) * 4]) = v;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 ra[UA], rb[UB];
  // LNF: this lane's partial (sum x, sum x^2) of rows wm*TM*32 + i*32 + l31.  Running sums in fp64 (E[x^2] - mean^2 cancels badly
  // in fp32 when a row's mean is large against its spread), fed once per k-tile from fp32 partial sums of that tile's 8 values per
  // lane: two fp64 additions per row and k-tile instead of three fp64 instructions per value
  double ln_s1[TM], ln_s2[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { ln_s1[i] = 0.0; ln_s2[i] = 0.0; }
  // MMFN_EPI_COLSUM_A: this lane's partial column sums of A (rows wm*TM*32 + i*32 + l31, its half of every k-chunk): a k-tile's
  // eight values in fp32, the running sum over the k-tiles in fp64 (thousands of sequential fp32 additions would cost 2e-6 relative)
  double csa[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csa[i] = 0.0;
  // stage(kt, dst): global -> LDS for one k-tile.  With USE_GLDS the 16-byte pieces go straight to LDS
  // (global_load_lds: wave-uniform LDS base + lane*16, so the image is lane-linear and the slot swizzle
  // is applied to the SOURCE address); otherwise through registers (issue now, ds_write after the MFMAs).
  auto stage_issue = [&](int kt, float* dst) {
    if (USE_GLDS) {
#pragma unroll
      for (int i = 0; i < UA; ++i) glds16(src_a(i, kt), dst + (i * NTH + wave * 64) * 4);
#pragma unroll
      for (int i = 0; i < UB; ++i) glds16(src_b(i, kt), dst + A_ELEMS + (i * NTH + wave * 64) * 4);
    } else {
#pragma unroll
      for (int i = 0; i < UA; ++i) ra[i] = ld4(src_a(i, kt));
#pragma unroll
      for (int i = 0; i < UB; ++i) rb[i] = ld4(src_b(i, kt));
    }
    advance_tap();
  };
  auto stage_commit = [&](float* dst) {
    if (!USE_GLDS) {
#pragma unroll
      for (int i = 0; i < UA; ++i) store_a(dst, i, ra[i]);
#pragma unroll
      for (int i = 0; i < UB; ++i) store_b(dst + A_ELEMS, i, rb[i]);
    }
  };
  constexpr int STG = A_ELEMS + B_ELEMS;
  if (NS > 2) {
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (kt_begin + s < kt_end) stage_issue(kt_begin + s, smem + s * STG);
  } else {
    if (kt_begin < kt_end) {
      stage_issue(kt_begin, smem);
      stage_commit(smem);
    }
    __syncthreads();
  }

  TL_MARK(2);
  int cur = 0, nxt = D;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (NS > 2) {
      // tiles kt+1 .. kt+D stay in flight; near the end fewer exist, and the (immediate) wait count shrinks with them
      const int ahead = kt_end - 1 - kt;            // tiles after kt
      if (ahead >= D) stage_issue(kt + D, smem + nxt * STG);
      if (ahead >= D) MMFN_WAIT_VMCNT((UA + UB) * D);   // this wave's pieces of tile kt have landed ...
      else if (D > 1 && ahead == 1) MMFN_WAIT_VMCNT(UA + UB);
      else MMFN_WAIT_VMCNT(0);
      __builtin_amdgcn_s_barrier();                 // ... and everybody else's
#ifdef MMFN_GEMM_TIMELINE
      if (kt - kt_begin < 36) TL_MARK(3 + kt - kt_begin);
#endif
    } else if (more) {
      stage_issue(kt + 1, smem + (cur ^ 1) * STG);
    }
    const float* As = smem + cur * STG;
    const float* Bs = As + A_ELEMS;
    float ln_p1[TM], ln_p2[TM], cs_p[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ln_p1[i] = 0.f; ln_p2[i] = 0.f; cs_p[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * TM * 32 + i * 32 + l31;
        if (A_KC) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&As[row * LDK + kc_slot(row, c * 2 + h) * 4]);
          a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[i][j] = As[(c * 8 + h * 4 + j) * BM + row];
        }
        if (LNF) {   // the k-tile's 16 values per row in fp32, the running sums over the k-tiles in fp64 (below)
#pragma unroll
          for (int j = 0; j < 4; ++j) { ln_p1[i] += a[i][j]; ln_p2[i] = fmaf(a[i][j], a[i][j], ln_p2[i]); }
        }
        if (AM == MMFN_A_COLMAJOR && BMODE == MMFN_B_KN) cs_p[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);   // (not in the conv weight-gradient forms)
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int col = wn * TN * 32 + i * 32 + l31;
        if (B_KC) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(&Bs[col * LDK + kc_slot(col, c * 2 + h) * 4]);
          b[i][0] = t[0]; b[i][1] = t[1]; b[i][2] = t[2]; b[i][3] = t[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[i][j] = Bs[(c * 8 + h * 4 + j) * BN + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < TN; ++q)
            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[q][j], acc[i][q], 0, 0, 0);
    }
    if (LNF) {
#pragma unroll
      for (int i = 0; i < TM; ++i) { ln_s1[i] += (double)ln_p1[i]; ln_s2[i] += (double)ln_p2[i]; }
    }
    if (AM == MMFN_A_COLMAJOR && BMODE == MMFN_B_KN) {
#pragma unroll
      for (int i = 0; i < TM; ++i) csa[i] += (double)cs_p[i];
    }
    if (NS > 2) {
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    } else {
      if (more) stage_commit(smem + (cur ^ 1) * STG);
      __syncthreads();
      cur ^= 1;
    }
  }

  TL_MARK(40);
  __shared__ float ln_stat[LNF ? 2 * BM : 1];
  if (LNF) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const double s1 = ln_s1[i] + __shfl_xor(ln_s1[i], 32, 64), s2 = ln_s2[i] + __shfl_xor(ln_s2[i], 32, 64);
      const double mud = s1 / (double)d.K;
      const double var = fmax(s2 / (double)d.K - mud * mud, 0.0);
      const float mu = (float)mud;
      const float rs = (float)(1.0 / sqrt(var + (double)d.ln_eps));
      if (wn == 0 && h == 0) {
        const int rl = wm * TM * 32 + i * 32 + l31;
        ln_stat[rl] = mu;
        ln_stat[BM + rl] = rs;
        if (n0 == 0 && d.ln_mean) { d.ln_mean[m0 + rl] = mu; d.ln_rstd[m0 + rl] = rs; }
      }
    }
    __syncthreads();
  }
  uint64_t key = 0;
  if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
  const bool to_slab = d.splitk > 1;
  float* slab = to_slab ? d.workspace + ((size_t)by * max(1, d_in.batch) + (d_in.batch > 1 ? bz : 0)) * d.M * d.N : nullptr;
  if (AM == MMFN_A_COLMAJOR && BMODE == MMFN_B_KN && (d.flags & MMFN_EPI_COLSUM_A) && wn == 0 && n0 == 0) {
    // the two lane halves saw complementary k's of the same rows; one block column (n0 == 0) of every row tile and k-slice writes:
    // without split-K straight into colsum, else into its slice's row of the partials behind the slabs (combined in slice order by
    // the split-K combine launch)
    float* dst = to_slab ? d.workspace + (size_t)d.splitk * d.M * d.N + (size_t)by * d.M : d.colsum;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float v = (float)(csa[i] + __shfl_xor(csa[i], 32, 64));
      const int row = m0 + wm * TM * 32 + i * 32 + l31;
      if (h == 0 && row < d.M) dst[row] = v;
    }
  }
  constexpr int EPI_OPS = MMFN_EPI_BIAS | MMFN_EPI_RELU | MMFN_EPI_GELU | MMFN_EPI_MASK_AUX | MMFN_EPI_DROPOUT | MMFN_EPI_RESIDUAL |
                          MMFN_EPI_ACCUM | MMFN_EPI_RELU_LAST;
  // Plain stores of an interior tile (no epilogue operation, or a split slab): one pointer per lane and 16 * TM * TN stores at
  // compile-time row multiples of the leading dimension.  The general loop below tests the tile edge and eight epilogue flags per
  // ELEMENT; tools/experiments/gemm32_pmc.sh counts ~730 non-MFMA instructions per wave around a tile's k-loop, most of them there -
  // more than the k-loop itself issues for the K = 64 ... 256 Winograd-domain GEMMs that are two thirds of the step's launches.
  if (!dgp && m0 + BM <= d.M && n0 + BN <= d.N && (to_slab || !(d.flags & EPI_OPS))) {
    const int ld = to_slab ? d.N : d.ldc;
    float* p0 = (to_slab ? slab : d.C) + (size_t)(m0 + wm * TM * 32 + 4 * h) * ld + n0 + wn * TN * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) p0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ld + q * 32] = acc[i][q][r];
    TL_END();
    return;
  }
  // Interior tile with the common epilogue operations (bias, ReLU, ReLU-backward mask, dropout, residual, final ReLU): the same
  // per-lane pointers; the flags are read once per tile instead of once per element.
  if (!dgp && !to_slab && m0 + BM <= d.M && n0 + BN <= d.N && !(d.flags & (MMFN_EPI_GELU | MMFN_EPI_ACCUM))) {
    const int f = d.flags;
    const size_t lrow = (size_t)(m0 + wm * TM * 32 + 4 * h);
    const int lcol = n0 + wn * TN * 32 + l31;
    float* p0 = d.C + lrow * d.ldc + lcol;
    const float* r0 = (f & MMFN_EPI_RESIDUAL) ? d.res + lrow * d.ldr + lcol : nullptr;
    const float* a0 = (f & MMFN_EPI_MASK_AUX) ? d.aux + lrow * d.ldaux + lcol : nullptr;
    // wave-uniform branches, NOT a max against -inf when the flag is off: v_max_f32 returns the non-NaN operand, so a NaN
    // accumulator would leave an interior tile as -inf while the edge tiles' general path (epilogue_store) keeps it NaN
    const bool relu1 = (f & MMFN_EPI_RELU) != 0, relu2 = (f & MMFN_EPI_RELU_LAST) != 0;
    const bool drop = (f & MMFN_EPI_DROPOUT) != 0;
    const float inv_keep = drop ? 1.0f / (1.0f - d.drop_p) : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const float bias = (f & MMFN_EPI_BIAS) ? d.bias[lcol + q * 32] : 0.0f;
        const float c1 = LNF ? d.ln_c1[lcol + q * 32] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
          float v;
          if (LNF) {
            const int rl = wm * TM * 32 + dr + 4 * h;
            v = fmaf(ln_stat[BM + rl], fmaf(-ln_stat[rl], c1, acc[i][q][r]), bias);   // rstd * (acc - mean * c1) + c2
          } else {
            v = acc[i][q][r] + bias;
          }
          if (relu1) v = fmaxf(v, 0.0f);
          if (a0) v = a0[(size_t)dr * d.ldaux + q * 32] > 0.0f ? v : 0.0f;
          if (drop)
            v *= mmfn_dropout_scale(key, (uint64_t)(lrow + dr) * (uint64_t)d.N + (uint64_t)(lcol + q * 32), d.drop_p, inv_keep);
          if (r0) v += r0[(size_t)dr * d.ldr + q * 32];
          if (relu2) v = fmaxf(v, 0.0f);
          p0[(size_t)dr * d.ldc + q * 32] = v;
        }
      }
    TL_END();
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      const int col = n0 + wn * TN * 32 + q * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= Mloc) continue;
        if (dgp) {  // local (b, i, j) -> input pixel (b, 2i+py, 2j+px)
          const int w2 = d.W >> 1, hw2 = (d.H >> 1) * w2;
          const int bb = row / hw2, rem = row - bb * hw2;
          row = (bb * d.H + 2 * (rem / w2) + py) * d.W + 2 * (rem % w2) + px;
        }
        if (to_slab) slab[(size_t)row * d.N + col] = acc[i][q][r];
        else epilogue_store(d, key, row, col, acc[i][q][r]);
      }
    }
  TL_END();
}

// ------------------------------------------------------------------------------------------
// bf16-operand mode (MMFN_EPI_BF16_OPERANDS, plain GEMM forms): A and B stay fp32 in HBM and are rounded to bf16
// (v_cvt_pk_bf16_f32, round to nearest even) on their way into LDS; v_mfma_f32_32x32x16_bf16 accumulates in fp32 and the
// epilogue / outputs are fp32 as everywhere else.  16x the MFMA rate of the fp32 instruction, so the kernel is bound by the
// fp32 operand traffic instead: 3-4x faster than the fp32 kernel on the transformer shapes (tools/experiments).
//   tile 128x128, 4 waves (2x2 of 64x64), BK = 32: LDS rows of 32 bf16 (64 B) whose 16-byte slots are XOR-swizzled by row so
//   the fragment ds_read_b128 is conflict-free; k-contiguous operands store 4 consecutive k per float4 directly, m-contiguous
//   operands (dX's weights, dW's activations) load a 4(k) x 4(m) micro-tile and transpose it in registers first.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int HBK = 32;  // k-tile of the bf16 kernel

__device__ __forceinline__ int hslot(int row, int slot) { return slot ^ ((row >> 2) & 3); }

// TERMS = 1: plain bf16 operands.  TERMS = 3 (MMFN_EPI_BF16X3): fp32 emulation - every operand element is split exactly into
// three bf16 terms x = hi + mid + lo (8 + 8 + 8 significand bits) kept as three LDS planes, and each product is formed from the
// six leading cross terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid (each exact in the fp32 accumulator; the dropped
// mid*lo, lo*mid, lo*lo are below 2^-24 relative), i.e. fp32-accurate at 16/6 = 2.7x the fp32 MFMA rate.
// CONV = 0: plain operands.  CONV = 1: A is the implicit im2col matrix of a convolution input (rows = output pixels,
// k = (tap, ci), k-contiguous; needs Cin % 32 == 0 so that a k-tile stays inside one tap) - forward convolutions and, over a
// flipped filter, stride-1 data gradients.  CONV = 2: B is that im2col matrix with k = pixels and n = (tap, ci) (the weight
// gradient dW = dY^T . im2col(x); OW and OH*OW powers of two).  In bf16 mode every convolution but the two 7x7 stems and the
// stride-2 data gradients runs here as a DIRECT convolution: rounding Winograd-domain operands to bf16 amplifies the error
// (the F(4x4,3x3) transforms have gains up to 100), measured as a gradient cosine of 0.65 against 0.79 for torch's own autocast.
template <bool A_KC, bool B_KC, int BM, int BN, int TERMS, int CONV = 0>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(const mmfn_gemm_desc d_in, const int kt_per_split, const int tiles_n,
                                                       const int log2_ow = 0, const int log2_ohw = 0) {
  const mmfn_gemm_desc d = batch_view(d_in);
  constexpr int TM = BM / 64, TN = BN / 64;            // 32x32 accumulator tiles per wave (2x2 waves)
  constexpr int UA = BM * 8 / NT, UB = BN * 8 / NT;    // k-contiguous staging units (row, k-quad) per thread
  constexpr int NBUF = TERMS == 1 ? 2 : 1;             // three planes: single LDS stage (48 KB), two barriers per k-tile
  constexpr int PLANE = (BM + BN) * HBK;
  __shared__ __attribute__((aligned(16))) __bf16 sm[NBUF][TERMS * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int nkt = d.K / HBK;
  const int kt_begin = blockIdx.y * kt_per_split, kt_end = min(nkt, kt_begin + kt_per_split);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // m-contiguous staging: one 4(k) x 4(m) micro-tile per thread, (k-quad, m-quad) = (tid / (rows/4), tid % (rows/4));
  // with a 64-row operand only the first 128 threads carry one
  constexpr int AQ = BM / 4, BQ = BN / 4;
  const bool a_on = A_KC || tid < AQ * 8, b_on = B_KC || tid < BQ * 8;
  f32x4 ra[4], rb[4];
  const float* pa[4];
  const float* pb[4];
  int ay0[4] = {0, 0, 0, 0}, ax0[4] = {0, 0, 0, 0};   // CONV 1: top-left input pixel of the unit's output pixel
  int bkh = 0, bkw = 0;                                // CONV 2: tap of the thread's n-quad
  const float* zero = g_zero_page;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + i * NT;
    if (CONV == 1) {
      const int m = min(m0 + (u >> 3), d.M - 1);
      const int ohw = d.OH * d.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oh = rem / d.OW, ow = rem - oh * d.OW;
      ay0[i] = oh * d.stride - d.pad;
      ax0[i] = ow * d.stride - d.pad;
      pa[i] = d.A + (size_t)b * d.H * d.W * d.Cin + (u & 7) * 4;
    } else if (A_KC) pa[i] = d.A + (size_t)min(m0 + (u >> 3), d.M - 1) * d.lda + (u & 7) * 4;
    else pa[i] = d.A + (size_t)(((tid / AQ) & 7) * 4 + i) * d.lda + min(m0 + (tid % AQ) * 4, d.M - 4);
    if (CONV == 2) {
      const int n = min(n0 + (tid % BQ) * 4, d.N - 4);
      const int khw = n / d.Cin;
      bkh = khw / d.KW;
      bkw = khw - bkh * d.KW;
      pb[i] = d.B + (n - khw * d.Cin);
    } else if (B_KC) pb[i] = d.B + (size_t)min(n0 + (u >> 3), d.N - 1) * d.ldb + (u & 7) * 4;
    else pb[i] = d.B + (size_t)(((tid / BQ) & 7) * 4 + i) * d.ldb + min(n0 + (tid % BQ) * 4, d.N - 4);
  }
  auto load = [&](int kt) {
    int t_kh = 0, t_kw = 0, t_c0 = 0;
    if (CONV == 1) {   // wave-uniform tap of this k-tile
      const int k0 = kt * HBK, tap = k0 / d.Cin;
      t_c0 = k0 - tap * d.Cin;
      t_kh = tap / d.KW;
      t_kw = tap - t_kh * d.KW;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (A_KC ? i < UA : a_on) {
        if (CONV == 1) {
          const int ih = ay0[i] + t_kh, iw = ax0[i] + t_kw;
          const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
          ra[i] = ld4(ok ? pa[i] + ((size_t)ih * d.W + iw) * d.Cin + t_c0 : zero);
        } else {
          ra[i] = ld4(pa[i] + (A_KC ? (size_t)kt * HBK : (size_t)kt * HBK * d.lda));
        }
      }
      if (B_KC ? i < UB : b_on) {
        if (CONV == 2) {
          const int kk = kt * HBK + ((tid / BQ) & 7) * 4 + i;   // pixel index of this k row
          const int b = kk >> log2_ohw, rem = kk & ((1 << log2_ohw) - 1);
          const int oh = rem >> log2_ow, ow = rem & ((1 << log2_ow) - 1);
          const int ih = oh * d.stride - d.pad + bkh, iw = ow * d.stride - d.pad + bkw;
          const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
          rb[i] = ld4(ok ? pb[i] + ((size_t)(b * d.H + ih) * d.W + iw) * d.Cin : zero);
        } else {
          rb[i] = ld4(pb[i] + (B_KC ? (size_t)kt * HBK : (size_t)kt * HBK * d.ldb));
        }
      }
    }
  };
  auto put = [&](__bf16* base, int off, const float* x) {  // 4 consecutive k of one row -> 1 or 3 bf16x4 planes
    bf16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (__bf16)x[e];
      if (TERMS == 3) {
        const float r1 = x[e] - (float)hi[e];
        mid[e] = (__bf16)r1;
        lo[e] = (__bf16)(r1 - (float)mid[e]);
      }
    }
    *reinterpret_cast<bf16x4*>(&base[off]) = hi;
    if (TERMS == 3) {
      *reinterpret_cast<bf16x4*>(&base[PLANE + off]) = mid;
      *reinterpret_cast<bf16x4*>(&base[2 * PLANE + off]) = lo;
    }
  };
  auto store_kc = [&](__bf16* base, const f32x4* r, int units) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= units) break;
      const int u = tid + i * NT, row = u >> 3, q = u & 7;
      const float x[4] = {r[i][0], r[i][1], r[i][2], r[i][3]};
      put(base, row * HBK + hslot(row, q >> 1) * 8 + (q & 1) * 4, x);
    }
  };
  auto store_mc = [&](__bf16* base, const f32x4* r, int quads) {  // r[j] = 4 consecutive rows at k = 4*kq + j
    const int kq = tid / quads, mq = tid % quads;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = mq * 4 + e;
      const float x[4] = {r[0][e], r[1][e], r[2][e], r[3][e]};
      put(base, row * HBK + hslot(row, kq >> 1) * 8 + (kq & 1) * 4, x);
    }
  };
  auto store = [&](int buf) {
    if (A_KC) store_kc(sm[buf], ra, UA); else if (a_on) store_mc(sm[buf], ra, AQ);
    if (B_KC) store_kc(sm[buf] + BM * HBK, rb, UB); else if (b_on) store_mc(sm[buf] + BM * HBK, rb, BQ);
  };
  if (kt_begin < kt_end) {
    load(kt_begin);
    store(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) load(kt + 1);
    const __bf16* As = sm[cur];
    const __bf16* Bs = sm[cur] + BM * HBK;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 a[TERMS][TM], b[TERMS][TN];
#pragma unroll
      for (int p = 0; p < TERMS; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ar = wm * TM * 32 + i * 32 + l31;
          a[p][i] = *reinterpret_cast<const bf16x8*>(&As[p * PLANE + ar * HBK + hslot(ar, 2 * s2 + h) * 8]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int br = wn * TN * 32 + j * 32 + l31;
          b[p][j] = *reinterpret_cast<const bf16x8*>(&Bs[p * PLANE + br * HBK + hslot(br, 2 * s2 + h) * 8]);
        }
      }
      // term loop outermost: consecutive MFMAs go to different accumulators (a dependent MFMA would wait out the full
      // latency of its predecessor); smallest cross terms first
      constexpr int NTERM = TERMS == 3 ? 6 : 1;
      constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < NTERM; ++t) {
        const int pa_ = TERMS == 3 ? TA[t] : 0, pb_ = TERMS == 3 ? TB[t] : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa_][i], b[pb_][j], acc[i][j], 0, 0, 0);
      }
    }
    if (NBUF == 2) {
      if (more) store(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();  // every wave is done reading the stage before it is overwritten
      if (more) store(0);
      __syncthreads();
    }
  }
  uint64_t key = 0;
  if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
  const bool to_slab = d.splitk > 1;
  float* slab = to_slab ? d.workspace + ((size_t)blockIdx.y * max(1, d_in.batch) + (d_in.batch > 1 ? blockIdx.z : 0)) * d.M * d.N : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        if (to_slab) slab[(size_t)row * d.N + col] = acc[i][j][r];
        else epilogue_store(d, key, row, col, acc[i][j][r]);
      }
    }
}

// Deterministic split-K combine: slabs [splitk][M][N] -> epilogue(C).  One thread per 4 consecutive
// columns (16-byte loads), 4 independent partial sums so the slab loads pipeline.
// MMFN_EPI_COLSUM_A with split-K: colsum[m] = sum over the slices (in slice order) of the partial rows behind the slabs
__device__ __forceinline__ void combine_colsum(const mmfn_gemm_desc& d) {
  if (!(d.flags & MMFN_EPI_COLSUM_A)) return;
  const float* part = d.workspace + (size_t)d.splitk * d.M * d.N;
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < d.M; m += gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int z = 0; z < d.splitk; ++z) v += part[(size_t)z * d.M + m];
    d.colsum[m] = v;
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const mmfn_gemm_desc d_in) {
  // batch > 1 here: outputs of the batch entries are NOT packed (strideC != M*N, e.g. the same weight of eight transformer
  // blocks in the flat gradient buffer): blockIdx.y is the batch entry, slabs are [split][batch][M][N]
  mmfn_gemm_desc d = d_in;
  const size_t per = (size_t)d.M * d.N;
  const size_t total = per * (size_t)max(1, d_in.batch);   // one slab
  if (d_in.batch > 1) {
    d.C += (size_t)blockIdx.y * d_in.strideC;
    d.workspace += (size_t)blockIdx.y * per;
  }
  const size_t total4 = per >> 2;
  if (d_in.batch <= 1 && blockIdx.y == 0) combine_colsum(d);
  uint64_t key = 0;
  if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
  const bool vec = (d.N & 3) == 0;
  if (vec) {
    for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      const float* p = d.workspace + i4 * 4;
      int z = 0;
      for (; z + 3 < d.splitk; z += 4) {
        const f32x4 a = ld4(p + (size_t)z * total), b = ld4(p + (size_t)(z + 1) * total);
        const f32x4 c = ld4(p + (size_t)(z + 2) * total), e = ld4(p + (size_t)(z + 3) * total);
#pragma unroll
        for (int q = 0; q < 4; ++q) { s0[q] += a[q]; s1[q] += b[q]; s2[q] += c[q]; s3[q] += e[q]; }
      }
      for (; z < d.splitk; ++z) {
        const f32x4 a = ld4(p + (size_t)z * total);
#pragma unroll
        for (int q = 0; q < 4; ++q) s0[q] += a[q];
      }
      const size_t idx = i4 * 4;
      const int row = (int)(idx / d.N), col = (int)(idx - (size_t)row * d.N);
#pragma unroll
      for (int q = 0; q < 4; ++q) epilogue_store(d, key, row, col + q, (s0[q] + s1[q]) + (s2[q] + s3[q]));
    }
  } else {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (size_t)gridDim.x * blockDim.x) {
      float v = 0.0f;
      for (int z = 0; z < d.splitk; ++z) v += d.workspace[(size_t)z * total + idx];
      const int row = (int)(idx / d.N), col = (int)(idx - (size_t)row * d.N);
      epilogue_store(d, key, row, col, v);
    }
  }
}

// Deep splits (first-layer weight gradients: K = B*OH*OW ~ 5e5, a handful of output tiles, 100+ slabs): 64 consecutive
// elements per block, 16 waves each summing one z-chunk with coalesced 256-byte reads, then a fixed-order LDS combine.
// (The flat kernel above walks all slabs serially per thread: 600 us for the 2-channel LiDAR stem.)
__global__ __launch_bounds__(1024) void splitk_reduce_deep_kernel(const mmfn_gemm_desc d) {
  __shared__ float part[16][64];
  combine_colsum(d);
  const size_t total = (size_t)d.M * d.N;
  const int e = threadIdx.x & 63, zc = threadIdx.x >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + e;
  const int per = (d.splitk + 15) / 16;
  const int z0 = zc * per, z1 = min(d.splitk, z0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < total) {
    const float* p = d.workspace + idx;
    int z = z0;
    for (; z + 3 < z1; z += 4) {
      s0 += p[(size_t)z * total];
      s1 += p[(size_t)(z + 1) * total];
      s2 += p[(size_t)(z + 2) * total];
      s3 += p[(size_t)(z + 3) * total];
    }
    for (; z < z1; ++z) s0 += p[(size_t)z * total];
  }
  part[zc][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zc == 0 && idx < total) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][e];
    uint64_t key = 0;
    if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
    const int row = (int)(idx / d.N), col = (int)(idx - (size_t)row * d.N);
    epilogue_store(d, key, row, col, v);
  }
}

// Batched launches may split K too when the batch's outputs are packed back to back (strideC == M*N, ldc == N) and the
// epilogue has no per-batch operand: the slabs are then [split][batch][M][N] and the combine kernel sees one (batch*M) x N matrix.
// ... or, when the outputs are strided (the same weight gradient of several transformer blocks, each in its own place of the
// flat gradient buffer), the combine kernel takes the batch entry from blockIdx.y.
bool batch_packed(const mmfn_gemm_desc& d) { return d.strideC == (int64_t)d.M * d.N && d.ldc == d.N; }
bool batch_can_split(const mmfn_gemm_desc& d) {
  return d.batch > 1 && !(d.flags & (MMFN_EPI_RESIDUAL | MMFN_EPI_MASK_AUX | MMFN_EPI_ACCUM)) &&
         (batch_packed(d) || !(d.flags & MMFN_EPI_DROPOUT));
}

void launch_splitk_reduce(const mmfn_gemm_desc& dd_in, hipStream_t s) {
  mmfn_gemm_desc dd = dd_in;
  if (dd.batch > 1 && !batch_packed(dd)) {  // strided outputs: one grid row per batch entry
    const size_t per = (size_t)dd.M * dd.N;
    const bool vec = (dd.N & 3) == 0;
    const size_t work = vec ? per / 4 : per;
    const int blocks = (int)std::min<size_t>((work + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, dd.batch), dim3(256), 0, s, dd);
    return;
  }
  if (dd.batch > 1) {  // packed outputs (only reached when batch_can_split): one (batch*M) x N matrix
    dd.M *= dd.batch;
    dd.batch = 1;
  }
  const size_t total = (size_t)dd.M * dd.N;
  if (dd.splitk >= 32 && total <= ((size_t)1 << 20)) {
    hipLaunchKernelGGL(splitk_reduce_deep_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, s, dd);
    return;
  }
  const bool vec = (dd.N & 3) == 0;
  const size_t work = vec ? total / 4 : total;
  const int blocks = (int)std::min<size_t>((work + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, dd);
}

struct TileCand { int id, bm, bn; float eff; int target; };
// id: 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128 (2x2 waves each); 5 = 192x64, 6 = 64x192 (2x2 waves, wave tile 96x32 /
// 32x96), 7 = 64x64 as two waves of 32x64.  eff = measured relative MFMA efficiency of the tile shape; target = resident blocks
// that saturate the chip (256 CUs x blocks/CU that fit).  5-7 are only taken on request (mmfn_gemm_desc.tile, i.e. from the
// measured table tuning/gfx950.json), never by the time model below.
const TileCand kTiles[7] = {{1, 128, 128, 1.00f, 512}, {3, 128, 64, 1.00f, 512}, {4, 64, 128, 1.00f, 512}, {2, 64, 64, 0.98f, 768},
                            {5, 192, 64, 1.00f, 512}, {6, 64, 192, 1.00f, 512}, {7, 64, 64, 0.98f, 1536}};
constexpr int kMaxTile = 7;
// X(id, BM, BN, waves along M, waves along N)
#define MMFN_F32_TILES(X) X(1, 128, 128, 2, 2) X(3, 128, 64, 2, 2) X(4, 64, 128, 2, 2) X(5, 192, 64, 2, 2) X(6, 64, 192, 2, 2) \
  X(7, 64, 64, 2, 1) X(2, 64, 64, 2, 2)

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

bool fast_ok(const mmfn_gemm_desc& d) {
  if (d.K % BK) return false;
  if ((((uintptr_t)d.A) | ((uintptr_t)d.B)) & 15) return false;
  switch (d.a_mode) {
    case MMFN_A_ROWMAJOR: if (d.lda & 3) return false; break;
    case MMFN_A_COLMAJOR: if ((d.lda & 3) || (d.M & 3) || d.M < 4) return false; break;
    case MMFN_A_IM2COL: if (d.Cin % BK) return false; break;
    case MMFN_A_DGRAD: if (d.Cout % BK) return false; break;
  }
  switch (d.b_mode) {
    case MMFN_B_NK: if (d.ldb & 3) return false; break;
    case MMFN_B_KN: if ((d.ldb & 3) || (d.N & 3) || d.N < 4) return false; break;
    case MMFN_B_DGRADW: if ((d.Cin & 3) || (d.N & 3) || d.N < 4 || d.Cout % BK) return false; break;
    case MMFN_B_IM2COL:
      if ((d.Cin & 3) || (d.N & 3) || d.N < 4 || ilog2_exact(d.OW) < 0 || ilog2_exact(d.OH * d.OW) < 0) return false;
      break;
  }
  return true;
}

// experiment builds (-DMMFN_GEMM_EXPERIMENTS, tools/experiments/cap_sweep.py): extra dynamic LDS per block caps the blocks per CU
#ifdef MMFN_GEMM_EXPERIMENTS
int dyn_lds_bytes() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MMFN_GEMM_DYN_LDS"); v = e ? atoi(e) : 0; }
  return v;
}
#else
constexpr int dyn_lds_bytes() { return 0; }
#endif

template <int AM, int BMODE>
int launch_form(const mmfn_gemm_desc& d, int tile, int splitk, hipStream_t s) {
  const int nkt = ceil_div(d.K, BK);
  const int kps = ceil_div(nkt, splitk);
  const int zdim = ceil_div(nkt, kps);
  mmfn_gemm_desc dd = d;
  dd.splitk = zdim;
  if (fast_ok(d)) {
    if (AM == MMFN_A_DGRAD && d.stride == 2 && !(d.H & 1) && !(d.W & 1) && d.batch <= 1) {
      // four output-parity classes in one launch (grid.z), no split-K: each class already has M/4 rows
      dd.dg_parity = 1;
      dd.splitk = 1;
      if (tile > 4) tile = 2;   // (the parity form is instantiated for the four 2x2-wave tiles)
      const int mloc = d.M / 4;
      const int bm = (tile == 1 || tile == 3) ? 128 : 64, bn = (tile == 1 || tile == 4) ? 128 : 64;
      const int tn = ceil_div(d.N, bn);
      dim3 grid(ceil_div(mloc, bm) * tn, 1, 4);
      if (tile == 1) hipLaunchKernelGGL((gemm_f32_fast_kernel<AM, BMODE, 128, 128>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
      else if (tile == 3) hipLaunchKernelGGL((gemm_f32_fast_kernel<AM, BMODE, 128, 64>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
      else if (tile == 4) hipLaunchKernelGGL((gemm_f32_fast_kernel<AM, BMODE, 64, 128>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
      else hipLaunchKernelGGL((gemm_f32_fast_kernel<AM, BMODE, 64, 64>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
      MMFN_LAUNCH_CHECK();
      return 0;
    }
    if (d.flags & MMFN_EPI_LN_FOLD) {
      if (AM != MMFN_A_ROWMAJOR || BMODE != MMFN_B_NK) return MMFN_EINVAL;
      if (tile > 4) tile = 2;
      int bm = (tile == 1 || tile == 3) ? 128 : 64, bn = (tile == 1 || tile == 4) ? 128 : 64;
      if (d.M % bm || d.N % bn) { tile = 2; bm = bn = 64; }   // the epilogue of this form only exists for interior tiles
      if (d.M % bm || d.N % bn || d.batch > 1) return MMFN_EINVAL;
      dd.splitk = 1;
      const int tn = d.N / bn;
      dim3 grid((d.M / bm) * tn, 1, 1);
      if (AM == MMFN_A_ROWMAJOR && BMODE == MMFN_B_NK) {   // (only this form instantiates the LNF kernels)
        if (tile == 1) hipLaunchKernelGGL((gemm_f32_fast_kernel<MMFN_A_ROWMAJOR, MMFN_B_NK, 128, 128, true>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
        else if (tile == 3) hipLaunchKernelGGL((gemm_f32_fast_kernel<MMFN_A_ROWMAJOR, MMFN_B_NK, 128, 64, true>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
        else if (tile == 4) hipLaunchKernelGGL((gemm_f32_fast_kernel<MMFN_A_ROWMAJOR, MMFN_B_NK, 64, 128, true>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
        else hipLaunchKernelGGL((gemm_f32_fast_kernel<MMFN_A_ROWMAJOR, MMFN_B_NK, 64, 64, true>), grid, dim3(NT), 0, s, dd, 1 << 20, tn, 0, 0);
      }
      MMFN_LAUNCH_CHECK();
      return 0;
    }
    const int l_ow = (BMODE == MMFN_B_IM2COL) ? ilog2_exact(d.OW) : 0;
    const int l_ohw = (BMODE == MMFN_B_IM2COL) ? ilog2_exact(d.OH * d.OW) : 0;
#define MMFN_LAUNCH_FAST(ID_, BM_, BN_, WM_, WN_)                                                                     \
  if (tile == ID_ && !launched) {                                                                                    \
    launched = true;                                                                                                 \
    const int tn = ceil_div(d.N, BN_);                                                                               \
    dim3 grid(ceil_div(d.M, BM_) * tn, zdim, d.batch > 1 ? d.batch : 1);                                             \
    hipLaunchKernelGGL((gemm_f32_fast_kernel<AM, BMODE, BM_, BN_, false, WM_, WN_>), grid, dim3(64 * WM_ * WN_), dyn_lds_bytes(), s, \
                       dd, kps, tn, l_ow, l_ohw);                                                                    \
  }
    bool launched = false;
    if (tile < 1 || tile > kMaxTile) tile = 2;
    MMFN_F32_TILES(MMFN_LAUNCH_FAST)
#undef MMFN_LAUNCH_FAST
    MMFN_LAUNCH_CHECK();
    if (zdim > 1 && !(d.flags & MMFN_EPI_KEEP_SLABS)) {
      launch_splitk_reduce(dd, s);
      MMFN_LAUNCH_CHECK();
    }
    return 0;
  }
  if (d.flags & MMFN_EPI_LN_FOLD) return MMFN_EINVAL;   // (needs the fast kernel: 16-byte aligned operands, K a multiple of 16)
  if (tile > 4) tile = 2;
#define MMFN_LAUNCH_TILE(BM_, BN_)                                                                              \
  {                                                                                                             \
    const int tn = ceil_div(d.N, BN_);                                                                          \
    dim3 grid(ceil_div(d.M, BM_) * tn, zdim, d.batch > 1 ? d.batch : 1);                                        \
    hipLaunchKernelGGL((gemm_f32_kernel<AM, BMODE, BM_, BN_>), grid, dim3(NT), 0, s, dd, kps, tn);              \
  }
  if (tile == 1) MMFN_LAUNCH_TILE(128, 128)
  else if (tile == 3) MMFN_LAUNCH_TILE(128, 64)
  else if (tile == 4) MMFN_LAUNCH_TILE(64, 128)
  else MMFN_LAUNCH_TILE(64, 64)
#undef MMFN_LAUNCH_TILE
  MMFN_LAUNCH_CHECK();
  if (zdim > 1 && !(d.flags & MMFN_EPI_KEEP_SLABS)) {
    launch_splitk_reduce(dd, s);
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}

int bf16_conv_form(const mmfn_gemm_desc& d) {   // 1: im2col A (fwd / flipped dgrad), 2: im2col B (wgrad), 0: not a covered convolution
  if (d.flags & MMFN_EPI_BF16X3) return 0;
  if (d.a_mode == MMFN_A_IM2COL && d.b_mode == MMFN_B_NK && d.Cin % HBK == 0 && d.K % HBK == 0 && !(d.ldb & 3) && d.batch <= 1)
    return 1;
  if (d.a_mode == MMFN_A_COLMAJOR && d.b_mode == MMFN_B_IM2COL && d.Cin % 4 == 0 && d.K % HBK == 0 && !(d.lda & 3) && !(d.M & 3) &&
      d.M >= 4 && !(d.N & 3) && d.N >= 4 && d.batch <= 1 && ilog2_exact(d.OW) >= 0 && ilog2_exact(d.OH * d.OW) >= 0)
    return 2;
  return 0;
}

bool bf16_ok(const mmfn_gemm_desc& d) {
  if (!(d.flags & (MMFN_EPI_BF16_OPERANDS | MMFN_EPI_BF16X3))) return false;
  if ((((uintptr_t)d.A) | ((uintptr_t)d.B)) & 15) return false;
  if (bf16_conv_form(d)) return true;
  const bool a_kc = d.a_mode == MMFN_A_ROWMAJOR, a_mc = d.a_mode == MMFN_A_COLMAJOR;
  const bool b_kc = d.b_mode == MMFN_B_NK, b_mc = d.b_mode == MMFN_B_KN;
  if (!(a_kc || a_mc) || !(b_kc || b_mc)) return false;
  if (d.K % HBK || (d.lda & 3) || (d.ldb & 3)) return false;
  if ((((uintptr_t)d.A) | ((uintptr_t)d.B)) & 15) return false;
  if (a_mc && ((d.M & 3) || d.M < 4)) return false;
  if (b_mc && ((d.N & 3) || d.N < 4)) return false;
  return true;
}

// Tile and split-K of the bf16 kernel: 128x128 tiles (the efficient shape) as long as they can give every CU a block, if
// need be by splitting K (never below 8 k-tiles = 256 of K per split); 64x64 tiles only when even that leaves CUs idle
// (short K, or batched launches, which do not split).  Then split until there are about two blocks per CU.
void bf16_config(const mmfn_gemm_desc& d, int* bt, int* splitk) {
  const int64_t nb = std::max(1, d.batch);
  const int nkt = d.K / HBK;
  const bool can_split = (d.batch <= 1 || batch_can_split(d)) && d.splitk != 1 && d.workspace != nullptr;
  const int64_t max_sk = can_split ? std::max(1, std::min(32, nkt / 8)) : 1;
  const int64_t big = (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128) * nb;
  *bt = (d.tile == 1 || (d.tile != 2 && big * max_sk >= 256)) ? 128 : 64;
  const int64_t blocks = (int64_t)ceil_div(d.M, *bt) * ceil_div(d.N, *bt) * nb;
  int sk = 1;
  if (can_split) {
    if (d.splitk > 1) sk = std::min(d.splitk, nkt);
    else if (blocks < 384) sk = (int)std::max<int64_t>(1, std::min<int64_t>((512 + blocks - 1) / blocks, max_sk));
  }
  *splitk = sk;
}

int launch_bf16(const mmfn_gemm_desc& d, hipStream_t s) {
  const int nkt = d.K / HBK;
  int bt, sk;
  bf16_config(d, &bt, &sk);
  const int kps = ceil_div(nkt, sk), zdim = ceil_div(nkt, kps);
  mmfn_gemm_desc dd = d;
  dd.splitk = zdim;
  const int tn = ceil_div(d.N, bt);
  dim3 grid(ceil_div(d.M, bt) * tn, zdim, d.batch > 1 ? d.batch : 1);
  const int conv = bf16_conv_form(d);
  if (conv) {
    const int l_ow = conv == 2 ? ilog2_exact(d.OW) : 0, l_ohw = conv == 2 ? ilog2_exact(d.OH * d.OW) : 0;
    if (conv == 1) {
      if (bt == 128) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 128, 128, 1, 1>), grid, dim3(NT), 0, s, dd, kps, tn, 0, 0);
      else hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 64, 64, 1, 1>), grid, dim3(NT), 0, s, dd, kps, tn, 0, 0);
    } else {
      if (bt == 128) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 128, 128, 1, 2>), grid, dim3(NT), 0, s, dd, kps, tn, l_ow, l_ohw);
      else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 64, 64, 1, 2>), grid, dim3(NT), 0, s, dd, kps, tn, l_ow, l_ohw);
    }
    MMFN_LAUNCH_CHECK();
    if (zdim > 1) {
      launch_splitk_reduce(dd, s);
      MMFN_LAUNCH_CHECK();
    }
    return 0;
  }
  const bool a_kc = d.a_mode == MMFN_A_ROWMAJOR, b_kc = d.b_mode == MMFN_B_NK;
  const bool x3 = (d.flags & MMFN_EPI_BF16X3) != 0;
#define MMFN_LAUNCH_BF16(AK, BK_, T)                                                                                \
  {                                                                                                                 \
    if (x3) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, T, T, 3>), grid, dim3(NT), 0, s, dd, kps, tn);             \
    else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, T, T, 1>), grid, dim3(NT), 0, s, dd, kps, tn);                \
  }
  if (bt == 128) {
    if (a_kc && b_kc) MMFN_LAUNCH_BF16(true, true, 128)
    else if (a_kc) MMFN_LAUNCH_BF16(true, false, 128)
    else if (b_kc) MMFN_LAUNCH_BF16(false, true, 128)
    else MMFN_LAUNCH_BF16(false, false, 128)
  } else {
    if (a_kc && b_kc) MMFN_LAUNCH_BF16(true, true, 64)
    else if (a_kc) MMFN_LAUNCH_BF16(true, false, 64)
    else if (b_kc) MMFN_LAUNCH_BF16(false, true, 64)
    else MMFN_LAUNCH_BF16(false, false, 64)
  }
#undef MMFN_LAUNCH_BF16
  MMFN_LAUNCH_CHECK();
  if (zdim > 1) {
    launch_splitk_reduce(dd, s);
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}

void pick_config(const mmfn_gemm_desc& d, int* tile, int* splitk) {
  // Time model per candidate tile (microseconds, constants fitted to tools/gemm_bench.py on MI355X):
  //   compute  = padded FLOPs / (95 TF/s * tile efficiency * fill), fill = min(1, blocks*sk / saturating blocks)
  //   split-K  = slab write + slab read at ~3 TB/s + one extra launch (~3 us)
  // and the split factor is the smallest one that saturates the chip (capped so slabs stay small).
  const int nkt = ceil_div(d.K, BK);
  const bool can_split = d.workspace != nullptr && d.splitk != 1 && (d.batch <= 1 || batch_can_split(d));
  int best = -1, best_sk = 1;
  double best_t = 0.0;
  const bool asked = d.tile >= 1 && d.tile <= kMaxTile;
  for (int i = 0; i < kMaxTile; ++i) {
    const TileCand& c = kTiles[i];
    if (asked ? d.tile != c.id : c.id > 4) continue;
    const int64_t tm = ceil_div(d.M, c.bm), tn = ceil_div(d.N, c.bn);
    const int64_t blocks = tm * tn * std::max(1, d.batch);
    // outputs of only a handful of tiles (first-layer weight gradients, K = B*H*W ~ 1e5..1e6) may split deeper
    const int sk_cap = (int)std::max<int64_t>(48, std::min<int64_t>(512, 1024 / blocks));
    const int sk_max = can_split ? std::min(sk_cap, std::max(1, nkt / 8)) : 1;
    int sk = 1;
    if (d.splitk > 1) sk = std::min(d.splitk, std::max(1, nkt));
    else if (d.splitk < 1 && blocks < c.target) sk = (int)std::min<int64_t>((c.target + blocks - 1) / blocks, sk_max);
    if (!can_split) sk = 1;
    const double flops = 2.0 * (double)(tm * c.bm) * (double)(tn * c.bn) * (double)d.K;
    const double fill = std::min(1.0, (double)(blocks * sk) / (double)c.target);
    double t = flops / (95e6 * c.eff * fill);  // us
    if (sk > 1) t += (double)sk * (double)d.M * (double)d.N * 8.0 / 3e6 + 3.0;
    if (best < 0 || t < best_t) { best = i; best_t = t; best_sk = sk; }
  }
  *tile = kTiles[best].id;
  *splitk = best_sk;
}

}  // namespace

#ifdef MMFN_GEMM_TIMELINE
// experiment builds only: buf = device memory for blocks * waves * 48 uint64 (zeroed by the caller)
extern "C" int mmfn_debug_gemm_timeline(void* buf, int blocks) {
  unsigned long long* p = (unsigned long long*)buf;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &p, sizeof(p)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_blocks), &blocks, sizeof(blocks)) != hipSuccess) return 1;
  return 0;
}
#endif

extern "C" int64_t mmfn_gemm_workspace_bytes(const mmfn_gemm_desc* d) {
  if (!d) return 0;
  mmfn_gemm_desc dd = *d;
  float dummy;
  dd.workspace = &dummy;  // let pick_config consider split-K
  int tile, sk;
  if (bf16_ok(dd)) bf16_config(dd, &tile, &sk);
  else pick_config(dd, &tile, &sk);
  const int64_t cs = (d->flags & MMFN_EPI_COLSUM_A) ? (int64_t)d->M : 0;   // one partial row per k-slice behind the slabs
  return sk > 1 ? (int64_t)sk * (std::max(1, d->batch) * (int64_t)d->M * d->N + cs) * (int64_t)sizeof(float) : 0;
}

extern "C" int mmfn_gemm_f32_splits(const mmfn_gemm_desc* d) {
  if (!d || d->K <= 0) return 1;
  if (bf16_ok(*d)) return 1;   // (the bf16-operand kernels always combine)
  int tile, sk;
  pick_config(*d, &tile, &sk);
  const int nkt = ceil_div(d->K, BK);
  return ceil_div(nkt, ceil_div(nkt, std::max(1, sk)));
}

extern "C" int mmfn_gemm_f32(const mmfn_gemm_desc* dp, void* stream) {
  if (!dp) return MMFN_EINVAL;
  const mmfn_gemm_desc& d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_KEEP_SLABS) && ((d.flags & ~MMFN_EPI_KEEP_SLABS) || (d.batch > 1 && !batch_packed(d)) || bf16_ok(d)))
    return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_DROPOUT) && (!d.rng_state || d.drop_p < 0.f || d.drop_p >= 1.f)) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_BIAS) && !d.bias) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_RESIDUAL) && !d.res) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_MASK_AUX) && !d.aux) return MMFN_EINVAL;
  if (d.flags & MMFN_EPI_LN_FOLD) {
    if (!(d.flags & MMFN_EPI_BIAS) || !d.ln_c1 || (d.flags & (MMFN_EPI_BF16_OPERANDS | MMFN_EPI_BF16X3 | MMFN_EPI_GELU | MMFN_EPI_ACCUM)) ||
        d.a_mode != MMFN_A_ROWMAJOR || d.b_mode != MMFN_B_NK || (d.ln_mean && !d.ln_rstd) || d.ln_eps <= 0.f)
      return MMFN_EINVAL;
  }
  if (d.flags & MMFN_EPI_COLSUM_A) {   // only the fast TN kernel forms the sums
    if (!d.colsum || d.a_mode != MMFN_A_COLMAJOR || d.b_mode != MMFN_B_KN || d.batch > 1 || !fast_ok(d) ||
        (d.flags & (MMFN_EPI_BF16_OPERANDS | MMFN_EPI_BF16X3)))
      return MMFN_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  if (bf16_ok(d)) return launch_bf16(d, s);
  int tile, sk;
  pick_config(d, &tile, &sk);
  const int a = d.a_mode, b = d.b_mode;
  if (a == MMFN_A_ROWMAJOR && b == MMFN_B_NK) return launch_form<MMFN_A_ROWMAJOR, MMFN_B_NK>(d, tile, sk, s);
  if (a == MMFN_A_ROWMAJOR && b == MMFN_B_KN) return launch_form<MMFN_A_ROWMAJOR, MMFN_B_KN>(d, tile, sk, s);
  if (a == MMFN_A_COLMAJOR && b == MMFN_B_KN) return launch_form<MMFN_A_COLMAJOR, MMFN_B_KN>(d, tile, sk, s);
  if (a == MMFN_A_IM2COL && b == MMFN_B_NK) return launch_form<MMFN_A_IM2COL, MMFN_B_NK>(d, tile, sk, s);
  if (a == MMFN_A_DGRAD && b == MMFN_B_DGRADW) return launch_form<MMFN_A_DGRAD, MMFN_B_DGRADW>(d, tile, sk, s);
  if (a == MMFN_A_COLMAJOR && b == MMFN_B_IM2COL) return launch_form<MMFN_A_COLMAJOR, MMFN_B_IM2COL>(d, tile, sk, s);
  return MMFN_EINVAL;
}
