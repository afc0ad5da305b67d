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

#include "common.h"

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

struct ConvPos {  // decoded position of a GEMM row in conv space
  int b, y0, x0;
  bool ok;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

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
  *c = v;
}

template <int AM, int BMODE, int BM, int BN>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(const mmfn_gemm_desc d, const int kt_per_split,
                                                      const int tiles_n) {
  constexpr bool A_KC = (AM != MMFN_A_COLMAJOR);
  constexpr bool B_KC = (BMODE == MMFN_B_NK);
  constexpr int WAVES_M = 2, WAVES_N = 2;
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int LDK = BK + 4;
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
      const int m = m0 + ((tid + i * NT) >> 2);
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
      const int m = m0 + (u >> 2);
      const int k = kt * BK + (u & 3) * 4;
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
      const int n = n0 + (u >> 2);
      const int k = kt * BK + (u & 3) * 4;
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
    if (A_KC) *reinterpret_cast<f32x4*>(&As[(u >> 2) * LDK + (u & 3) * 4]) = v;
    else *reinterpret_cast<f32x4*>(&As[(u / (BM / 4)) * BM + (u % (BM / 4)) * 4]) = v;
  };
  auto store_b = [&](float* Bs, int i, f32x4 v) {
    const int u = tid + i * NT;
    if (B_KC) *reinterpret_cast<f32x4*>(&Bs[(u >> 2) * LDK + (u & 3) * 4]) = v;
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
          const f32x4 t = *reinterpret_cast<const f32x4*>(&As[row * LDK + c * 8 + h * 4]);
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
          const f32x4 t = *reinterpret_cast<const f32x4*>(&Bs[col * LDK + c * 8 + h * 4]);
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
  float* slab = to_slab ? d.workspace + (size_t)blockIdx.y * d.M * d.N : nullptr;
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

__global__ void splitk_reduce_kernel(const mmfn_gemm_desc d) {
  const size_t total = (size_t)d.M * d.N;
  uint64_t key = 0;
  if (d.flags & MMFN_EPI_DROPOUT) key = mmfn_rng_key(d.rng_state, d.rng_stream);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float v = 0.0f;
    for (int z = 0; z < d.splitk; ++z) v += d.workspace[(size_t)z * total + idx];
    const int row = (int)(idx / d.N), col = (int)(idx - (size_t)row * d.N);
    epilogue_store(d, key, row, col, v);
  }
}

struct TileCand { int id, bm, bn; float eff; int target; };
// id: 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128.  eff = measured relative MFMA efficiency of the
// tile shape; target = resident blocks that saturate the chip (256 CUs x blocks/CU that fit).
const TileCand kTiles[4] = {{1, 128, 128, 1.00f, 512}, {3, 128, 64, 0.93f, 768}, {4, 64, 128, 0.93f, 768}, {2, 64, 64, 0.85f, 1024}};

template <int AM, int BMODE>
int launch_form(const mmfn_gemm_desc& d, int tile, int splitk, hipStream_t s) {
  const int nkt = ceil_div(d.K, BK);
  const int kps = ceil_div(nkt, splitk);
  const int zdim = ceil_div(nkt, kps);
  mmfn_gemm_desc dd = d;
  dd.splitk = zdim;
#define MMFN_LAUNCH_TILE(BM_, BN_)                                                                              \
  {                                                                                                             \
    const int tn = ceil_div(d.N, BN_);                                                                          \
    dim3 grid(ceil_div(d.M, BM_) * tn, zdim);                                                                   \
    hipLaunchKernelGGL((gemm_f32_kernel<AM, BMODE, BM_, BN_>), grid, dim3(NT), 0, s, dd, kps, tn);              \
  }
  if (tile == 1) MMFN_LAUNCH_TILE(128, 128)
  else if (tile == 3) MMFN_LAUNCH_TILE(128, 64)
  else if (tile == 4) MMFN_LAUNCH_TILE(64, 128)
  else MMFN_LAUNCH_TILE(64, 64)
#undef MMFN_LAUNCH_TILE
  MMFN_LAUNCH_CHECK();
  if (zdim > 1) {
    const size_t total = (size_t)d.M * d.N;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, dd);
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}

void pick_config(const mmfn_gemm_desc& d, int* tile, int* splitk) {
  const int nkt = ceil_div(d.K, BK);
  const bool can_split = d.workspace != nullptr && d.splitk != 1;
  const int sk_max = can_split ? std::max(1, nkt / 8) : 1;
  int best = -1;
  float best_cost = 0.f;
  for (int i = 0; i < 4; ++i) {
    const TileCand& c = kTiles[i];
    if (d.tile >= 1 && d.tile <= 4 && d.tile != c.id) continue;
    const int64_t tm = ceil_div(d.M, c.bm), tn = ceil_div(d.N, c.bn);
    const float waste = (float)(tm * c.bm * tn * c.bn) / ((float)d.M * (float)d.N);
    const int64_t par = tm * tn * sk_max;
    const float under = par >= c.target ? 1.0f : (float)c.target / (float)par;
    // split-K is not free (slab round trip + reduce launch): prefer shapes that fill the chip unsplit
    const float split_pen = (tm * tn >= c.target / 2) ? 1.0f : 1.05f;
    const float cost = waste / c.eff * under * split_pen;
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  const TileCand& c = kTiles[best];
  const int64_t blocks = (int64_t)ceil_div(d.M, c.bm) * ceil_div(d.N, c.bn);
  int sk = d.splitk;
  if (sk < 1) {
    sk = 1;
    if (blocks < c.target / 2) sk = (int)std::min<int64_t>((c.target + blocks - 1) / blocks, sk_max);
    if (sk > 256) sk = 256;
  }
  if (!can_split) sk = 1;
  if (sk > nkt) sk = std::max(1, nkt);
  *tile = c.id;
  *splitk = sk;
}

}  // namespace

extern "C" int64_t mmfn_gemm_workspace_bytes(const mmfn_gemm_desc* d) {
  if (!d) return 0;
  mmfn_gemm_desc dd = *d;
  float dummy;
  dd.workspace = &dummy;  // let pick_config consider split-K
  int tile, sk;
  pick_config(dd, &tile, &sk);
  return sk > 1 ? (int64_t)sk * d->M * d->N * (int64_t)sizeof(float) : 0;
}

extern "C" int mmfn_gemm_f32(const mmfn_gemm_desc* dp, void* stream) {
  if (!dp) return MMFN_EINVAL;
  const mmfn_gemm_desc& d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_DROPOUT) && (!d.rng_state || d.drop_p < 0.f || d.drop_p >= 1.f)) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_BIAS) && !d.bias) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_RESIDUAL) && !d.res) return MMFN_EINVAL;
  if ((d.flags & MMFN_EPI_MASK_AUX) && !d.aux) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
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
