// BatchNorm2d (training statistics, apply, backward) and LayerNorm (fwd/bwd) for [rows, C]
// channels-last matrices.  HBM-bound: every kernel streams its operands once with 16-byte
// loads; statistics are accumulated in fp64 so the fp32 result matches a double-accumulating
// CPU reference to the last bits that matter for the 1e-4 loss tolerance.
//
// Reference semantics restated: BatchNorm2d eps 1e-5, momentum 0.1, biased
// variance for normalisation, unbiased for the running update (model_vec.py:510,516 and every
// torchvision BasicBlock); LayerNorm eps 1e-5 biased variance (model_vec.py:117-118,162,252).
#include <algorithm>

#include "common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ column statistics
// MODE 0: s1 = sum x,        s2 = sum x^2                      (BN forward statistics)
// MODE 1: s1 = sum ge,       s2 = sum ge * xhat                (BN backward reductions)
//         ge = g * (y > 0) when y != null; g * (bn(x) > 0) when y == null and zb != null (the BatchNorm + ReLU output was never
//         written: its sign is recomputed from x with the forward's own expression, zw / zb = the BatchNorm's weight / bias);
//         else g;  xhat = (x - mean) * rstd
// TX: element type of x (the convolution output), TA: of the activation-side tensors g, y (fp32 path: both float; bf16 mode:
// both bf16, except the fp32 stems whose convolution output stays fp32)
// MASK (MODE 1): 0 none, 1 from y, 2 recomputed from x - a template parameter so that the loads of an iteration are issued together
// (a run-time test of the y pointer puts the third load behind a branch: two memory round trips per row instead of one)
template <int MODE, int MASK, typename TX, typename TA>
__global__ __launch_bounds__(NT) void col_partial_kernel(const TX* __restrict__ x, const TA* __restrict__ g,
                                                         const TA* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ zw,
                                                         const float* __restrict__ zb, int64_t M, int C,
                                                         int64_t rows_per_block, double* __restrict__ partials) {
  const int cq = C >> 2;           // float4 columns
  const int tid = threadIdx.x;
  const int col4 = tid % cq;       // requires cq <= 256 and 256 % cq == 0
  const int rl = tid / cq;
  const int RL = NT / cq;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  f32x4 mu = {0, 0, 0, 0}, rs = {0, 0, 0, 0}, al = {0, 0, 0, 0}, be = {0, 0, 0, 0};
  constexpr bool zmask = MODE == 1 && MASK == 2;
  if (MODE == 1) {
    mu = *reinterpret_cast<const f32x4*>(mean + col4 * 4);
    rs = *reinterpret_cast<const f32x4*>(rstd + col4 * 4);
    if (zmask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { al[e] = mmfn_bn_alpha(zw[col4 * 4 + e], rs[e]); be[e] = mmfn_bn_beta(zb[col4 * 4 + e], mu[e], al[e]); }
    }
  }
  for (int64_t r = r0 + rl; r < r1; r += RL) {
    const size_t off = (size_t)r * C + col4 * 4;
    const f32x4 xv = ldx4(x + off);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const double v = xv[e]; s1[e] += v; s2[e] += v * v; }
    } else {
      f32x4 gv = ldx4(g + off);
      if (MASK == 1) {
        const f32x4 yv = ldx4(y + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.0f ? gv[e] : 0.0f;
      } else if (MASK == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = mmfn_bn_affine(xv[e], al[e], be[e]) > 0.0f ? gv[e] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mu[e]) * rs[e];
        s1[e] += (double)gv[e];
        s2[e] += (double)gv[e] * (double)xh;
      }
    }
  }
  __shared__ double red[2][NT][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += red[0][k * cq + col4][e]; s2[e] += red[1][k * cq + col4][e]; }
    double* p = partials + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[col4 * 4 + e] = s1[e]; p[C + col4 * 4 + e] = s2[e]; }
  }
}


// Sum `nblk` partial rows of NV value-rows for column c in one pass: 4 columns x 64 row-lanes per block, the loads of all
// NV sums in flight together, then a fixed-order two-stage LDS combine (the serial per-column loop this replaces took
// 50-90 us per launch at nblk ~ 400; the 8 x 32 version with one value-row per pass 5.6-7.5 us - these launches sit on the
// dependent chain of every BatchNorm).  partials: value-row v of partial row b at partials[b * row_stride + v * C + c].
constexpr int FIN_COLS = 4, FIN_LANES = 64;
template <int NV, typename T>
__device__ __forceinline__ void reduce_partials(const T* __restrict__ partials, int nblk, size_t row_stride, int C, int c, bool valid,
                                                double (*sh)[FIN_LANES][FIN_COLS], double* out) {
  const int cl = threadIdx.x % FIN_COLS, rl = threadIdx.x / FIN_COLS;
  double t0[NV], t1[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { t0[v] = 0; t1[v] = 0; }
  if (valid) {
    int b = rl;
    for (; b + FIN_LANES < nblk; b += 2 * FIN_LANES) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        t0[v] += (double)partials[(size_t)b * row_stride + (size_t)v * C + c];
        t1[v] += (double)partials[(size_t)(b + FIN_LANES) * row_stride + (size_t)v * C + c];
      }
    }
    if (b < nblk) {
#pragma unroll
      for (int v = 0; v < NV; ++v) t0[v] += (double)partials[(size_t)b * row_stride + (size_t)v * C + c];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) sh[v][rl][cl] = t0[v] + t1[v];
  __syncthreads();
  if (rl < 8) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      double a = 0;
      for (int k = rl; k < FIN_LANES; k += 8) a += sh[v][k][cl];
      t0[v] = a;
    }
  }
  __syncthreads();
  if (rl < 8) {
#pragma unroll
    for (int v = 0; v < NV; ++v) sh[v][rl][cl] = t0[v];
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double a = 0;
    if (rl == 0)
      for (int k = 0; k < 8; ++k) a += sh[v][k][cl];
    out[v] = a;
  }
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ partials, int nblk, int64_t M, int C, float eps,
                                         float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         int64_t* __restrict__ num_batches_tracked) {
  __shared__ double sh[2][FIN_LANES][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[2];
  reduce_partials<2>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  const double s1 = r[0], s2 = r[1];
  if (!valid || threadIdx.x >= FIN_COLS) return;
  const double mu = s1 / (double)M;
  double var = s2 / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                       float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  rstd[c] = 1.0f / sqrtf(rv[c] + eps);
}

__global__ __launch_bounds__(NT) void bn_fold_kernel(const float* __restrict__ w, int K, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ rm,
                                                     const float* __restrict__ rv, float eps, float* __restrict__ w_out,
                                                     float* __restrict__ b_out) {
  const int co = blockIdx.x;
  const float s = gamma[co] / sqrtf(rv[co] + eps);
  for (int k = threadIdx.x; k < K; k += NT) w_out[(size_t)co * K + k] = w[(size_t)co * K + k] * s;
  if (threadIdx.x == 0) b_out[co] = beta[co] - rm[co] * s;
}

// y = [relu]( x * alpha + beta [+ res] ), alpha = w * rstd, beta = b - mean * alpha
template <bool HAS_RES, bool RELU, typename TX, typename TA>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const TX* __restrict__ x, const TA* __restrict__ res,
                                                      TA* __restrict__ y, int64_t total4, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ w, const float* __restrict__ b) {
  const int cq = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const f32x4 xv = ldx4(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float alpha, beta;
      { alpha = mmfn_bn_alpha(w[c4 + e], rstd[c4 + e]); beta = mmfn_bn_beta(b[c4 + e], mean[c4 + e], alpha); }
      o[e] = mmfn_bn_affine(xv[e], alpha, beta);
    }
    if (HAS_RES) {
      const f32x4 rv = ldx4(res + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += rv[e];
    }
    if (RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.0f);
    }
    stx4(y + i * 4, o);
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partials, int nblk, int64_t M, int C,
                                       float* __restrict__ dweight, float* __restrict__ dbias,
                                       float* __restrict__ means /* [2][C]: s1/M, s2/M */) {
  __shared__ double sh[2][FIN_LANES][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[2];
  reduce_partials<2>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  const double s1 = r[0], s2 = r[1];
  if (!valid || threadIdx.x >= FIN_COLS) return;
  dbias[c] = (float)s1;
  dweight[c] = (float)s2;
  means[c] = (float)(s1 / (double)M);
  means[C + c] = (float)(s2 / (double)M);
}

// dx = w * rstd * (ge - mean(ge) - xhat * mean(ge * xhat));  optionally ge_out = ge
template <int MASK, bool HAS_GE, typename TX, typename TA>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const TA* __restrict__ g, const TA* __restrict__ y,
                                                          const TX* __restrict__ x, TX* __restrict__ dx,
                                                          TA* __restrict__ ge_out, int64_t total4, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, const float* __restrict__ zb,
                                                          const float* __restrict__ means) {
  const int cq = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    f32x4 gv = ldx4(g + i * 4);
    const f32x4 xv = ldx4(x + i * 4);
    if (MASK == 1) {
      const f32x4 yv = ldx4(y + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.0f ? gv[e] : 0.0f;
    } else if (MASK == 2) {   // ReLU mask recomputed from x (see col_partial_kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float alpha, beta;
        { alpha = mmfn_bn_alpha(w[c4 + e], rstd[c4 + e]); beta = mmfn_bn_beta(zb[c4 + e], mean[c4 + e], alpha); }
        gv[e] = mmfn_bn_affine(xv[e], alpha, beta) > 0.0f ? gv[e] : 0.0f;
      }
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float rs = rstd[c4 + e];
      const float xh = (xv[e] - mean[c4 + e]) * rs;
      o[e] = (gv[e] - means[c4 + e] - xh * means[C + c4 + e]) * (w[c4 + e] * rs);
    }
    stx4(dx + i * 4, o);
    if (HAS_GE) stx4(ge_out + i * 4, gv);
  }
}

int bn_grid(int64_t M, int C, int64_t* rows_per_block) {
  // ~4 blocks per CU (these kernels only stream: one block per CU reaches ~2.6 TB/s), at least 32 rows each
  int64_t rpb = std::max<int64_t>(32, ceil_div64(M, 1024));
  *rows_per_block = rpb;
  return (int)ceil_div64(M, rpb);
}

// ------------------------------------------------------------------ LayerNorm
// One wave per row, lane owns columns lane, lane+64, ... (C <= 512 -> <= 8 per lane).
// TA: input element type, TO: output (bf16 mode: the transformers' residual stream stays fp32, the normalised GEMM operand is bf16)
template <int MAXPL, typename TA, typename TO = TA>
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(const TA* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, TO* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int M, int C,
                                                           float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int pl = C >> 6;
  float v[MAXPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i)
    if (i < pl) { v[i] = ldx1(x + (size_t)row * C + lane + 64 * i); s += v[i]; }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i)
    if (i < pl) { const float d = v[i] - mu; q += d * d; }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXPL; ++i)
    if (i < pl) {
      const int c = lane + 64 * i;
      float o = (v[i] - mu) * rs * w[c] + b[c];
      if (act == 1) o = fmaxf(o, 0.f);
      else if (act == 2) o = mmfn_gelu(o);
      stx1(y + (size_t)row * C + c, o);
    }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// One row per wave iteration; lane l owns VW consecutive columns per 64*VW-column chunk (4/8/16-byte accesses),
// NCH chunks per row (C = 64 * VW * NCH).  Blocks are small (8 rows at M = 6144 -> 768 blocks) so the whole chip
// streams rows; the per-block dweight / dbias partial rows are combined by colsum_finalize_kernel in fp64.
// TA: element type of the incoming gradient g and of the dropped copy dxd (GEMM operands); TX: of x, dres and dx (the residual
// stream and its gradient).  fp32 path: both float; bf16 mode: TA = bf16, TX = float in the transformers, bf16 elsewhere.
template <int VW, int NCH, typename TA, typename TX = TA>
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const TA* __restrict__ g, const TX* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const TX* __restrict__ dres, TX* __restrict__ dx,
                                                           float* __restrict__ partials /* [grid][2][C] */, int M,
                                                           int act, int rows_per_block, TA* __restrict__ dxd, float drop_p,
                                                           const uint64_t* __restrict__ rng_state, uint32_t rng_stream,
                                                           int want_sum) {
  typedef float vec __attribute__((ext_vector_type(VW)));
  typedef VecIO<VW, TA> IO;
  typedef VecIO<VW, TX> IOX;
  constexpr int C = 64 * VW * NCH;
  // optional second output dxd = dx * dropout keep-scale(row * C + col): the gradient entering the residual branch whose
  // forward applied that dropout in a GEMM epilogue (same counter RNG index) - saves a separate elementwise pass
  uint64_t dkey = 0;
  float inv_keep = 1.f;
  const bool dropping = dxd && drop_p > 0.f;   // (drop_p == 0: dxd is the plain copy of dx in the operand type)
  if (dropping) { dkey = mmfn_rng_key(rng_state, rng_stream); inv_keep = 1.0f / (1.0f - drop_p); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // want_sum: third partial row = column sums of what leaves in dxd (or in dx when there is no dropped copy): the bias
  // gradient of the Linear whose output gradient this is - saves the separate two-launch column sum
  vec dw[NCH], db[NCH], wv[NCH], bv[NCH], ds[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (i * 64 + lane) * VW;
    wv[i] = *reinterpret_cast<const vec*>(w + c);
    bv[i] = *reinterpret_cast<const vec*>(b + c);
#pragma unroll
    for (int j = 0; j < VW; ++j) { dw[i][j] = 0.f; db[i][j] = 0.f; ds[i][j] = 0.f; }
  }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += NT / 64) {
    const float mu = mean[row], rs = rstd[row];
    vec a[NCH], xh[NCH];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const size_t off = (size_t)row * C + (i * 64 + lane) * VW;
      const vec xv = IOX::ld(x + off);
      vec gg = IO::ld(g + off);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float xhat = (xv[j] - mu) * rs;
        float gj = gg[j];
        if (act != 0) {
          const float pre = xhat * wv[i][j] + bv[i][j];
          if (act == 1) gj = pre > 0.f ? gj : 0.f;
          else gj *= mmfn_gelu_grad(pre);
        }
        dw[i][j] += gj * xhat;
        db[i][j] += gj;
        const float aj = gj * wv[i][j];
        xh[i][j] = xhat;
        a[i][j] = aj;
        c1 += aj;
        c2 += aj * xhat;
      }
    }
    c1 = wave_sum(c1) / (float)C;
    c2 = wave_sum(c2) / (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const size_t off = (size_t)row * C + (i * 64 + lane) * VW;
      vec o;
#pragma unroll
      for (int j = 0; j < VW; ++j) o[j] = rs * (a[i][j] - c1 - xh[i][j] * c2);
      if (dres) o += IOX::ld(dres + off);
      IOX::st(dx + off, o);
      if (dxd) {
        vec od = o;
        if (dropping) {
#pragma unroll
          for (int j = 0; j < VW; ++j) od[j] = o[j] * mmfn_dropout_scale(dkey, (uint64_t)off + j, drop_p, inv_keep);
        }
        IO::st(dxd + off, od);
        o = od;
      }
      if (want_sum) ds[i] += o;
    }
  }
  __shared__ float red[3][NT / 64][C];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      red[0][wave][(i * 64 + lane) * VW + j] = dw[i][j];
      red[1][wave][(i * 64 + lane) * VW + j] = db[i][j];
      red[2][wave][(i * 64 + lane) * VW + j] = ds[i][j];
    }
  __syncthreads();
  const int rows = want_sum ? 3 : 2;
  for (int c = threadIdx.x; c < C; c += NT) {
    float sw = 0.f, sb = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { sw += red[0][k][c]; sb += red[1][k][c]; ss += red[2][k][c]; }
    partials[(size_t)blockIdx.x * rows * C + c] = sw;
    partials[(size_t)blockIdx.x * rows * C + C + c] = sb;
    if (want_sum) partials[(size_t)blockIdx.x * rows * C + 2 * C + c] = ss;
  }
}

__global__ void colsum_finalize_kernel(const float* __restrict__ partials, int nblk, int C, float* __restrict__ out0,
                                       float* __restrict__ out1, float* __restrict__ out2) {
  __shared__ double sh[3][FIN_LANES][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[3] = {0, 0, 0};
  if (out2) reduce_partials<3>(partials, nblk, (size_t)3 * C, C, c, valid, sh, r);
  else if (out1) reduce_partials<2>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  else reduce_partials<1>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  const double s0 = r[0], s1 = r[1], s2 = r[2];
  if (!valid || threadIdx.x >= FIN_COLS) return;
  out0[c] = (float)s0;
  if (out1) out1[c] = (float)s1;
  if (out2) out2[c] = (float)s2;
}

// generic column sum: out[c] = sum_r in[r, c]   (bias gradients)
template <typename TA>
__global__ __launch_bounds__(NT) void colsum_partial_kernel(const TA* __restrict__ in, int64_t M, int C, int ld,
                                                            int64_t rows_per_block, float* __restrict__ partials,
                                                            int64_t stride_in) {
  in += (size_t)blockIdx.z * stride_in;                       // batch entry (blockIdx.z): its matrix, its partial rows
  partials += (size_t)blockIdx.z * gridDim.x * C;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  for (int c = blockIdx.y * NT + threadIdx.x; c < C; c += gridDim.y * NT) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
      s0 += ldx1(in + (size_t)r * ld + c);
      s1 += ldx1(in + (size_t)(r + 1) * ld + c);
      s2 += ldx1(in + (size_t)(r + 2) * ld + c);
      s3 += ldx1(in + (size_t)(r + 3) * ld + c);
    }
    for (; r < r1; ++r) s0 += ldx1(in + (size_t)r * ld + c);
    partials[(size_t)blockIdx.x * C + c] = (s0 + s1) + (s2 + s3);
  }
}

__global__ void colsum_finalize1_kernel(const float* __restrict__ partials, int nblk, int C, float* __restrict__ out,
                                        int64_t stride_out) {
  partials += (size_t)blockIdx.y * nblk * C;                  // batch entry (blockIdx.y)
  out += (size_t)blockIdx.y * stride_out;
  __shared__ double sh[1][FIN_LANES][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[1];
  reduce_partials<1>(partials, nblk, (size_t)C, C, c, valid, sh, r);
  const double s0 = r[0];
  if (!valid || threadIdx.x >= FIN_COLS) return;
  out[c] = (float)s0;
}

}  // namespace

extern "C" int64_t mmfn_norm_workspace_bytes(int C) { return (int64_t)1024 * 2 * C * (int64_t)sizeof(double); }

namespace {
template <typename TX>
int bn_train_stats_launch(const TX* x, int64_t M, int C, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, void* workspace, void* stream) {
  if (C % 4 || C > 1024 || (NT % (C / 4)) || M <= 0 || !workspace) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int64_t rpb;
  const int nblk = bn_grid(M, C, &rpb);
  hipLaunchKernelGGL((col_partial_kernel<0, 0, TX, TX>), dim3(nblk), dim3(NT), 0, s, x, (const TX*)nullptr, (const TX*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, rpb,
                     (double*)workspace);
  MMFN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, s, (const double*)workspace, nblk, M, C,
                     eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked);
  MMFN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmfn_bn_train_stats_f32(const float* x, int64_t M, int C, float eps, float momentum, float* mean,
                                       float* rstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, void* workspace, void* stream) {
  return bn_train_stats_launch(x, M, C, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked, workspace, stream);
}
extern "C" int mmfn_bn_train_stats_bf16(const void* x, int64_t M, int C, float eps, float momentum, float* mean,
                                        float* rstd, float* running_mean, float* running_var,
                                        int64_t* num_batches_tracked, void* workspace, void* stream) {
  return bn_train_stats_launch((const bf16_t*)x, M, C, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked,
                               workspace, stream);
}

// Second half of mmfn_bn_train_stats_f32 for producers that emit the per-block (sum, sum of squares) rows themselves
// (the Winograd output transform): partials is [nblk][2][C] doubles.
extern "C" int mmfn_bn_finalize_stats_f32(const double* partials, int nblk, int64_t M, int C, float eps, float momentum,
                                          float* mean, float* rstd, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, void* stream) {
  if (!partials || nblk <= 0 || M <= 0 || C <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream, partials,
                     nblk, M, C, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_bn_eval_prepare_f32(const float* running_mean, const float* running_var, float eps, int C, float* mean,
                                        float* rstd, void* stream) {
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean,
                     running_var, eps, C, mean, rstd);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_bn_fold_f32(const float* w, int Cout, int K, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, float* w_out, float* b_out, void* stream) {
  if (!w || !w_out || !b_out || Cout <= 0 || K <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(Cout), dim3(NT), 0, (hipStream_t)stream, w, K, gamma, beta, running_mean, running_var, eps,
                     w_out, b_out);
  MMFN_LAUNCH_CHECK();
  return 0;
}

namespace {
template <typename TX, typename TA>
int bn_apply_launch(const TX* x, const TA* res, TA* y, int64_t M, int C, const float* mean, const float* rstd, const float* weight,
                    const float* bias, int relu, void* stream) {
  if (C % 4 || M <= 0) return MMFN_EINVAL;
  const int64_t total4 = M * (C / 4);
  const int blocks = (int)std::min<int64_t>(ceil_div64(total4, NT), 8192);
#define MMFN_AP(R, L) hipLaunchKernelGGL((bn_apply_kernel<R, L, TX, TA>), dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, res, y, total4, C, mean, rstd, weight, bias)
  if (res) { if (relu) MMFN_AP(true, true); else MMFN_AP(true, false); }
  else     { if (relu) MMFN_AP(false, true); else MMFN_AP(false, false); }
#undef MMFN_AP
  MMFN_LAUNCH_CHECK();
  return 0;
}
// reduce != 0: the two reductions (dweight, dbias, means);  apply != 0: the elementwise pass (dx, ge_out) from `means`
// bias != NULL with y == NULL: the ReLU mask is recomputed from x (needs weight too)
template <typename TX, typename TA>
int bn_bwd_launch(const TA* g, const TA* y, const TX* x, int64_t M, int C, const float* mean, const float* rstd, const float* weight,
                  const float* bias, TX* dx, TA* ge_out, float* dweight, float* dbias, float* means, void* workspace, int reduce,
                  int apply, void* stream) {
  if (C % 4 || C > 1024 || (NT % (C / 4)) || M <= 0 || !workspace || (bias && !weight)) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int64_t rpb;
  const int nblk = bn_grid(M, C, &rpb);
  double* partials = (double*)workspace;
  if (!means) means = (float*)(partials + (size_t)nblk * 2 * C);
  if (reduce) {
#define MMFN_CP(MK) hipLaunchKernelGGL((col_partial_kernel<1, MK, TX, TA>), dim3(nblk), dim3(NT), 0, s, x, g, y, mean, rstd, weight, bias, M, C, rpb, partials)
    if (y) MMFN_CP(1); else if (bias) MMFN_CP(2); else MMFN_CP(0);
#undef MMFN_CP
    MMFN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, s, partials, nblk, M, C, dweight, dbias,
                       means);
    MMFN_LAUNCH_CHECK();
  }
  if (apply) {
    const int64_t total4 = M * (C / 4);
    const int blocks = (int)std::min<int64_t>(ceil_div64(total4, NT), 8192);
#define MMFN_BA(MK, GE) hipLaunchKernelGGL((bn_bwd_apply_kernel<MK, GE, TX, TA>), dim3(blocks), dim3(NT), 0, s, g, y, x, dx, ge_out, total4, C, mean, rstd, weight, bias, means)
#define MMFN_BA2(MK) do { if (ge_out) MMFN_BA(MK, true); else MMFN_BA(MK, false); } while (0)
    if (y) MMFN_BA2(1); else if (bias) MMFN_BA2(2); else MMFN_BA2(0);
#undef MMFN_BA2
#undef MMFN_BA
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace

extern "C" int mmfn_bn_apply_f32(const float* x, const float* res, float* y, int64_t M, int C, const float* mean,
                                 const float* rstd, const float* weight, const float* bias, int relu, void* stream) {
  return bn_apply_launch(x, res, y, M, C, mean, rstd, weight, bias, relu, stream);
}
/* bf16 mode: y / res bf16; x (the convolution output) bf16, or fp32 when x_is_f32 (the 7x7 stems, whose convolution stays fp32) */
extern "C" int mmfn_bn_apply_bf16(const void* x, int x_is_f32, const void* res, void* y, int64_t M, int C, const float* mean,
                                  const float* rstd, const float* weight, const float* bias, int relu, void* stream) {
  if (x_is_f32) return bn_apply_launch((const float*)x, (const bf16_t*)res, (bf16_t*)y, M, C, mean, rstd, weight, bias, relu, stream);
  return bn_apply_launch((const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y, M, C, mean, rstd, weight, bias, relu, stream);
}

extern "C" int mmfn_bn_bwd_f32(const float* g, const float* y, const float* x, int64_t M, int C, const float* mean,
                               const float* rstd, const float* weight, const float* relu_bias, float* dx, float* ge_out,
                               float* dweight, float* dbias, void* workspace, void* stream) {
  return bn_bwd_launch(g, y, x, M, C, mean, rstd, weight, relu_bias, dx, ge_out, dweight, dbias, (float*)nullptr, workspace, 1, 1,
                       stream);
}
/* bf16 mode: g, y, ge_out bf16; x and dx (the convolution output and its gradient) bf16, or both fp32 when x_is_f32 (stems) */
extern "C" int mmfn_bn_bwd_bf16(const void* g, const void* y, const void* x, int x_is_f32, int64_t M, int C, const float* mean,
                                const float* rstd, const float* weight, void* dx, void* ge_out, float* dweight, float* dbias,
                                void* workspace, void* stream) {
  if (x_is_f32)
    return bn_bwd_launch((const bf16_t*)g, (const bf16_t*)y, (const float*)x, M, C, mean, rstd, weight, (const float*)nullptr, (float*)dx,
                         (bf16_t*)ge_out, dweight, dbias, (float*)nullptr, workspace, 1, 1, stream);
  return bn_bwd_launch((const bf16_t*)g, (const bf16_t*)y, (const bf16_t*)x, M, C, mean, rstd, weight, (const float*)nullptr, (bf16_t*)dx,
                       (bf16_t*)ge_out, dweight, dbias, (float*)nullptr, workspace, 1, 1, stream);
}

// First two launches of mmfn_bn_bwd_f32 only: dweight, dbias and means[2][C] = (mean(ge), mean(ge * xhat)); the caller
// applies them itself (mmfn_wino_outgrad_bn_f32 forms dx inside the Winograd output-gradient transform).
extern "C" int mmfn_bn_bwd_reduce_f32(const float* g, const float* y, const float* x, int64_t M, int C, const float* mean,
                                      const float* rstd, const float* relu_weight, const float* relu_bias, float* dweight,
                                      float* dbias, float* means, void* workspace, void* stream) {
  if (!means) return MMFN_EINVAL;
  return bn_bwd_launch(g, y, x, M, C, mean, rstd, relu_weight, relu_bias, (float*)nullptr, (float*)nullptr, dweight, dbias, means,
                       workspace, 1, 0, stream);
}

namespace {
template <typename TA, typename TO>
int layernorm_fwd_launch(const TA* x, const float* weight, const float* bias, TO* y, float* mean, float* rstd, int M, int C, float eps,
                         int act, void* stream) {
  if (C % 64 || C > 512 || M <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL((layernorm_fwd_kernel<8, TA, TO>), dim3(ceil_div(M, NT / 64)), dim3(NT), 0, (hipStream_t)stream, x, weight, bias, y,
                     mean, rstd, M, C, eps, act);
  MMFN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmfn_layernorm_fwd_f32(const float* x, const float* weight, const float* bias, float* y, float* mean,
                                      float* rstd, int M, int C, float eps, int act, void* stream) {
  return layernorm_fwd_launch(x, weight, bias, y, mean, rstd, M, C, eps, act, stream);
}
extern "C" int mmfn_layernorm_fwd_bf16(const void* x, int x_is_f32, const float* weight, const float* bias, void* y, float* mean,
                                       float* rstd, int M, int C, float eps, int act, void* stream) {
  if (x_is_f32) return layernorm_fwd_launch((const float*)x, weight, bias, (bf16_t*)y, mean, rstd, M, C, eps, act, stream);
  return layernorm_fwd_launch((const bf16_t*)x, weight, bias, (bf16_t*)y, mean, rstd, M, C, eps, act, stream);
}

extern "C" int mmfn_layernorm_bwd_f32(const float* g, const float* x, const float* weight, const float* bias,
                                      const float* mean, const float* rstd, const float* dres, float* dx, float* dweight,
                                      float* dbias, int M, int C, int act, void* workspace, void* stream) {
  return mmfn_layernorm_bwd_drop_f32(g, x, weight, bias, mean, rstd, dres, dx, dweight, dbias, M, C, act, nullptr, 0.f, nullptr, 0,
                                     nullptr, workspace, stream);
}

// 384 blocks of >= 8 rows: interleaved A/B of the training step (tools/ab_bench.sh) - 768 blocks 34.71 ms, 512 34.65, 384 34.46,
// 256 34.44, 192 34.50, 128 34.77: fewer partial rows for the finalize and more rows per block to amortise the block's LDS combine
static int ln_bwd_rows_per_block(int M) { return std::max(8, ceil_div(M, 384)); }

extern "C" int mmfn_layernorm_bwd_rows(int M) { return M > 0 ? ceil_div(M, ln_bwd_rows_per_block(M)) : 0; }

namespace {
template <typename TA, typename TX>
int layernorm_bwd_partial_launch(const TA* g, const TX* x, const float* weight, const float* bias, const float* mean, const float* rstd,
                                 const TX* dres, TX* dx, int M, int C, int act, TA* dx_dropped, float drop_p,
                                 const uint64_t* rng_state, uint32_t rng_stream, int want_colsum, float* partials, void* stream) {
  if (M <= 0 || !partials) return MMFN_EINVAL;
  if (dx_dropped && (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !rng_state))) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int rpb = ln_bwd_rows_per_block(M);
  const int nblk = ceil_div(M, rpb);
#define MMFN_LN_BWD(VW, NCH) \
  hipLaunchKernelGGL((layernorm_bwd_kernel<VW, NCH, TA, TX>), dim3(nblk), dim3(NT), 0, s, g, x, weight, bias, mean, rstd, dres, dx, \
                     partials, M, act, rpb, dx_dropped, drop_p, rng_state, rng_stream, want_colsum ? 1 : 0)
  switch (C) {
    case 64: MMFN_LN_BWD(1, 1); break;
    case 128: MMFN_LN_BWD(2, 1); break;
    case 256: MMFN_LN_BWD(4, 1); break;
    case 512: MMFN_LN_BWD(4, 2); break;
    default: return MMFN_EINVAL;  // the MMFN transformers / VectorNet use exactly these widths
  }
#undef MMFN_LN_BWD
  MMFN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmfn_layernorm_bwd_partial_f32(const float* g, const float* x, const float* weight, const float* bias,
                                              const float* mean, const float* rstd, const float* dres, float* dx, int M, int C,
                                              int act, float* dx_dropped, float drop_p, const uint64_t* rng_state,
                                              uint32_t rng_stream, int want_colsum, float* partials, void* stream) {
  return layernorm_bwd_partial_launch(g, x, weight, bias, mean, rstd, dres, dx, M, C, act, dx_dropped, drop_p, rng_state, rng_stream,
                                      want_colsum, partials, stream);
}
extern "C" int mmfn_layernorm_bwd_partial_bf16(const void* g, const void* x, int stream_is_f32, const float* weight, const float* bias,
                                               const float* mean, const float* rstd, const void* dres, void* dx, int M, int C,
                                               int act, void* dx_dropped, float drop_p, const uint64_t* rng_state,
                                               uint32_t rng_stream, int want_colsum, float* partials, void* stream) {
  if (stream_is_f32)
    return layernorm_bwd_partial_launch((const bf16_t*)g, (const float*)x, weight, bias, mean, rstd, (const float*)dres, (float*)dx, M, C,
                                        act, (bf16_t*)dx_dropped, drop_p, rng_state, rng_stream, want_colsum, partials, stream);
  return layernorm_bwd_partial_launch((const bf16_t*)g, (const bf16_t*)x, weight, bias, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, M,
                                      C, act, (bf16_t*)dx_dropped, drop_p, rng_state, rng_stream, want_colsum, partials, stream);
}

// The same reduction for n LayerNorms of one shape in ONE launch (blockIdx.y = entry): table[4 * e + {0, 1, 2, 3}] = partial rows
// ([rows][3][C] with a column-sum output, [rows][2][C] without), dweight, dbias, dx_colsum or NULL.  A fusion transformer's 17
// LayerNorm backward passes leave their row reductions to one such launch at its end instead of 17 on its side streams.
__global__ void colsum_finalize_batched_kernel(const void* const* __restrict__ table, int nblk, int C) {
  __shared__ double sh[3][FIN_LANES][FIN_COLS];
  const void* const* e = table + 4 * blockIdx.y;
  const float* partials = reinterpret_cast<const float*>(e[0]);
  float* out0 = reinterpret_cast<float*>(const_cast<void*>(e[1]));
  float* out1 = reinterpret_cast<float*>(const_cast<void*>(e[2]));
  float* out2 = reinterpret_cast<float*>(const_cast<void*>(e[3]));
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[3] = {0, 0, 0};
  if (out2) reduce_partials<3>(partials, nblk, (size_t)3 * C, C, c, valid, sh, r);
  else reduce_partials<2>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  if (!valid || threadIdx.x >= FIN_COLS) return;
  out0[c] = (float)r[0];
  out1[c] = (float)r[1];
  if (out2) out2[c] = (float)r[2];
}

extern "C" int mmfn_layernorm_bwd_finalize_batched_f32(const void* const* table, int n, int rows, int C, void* stream) {
  if (!table || n <= 0 || rows <= 0 || C <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(colsum_finalize_batched_kernel, dim3(ceil_div(C, FIN_COLS), n), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream,
                     table, rows, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_layernorm_bwd_finalize_f32(const float* partials, int rows, int C, float* dweight, float* dbias,
                                               float* dx_colsum, void* stream) {
  if (!partials || rows <= 0 || C <= 0 || !dweight || !dbias) return MMFN_EINVAL;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream, partials,
                     rows, C, dweight, dbias, dx_colsum);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_layernorm_bwd_drop_f32(const float* g, const float* x, const float* weight, const float* bias,
                                           const float* mean, const float* rstd, const float* dres, float* dx, float* dweight,
                                           float* dbias, int M, int C, int act, float* dx_dropped, float drop_p,
                                           const uint64_t* rng_state, uint32_t rng_stream, float* dx_colsum, void* workspace,
                                           void* stream) {
  if (M <= 0 || !workspace) return MMFN_EINVAL;
  const int rc = mmfn_layernorm_bwd_partial_f32(g, x, weight, bias, mean, rstd, dres, dx, M, C, act, dx_dropped, drop_p, rng_state,
                                                rng_stream, dx_colsum ? 1 : 0, (float*)workspace, stream);
  if (rc) return rc;
  return mmfn_layernorm_bwd_finalize_f32((const float*)workspace, mmfn_layernorm_bwd_rows(M), C, dweight, dbias, dx_colsum, stream);
}

static int colsum_blocks(int64_t M, int64_t* rpb) {
  *rpb = std::max<int64_t>(32, ceil_div64(M, 128));
  return (int)ceil_div64(M, *rpb);
}

extern "C" int64_t mmfn_colsum_workspace_bytes(int64_t M, int C) {
  int64_t rpb;
  return (int64_t)colsum_blocks(M, &rpb) * C * (int64_t)sizeof(float);
}

namespace {
template <typename TA>
int colsum_batched_launch(const TA* in, int batch, int64_t stride_in, int64_t M, int C, int ld, float* out, int64_t stride_out,
                          void* workspace, void* stream) {
  if (M <= 0 || C <= 0 || batch <= 0 || batch > 65535 || !workspace) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int64_t rpb;
  const int nblk = colsum_blocks(M, &rpb);
  float* partials = (float*)workspace;
  hipLaunchKernelGGL(colsum_partial_kernel<TA>, dim3(nblk, std::min(ceil_div(C, NT), 1024), batch), dim3(NT), 0, s, in, M, C, ld, rpb,
                     partials, stride_in);
  MMFN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finalize1_kernel, dim3(ceil_div(C, FIN_COLS), batch), dim3(FIN_COLS * FIN_LANES), 0, s, partials, nblk, C,
                     out, stride_out);
  MMFN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

namespace {
__global__ void colsum_partials_kernel(const double* __restrict__ partials, int nblk, int C, float* __restrict__ out) {
  __shared__ double sh[1][FIN_LANES][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
  const bool valid = c < C;
  double r[1];
  reduce_partials<1>(partials, nblk, (size_t)2 * C, C, c, valid, sh, r);
  if (!valid || threadIdx.x >= FIN_COLS) return;
  out[c] = (float)r[0];
}
}  // namespace

extern "C" int mmfn_colsum_partials_f64(const double* partials, int rows, int C, float* out, void* stream) {
  if (!partials || rows <= 0 || C <= 0 || !out) return MMFN_EINVAL;
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream, partials, rows,
                     C, out);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_bn_bwd_partials_bf16(const double* partials, int rows, const void* g, const void* y, const void* x, int64_t M, int C,
                                         const float* mean, const float* rstd, const float* weight, void* dx, void* ge_out,
                                         float* dweight, float* dbias, void* workspace, void* stream) {
  if (!partials || rows <= 0 || C % 4 || M <= 0 || !workspace) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* means = (float*)workspace;   // [2][C]
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, s, partials, rows, M, C, dweight, dbias,
                     means);
  MMFN_LAUNCH_CHECK();
  const int64_t total4 = M * (C / 4);
  const int blocks = (int)std::min<int64_t>(ceil_div64(total4, NT), 8192);
#define MMFN_BA16(MK, GE) hipLaunchKernelGGL((bn_bwd_apply_kernel<MK, GE, bf16_t, bf16_t>), dim3(blocks), dim3(NT), 0, s, (const bf16_t*)g, (const bf16_t*)y, (const bf16_t*)x, (bf16_t*)dx, (bf16_t*)ge_out, total4, C, mean, rstd, weight, (const float*)nullptr, means)
  if (y) { if (ge_out) MMFN_BA16(1, true); else MMFN_BA16(1, false); }
  else   { if (ge_out) MMFN_BA16(0, true); else MMFN_BA16(0, false); }
#undef MMFN_BA16
  MMFN_LAUNCH_CHECK();
  return 0;
}

/* the finalize half of mmfn_bn_bwd_partials_bf16 alone (the elementwise half runs in the loader of mmfn_conv3x3_halo_bf16, pro 2) */
extern "C" int mmfn_bn_bwd_finalize_f64(const double* partials, int rows, int64_t M, int C, float* dweight, float* dbias, float* means,
                                        void* stream) {
  if (!partials || rows <= 0 || C <= 0 || M <= 0 || !dweight || !dbias || !means) return MMFN_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream, partials, rows, M, C,
                     dweight, dbias, means);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_colsum_batched_f32(const float* in, int batch, int64_t stride_in, int64_t M, int C, int ld, float* out,
                                       int64_t stride_out, void* workspace, void* stream) {
  return colsum_batched_launch(in, batch, stride_in, M, C, ld, out, stride_out, workspace, stream);
}

extern "C" int mmfn_colsum_f32(const float* in, int64_t M, int C, int ld, float* out, void* workspace, void* stream) {
  return mmfn_colsum_batched_f32(in, 1, 0, M, C, ld, out, 0, workspace, stream);
}
/* column sums of a bf16 [M, C] matrix (row stride ld elements) into fp32: bias gradients in the bf16 mode */
extern "C" int mmfn_colsum_bf16(const void* in, int64_t M, int C, int ld, float* out, void* workspace, void* stream) {
  return colsum_batched_launch((const bf16_t*)in, 1, 0, M, C, ld, out, 0, workspace, stream);
}
