// Library info + small utility kernels (fill, scale-add) used by the host orchestration.
#include <algorithm>

#include "common.h"

namespace {
__global__ void noop_kernel() {}

__global__ void fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// y = a*x + b*y
__global__ void axpby_kernel(float* y, const float* x, float a, float b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0.0f ? 0.0f : b * y[i]);
}
// out = y > 0 ? g : 0
__global__ void relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = y[i] > 0.0f ? g[i] : 0.0f;
}
__global__ void rng_advance_kernel(uint64_t* state) { state[1] += 1; }
// out[i] = in[i] * keepscale(i): the backward of an epilogue dropout (same index = row*N + col)
__global__ void dropout_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float p,
                                     const uint64_t* __restrict__ state, uint32_t stream_id) {
  const uint64_t key = mmfn_rng_key(state, stream_id);
  const float inv_keep = 1.0f / (1.0f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] * mmfn_dropout_scale(key, (uint64_t)i, p, inv_keep);
}
}  // namespace

extern "C" int mmfn_abi_version(void) { return 1; }
extern "C" int mmfn_sizeof_gemm_desc(void) { return (int)sizeof(mmfn_gemm_desc); }

extern "C" int mmfn_device_selftest(void* stream) {
  hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}

// w[Co][T][Ci] -> wt[Ci][T][Co] with the taps reversed (t -> T-1-t): the filter of the transposed convolution.
// One 32x32 (co, ci) tile per block through LDS so both the read (ci contiguous) and the write (co contiguous)
// are 128-byte runs.
__global__ __launch_bounds__(256) void conv_weight_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co,
                                                               int T, int Ci) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Co && ci < Ci) ? w[((size_t)co * T + t) * Ci + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Ci && co < Co) wt[((size_t)ci * T + (T - 1 - t)) * Co + co] = tile[tx][r];
  }
}

extern "C" int mmfn_conv_weight_flip_f32(const float* w, float* wt, int Co, int T, int Ci, void* stream) {
  if (Co <= 0 || T <= 0 || Ci <= 0 || !w || !wt) return MMFN_EINVAL;
  hipLaunchKernelGGL(conv_weight_flip_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32, T), dim3(256), 0, (hipStream_t)stream, w, wt,
                     Co, T, Ci);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_fill_f32(float* p, float v, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, v, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_axpby_f32(float* y, const float* x, float a, float b, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, x, a, b, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_rng_advance(uint64_t* state, void* stream) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_dropout_apply_f32(const float* in, float* out, int64_t n, float p, const uint64_t* rng_state,
                                      uint32_t rng_stream, void* stream) {
  if (n <= 0) return 0;
  if (!rng_state || p < 0.f || p >= 1.f) return MMFN_EINVAL;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, p, rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_relu_mask_f32(const float* g, const float* y, float* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, y, out, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- bf16 weight shadows (bf16 training mode)
// The master weights stay fp32 in the flat buffer; once per step (they only change at the optimizer update) the GEMM
// operands are derived from them: (a) the whole flat buffer rounded to bf16 - same offsets, so every forward weight view
// ([Cout][KH][KW][Cin] filters, [out, in] Linear weights) is a view of the shadow - and (b) transposed copies for the data
// gradients, which read the weight with the OTHER index contiguous: w[R][T][C] -> wt[C][T][R] (Linear: T = 1; convolution:
// [Cout][taps][Cin] -> [Cin][taps][Cout], taps in place).  Replaces the per-call casts torch.autocast inserts in front of
// every aten::linear / convolution (and their backward) of the reference's training step.
namespace {
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    stx4(out + i * 4, ldx4(in + i * 4));
}

struct ShadowEntry {   // 40 bytes
  const float* src;
  bf16_t* dst;
  int32_t R, T, C;
  int32_t tiles_c;     // ceil(C / 32)
  int64_t tile0;       // first tile index of this entry (tiles = T * ceil(R/32) * ceil(C/32))
};

__global__ __launch_bounds__(256) void shadow_transpose_kernel(const ShadowEntry* __restrict__ table, int n) {
  __shared__ float tile[32][33];
  // the entry that holds this block's tile: binary search over the ascending tile0
  int lo = 0, hi = n - 1;
  const int64_t id = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile0 <= id) lo = mid; else hi = mid - 1;
  }
  const ShadowEntry e = table[lo];
  int64_t q = id - e.tile0;
  const int tiles_r = (e.R + 31) >> 5;
  const int tc = (int)(q % e.tiles_c); q /= e.tiles_c;
  const int tr = (int)(q % tiles_r);
  const int t = (int)(q / tiles_r);
  const int r0 = tr * 32, c0 = tc * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < e.R && c < e.C) ? e.src[((size_t)r * e.T + t) * e.C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < e.C && r < e.R) stx1(e.dst + ((size_t)c * e.T + t) * e.R + r, tile[tx][j]);
  }
}
}  // namespace

// ---- LayerNorm folded into the Linear that consumes it (mmfn_gemm_desc.ln_c1): per output row n of every registered Linear
//   Wf[n][k] = W[n][k] * gamma[k],   c1[n] = sum_k Wf[n][k],   c2[n] = sum_k beta[k] * W[n][k] + bias[n]
// One wave per row, all layers of the step in one launch (the weights change only at the optimizer step).
namespace {
struct LnFoldEntry {   // 72 bytes
  const float* W; const float* gamma; const float* beta; const float* bias;
  float* Wf; float* c1; float* c2;
  int32_t N, K;
  int64_t row0;        // first global row index of this entry
};

__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const LnFoldEntry* __restrict__ table, int n_entries, int64_t total_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].row0 <= row) lo = mid; else hi = mid - 1;
  }
  const LnFoldEntry e = table[lo];
  const int n = (int)(row - e.row0);
  const float* w = e.W + (size_t)n * e.K;
  float* wf = e.Wf + (size_t)n * e.K;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < e.K; k += 64) {
    const float v = w[k], f = v * e.gamma[k];
    wf[k] = f;
    s1 += (double)f;
    s2 += (double)e.beta[k] * (double)v;
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    e.c1[n] = (float)s1;
    e.c2[n] = (float)(s2 + (e.bias ? (double)e.bias[n] : 0.0));
  }
}
}  // namespace

extern "C" int mmfn_ln_fold_weights_f32(const void* table, int n_entries, int64_t total_rows, void* stream) {
  if (!table || n_entries <= 0 || total_rows <= 0) return MMFN_EINVAL;
  hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((unsigned)((total_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const LnFoldEntry*)table, n_entries, total_rows);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (n % 4 || !in || !out) return MMFN_EINVAL;
  const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, n / 4);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (n % 4 || !in || !out) return MMFN_EINVAL;
  const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, n / 4);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_shadow_transpose_bf16(const void* table, int n_entries, int64_t total_tiles, void* stream) {
  if (!table || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffff) return MMFN_EINVAL;
  hipLaunchKernelGGL(shadow_transpose_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const ShadowEntry*)table, n_entries);
  MMFN_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- the 7x7 stems as explicit im2col + plain GEMM
// conv1 of the torchvision ResNets (model_vec.py:509,515: 7x7, stride 2, 3 camera / 2 BEV channels).  With K = 49 * Cin =
// 147 / 98 the implicit-GEMM gather spends its time on per-element tap arithmetic (40 TF/s on the generic kernel); because Cin
// is tiny the im2col matrix itself is cheap - [B*OH*OW][KP] is 2.5x the convolution's OUTPUT - and turns both the forward and
// the weight gradient (which re-reads the same matrix) into plain tuned GEMMs.  KP = K rounded up to the GEMM's k-tile,
// zero-filled, as are the taps that fall into the padding.
namespace {
// One thread per (output pixel, 4 consecutive k): the tap of every k - its offset inside the image and its (kh, kw) for the padding
// test - comes from a per-block LDS table (K <= 256 entries, built once per block), the pixel from a multiply-high by the host's
// reciprocal of KP / 4 (plus a correction step) and, where OW and OH * OW are powers of two (every stem here), shifts: the first version decoded both with
// eight integer divisions per thread and ran at 1-3 TB/s of its 190-360 MB (114-175 us per stem, the longest kernel of the forward).
template <typename T>
__global__ __launch_bounds__(256) void im2col_small_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int H, int W, int Cin,
                                                           int OH, int OW, int KH, int KW, int stride, int pad, int K, int KP,
                                                           unsigned inv_kq, int log2_ow, int log2_ohw) {
  __shared__ int tap_off[256];
  __shared__ int tap_hw[256];
  for (int k = threadIdx.x; k < KP && k < 256; k += 256) {
    int off = 0, hw = -1;
    if (k < K) {
      const int tap = k / Cin, ci = k - tap * Cin;
      const int kh = tap / KW, kw = tap - kh * KW;
      off = (kh * W + kw) * Cin + ci;
      hw = (kh << 16) | kw;
    }
    tap_off[k] = off;
    tap_hw[k] = hw;
  }
  __syncthreads();
  const int kq = KP >> 2;
  const int64_t total = (int64_t)B * OH * OW * kq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned m = (unsigned)(((uint64_t)(unsigned)i * inv_kq) >> 32);   // floor(2^32 / kq) underestimates i / kq by at most 2 for i < 2^31
    unsigned rq = (unsigned)i - m * (unsigned)kq;
    while (rq >= (unsigned)kq) { ++m; rq -= (unsigned)kq; }
    const int k4 = (int)rq * 4;
    int b, oh, ow;
    if (log2_ow >= 0) {
      ow = m & (OW - 1);
      oh = (m >> log2_ow) & (OH - 1);
      b = m >> log2_ohw;
    } else {
      ow = m % OW;
      const unsigned t = m / OW;
      oh = t % OH;
      b = t / OH;
    }
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    const float* base = x + ((size_t)(b * H + ih0) * W + iw0) * Cin;   // (may point outside for padded taps: only dereferenced when inside)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int hw = tap_hw[k4 + e];
      const int ih = ih0 + (hw >> 16), iw = iw0 + (hw & 0xffff);
      if (hw >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v[e] = base[tap_off[k4 + e]];
    }
    stx4(col + i * 4, v);
  }
}

// The same matrix through LDS: a block takes SEG consecutive output pixels of one output row, copies the KH input rows they read
// (contiguous runs: coalesced loads, padding zero-filled once) into LDS and writes the SEG x KP tile with consecutive 16-byte (8-byte
// for bf16) stores - the kernel is then bound by writing the matrix (fp32 335 MB, bf16 201 MB per stem) instead of by gather loads.
constexpr int IM2COL_SEG = 64;
template <typename T>
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ x, T* __restrict__ col, int H, int W, int Cin, int OH,
                                                          int OW, int KH, int KW, int stride, int pad, int K, int KP, int row_len,
                                                          int segs_per_row) {
  extern __shared__ float patch[];          // [KH][row_len]: input columns iw0 .. iw0 + (SEG - 1) * stride + KW - 1, Cin channels each
  __shared__ int tap_off[256];              // k -> kh * row_len + kw * Cin + ci, or -1 (k >= K: zero column)
  const int seg = blockIdx.x % segs_per_row, oh = (blockIdx.x / segs_per_row) % OH, b = blockIdx.x / (segs_per_row * OH);
  const int ow0 = seg * IM2COL_SEG, npix = min(IM2COL_SEG, OW - ow0);
  const int ih0 = oh * stride - pad, iw0 = ow0 * stride - pad;
  for (int k = threadIdx.x; k < KP; k += 256) {
    int off = -1;
    if (k < K) {
      const int tap = k / Cin, ci = k - tap * Cin;
      const int kh = tap / KW, kw = tap - kh * KW;
      off = kh * row_len + kw * Cin + ci;
    }
    tap_off[k] = off;
  }
  // rows: one contiguous run of row_len floats each, valid where the input column iw0 + r / Cin lies inside the image
  const int r_lo = max(0, -iw0) * Cin, r_hi = min(row_len, (W - iw0) * Cin);
  for (int kh = 0; kh < KH; ++kh) {
    const int ih = ih0 + kh;
    const bool row_ok = (unsigned)ih < (unsigned)H;
    const float* src = x + ((size_t)(b * H + (row_ok ? ih : 0)) * W) * Cin + (ptrdiff_t)iw0 * Cin;
    for (int r = threadIdx.x; r < row_len; r += 256) patch[kh * row_len + r] = (row_ok && r >= r_lo && r < r_hi) ? src[r] : 0.f;
  }
  __syncthreads();
  // tile: 64 lanes per output pixel (lane q < KP / 4 writes that pixel's k = 4q .. 4q+3), four pixels per pass
  const int q = threadIdx.x & 63, kq = KP >> 2;
  T* out = col + ((size_t)(b * OH + oh) * OW + ow0) * KP;
  if (q < kq) {
    int off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) off[e] = tap_off[q * 4 + e];
    for (int p = threadIdx.x >> 6; p < npix; p += 4) {
      const float* base = patch + p * stride * Cin;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = off[e] >= 0 ? base[off[e]] : 0.f;
      stx4(out + (size_t)p * KP + q * 4, v);
    }
  }
}

// dst[r][0..KP) = src[r][0..K) zero-padded (T out), or - unpad - dst[r][0..K) = src[r][0..K) out of rows of KP
template <typename TI, typename TO>
__global__ void repitch_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int R, int K, int ps, int pd) {
  const int64_t total = (int64_t)R * pd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / pd), k = (int)(i - (int64_t)r * pd);
    if (k < K) stx1(dst + (size_t)r * pd + k, ldx1(src + (size_t)r * ps + k));
    else if (k < pd) stx1(dst + (size_t)r * pd + k, 0.f);
  }
}
}  // namespace

extern "C" int mmfn_im2col_small(const float* x, void* col, int out_bf16, int B, int H, int W, int Cin, int KH, int KW, int stride,
                                 int pad, int KP, void* stream) {
  const int K = KH * KW * Cin;
  if (!x || !col || KP < K || KP % 4 || Cin <= 0 || Cin > 4) return MMFN_EINVAL;
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  {
    // LDS path: KH input rows of the segment's span fit the patch buffer (every stem does: 7 x 399 floats)
    const int row_len = ((IM2COL_SEG - 1) * stride + KW) * Cin;
    const size_t lds = (size_t)KH * row_len * sizeof(float);
    if (KP <= 256 && lds <= 48 * 1024 && (int64_t)B * OH * ceil_div(OW, IM2COL_SEG) < ((int64_t)1 << 30)) {
      const int segs = ceil_div(OW, IM2COL_SEG);
      const dim3 grid((unsigned)(B * OH * segs));
      if (out_bf16)
        hipLaunchKernelGGL(im2col_rows_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, x, (bf16_t*)col, H, W, Cin, OH, OW, KH, KW,
                           stride, pad, K, KP, row_len, segs);
      else
        hipLaunchKernelGGL(im2col_rows_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, x, (float*)col, H, W, Cin, OH, OW, KH, KW,
                           stride, pad, K, KP, row_len, segs);
      MMFN_LAUNCH_CHECK();
      return 0;
    }
  }
  const int64_t total = (int64_t)B * OH * OW * (KP / 4);
  if (KP > 256 || total >= ((int64_t)1 << 31) || KH >= 32768 || KW >= 32768) return MMFN_EINVAL;   // tap table / 32-bit element index
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  const unsigned kq = (unsigned)(KP / 4);
  const unsigned inv_kq = kq == 1 ? 0xFFFFFFFFu : (unsigned)(((uint64_t)1 << 32) / kq);   // (i * inv) >> 32 <= i / kq, corrected in the kernel
  auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  const int l_ow = lg(OW), l_oh = lg(OH);
  const int log2_ow = (l_ow >= 0 && l_oh >= 0) ? l_ow : -1, log2_ohw = log2_ow >= 0 ? l_ow + l_oh : -1;
  if (out_bf16)
    hipLaunchKernelGGL(im2col_small_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)col, B, H, W, Cin, OH, OW,
                       KH, KW, stride, pad, K, KP, inv_kq, log2_ow, log2_ohw);
  else
    hipLaunchKernelGGL(im2col_small_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (float*)col, B, H, W, Cin, OH, OW, KH,
                       KW, stride, pad, K, KP, inv_kq, log2_ow, log2_ohw);
  MMFN_LAUNCH_CHECK();
  return 0;
}

/* rows of K fp32 values: pitch ps -> pitch pd (pd >= K; columns K..pd zero-filled); dst fp32 or bf16 */
extern "C" int mmfn_repitch_rows(const float* src, void* dst, int dst_bf16, int R, int K, int ps, int pd, void* stream) {
  if (!src || !dst || R <= 0 || K <= 0 || ps < K || pd < K) return MMFN_EINVAL;
  const int blocks = (int)std::min<int64_t>(((int64_t)R * pd + 255) / 256, 4096);
  if (dst_bf16) hipLaunchKernelGGL((repitch_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, K, ps, pd);
  else hipLaunchKernelGGL((repitch_kernel<float, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, K, ps, pd);
  MMFN_LAUNCH_CHECK();
  return 0;
}
