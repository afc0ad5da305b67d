// Spatial glue of the MMFN encoder on NHWC feature maps (all HBM-bound, 16-byte accesses):
//   maxpool 3x3/s2/p1 fwd+bwd                         (torchvision stem, model_vec.py:512,518)
//   adaptive-avgpool(8x8) + token assembly fwd+bwd    (model_vec.py:527-529 + GPT.forward :223-235)
//   bilinear(align_corners) upsample + residual add, and its adjoint     (model_vec.py:531-536)
//   global avgpool + branch sum fwd+bwd                (model_vec.py:585-596)
//   NCHW <-> NHWC transposes at the module boundary
// Because feature maps are channels-last, a GPT token row [C] IS a pooled pixel: the reference's
// cat/permute/contiguous copies (model_vec.py:228,240-244) disappear.
#include <algorithm>

#include "common.h"

namespace {
constexpr int NT = 256;

// ---------------------------------------------------------------- maxpool 3x3 s2 p1
template <typename T>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         uint8_t* __restrict__ idx, int B, int H, int W, int C, int OH,
                                                         int OW) {
  const int cq = C >> 2;
  const int64_t total = (int64_t)B * OH * OW * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    int64_t p = i / cq;
    const int ow = (int)(p % OW); p /= OW;
    const int oh = (int)(p % OH);
    const int b = (int)(p / OH);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = ldx4(x + ((size_t)(b * H + ih) * W + iw) * C + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
        first = false;
      }
    }
    stx4(y + (size_t)(i / cq) * C + c4, best);
    uchar4 o;
    o.x = (uint8_t)bi[0]; o.y = (uint8_t)bi[1]; o.z = (uint8_t)bi[2]; o.w = (uint8_t)bi[3];
    *reinterpret_cast<uchar4*>(idx + (size_t)(i / cq) * C + c4) = o;
  }
}

// gather form: input pixel collects from the <=4 windows that contain it and whose argmax is it
template <typename T>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const T* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                         T* __restrict__ gx, int B, int H, int W, int C, int OH, int OW) {
  const int cq = C >> 2;
  const int64_t total = (int64_t)B * H * W * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    int64_t p = i / cq;
    const int iw = (int)(p % W); p /= W;
    const int ih = (int)(p % H);
    const int b = (int)(p / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oh_lo = max(0, ih >> 1), oh_hi = min(OH - 1, (ih + 1) >> 1);
    const int ow_lo = max(0, iw >> 1), ow_hi = min(OW - 1, (iw + 1) >> 1);
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int kh = ih - (oh * 2 - 1), kw = iw - (ow * 2 - 1);
        if ((unsigned)kh > 2u || (unsigned)kw > 2u) continue;
        const int me = kh * 3 + kw;
        const size_t off = ((size_t)(b * OH + oh) * OW + ow) * C + c4;
        const uchar4 a = *reinterpret_cast<const uchar4*>(idx + off);
        const f32x4 g = ldx4(gy + off);
        if (a.x == me) acc[0] += g[0];
        if (a.y == me) acc[1] += g[1];
        if (a.z == me) acc[2] += g[2];
        if (a.w == me) acc[3] += g[3];
      }
    stx4(gx + (size_t)(i / cq) * C + c4, acc);
  }
}

// ---------------------------------------------------------------- token assembly
// Token groups (64 tokens = one 8x8 pooled frame each), in the order GPT.forward concatenates them (model_vec.py:226-231):
// the frames of modality 0 (n_views * seq_len camera frames per sample), then modality 1 (seq_len LiDAR frames), ...
// cnt[m] = frames per sample of modality m, base[m] = its first group.  seq_len = n_views = 1: cnt = 1, base[m] = m.
struct GroupMap { int cnt[4]; int base[4]; };
static GroupMap make_groups(int n_modal, const int32_t* frames) {
  GroupMap g;
  int b = 0;
  for (int m = 0; m < 4; ++m) {
    g.cnt[m] = m < n_modal ? (frames ? frames[m] : 1) : 0;
    g.base[m] = b;
    b += g.cnt[m];
  }
  return g;
}
template <typename T> struct FeatPtrsT { const T* p[4]; };
template <typename T> struct GradPtrsT { T* p[4]; };
typedef FeatPtrsT<float> FeatPtrs;
typedef GradPtrsT<float> GradPtrs;

// tok[b, m*64 + ay*8+ax, c] = drop( pos[t,c] + mean_{kxk}(F_m[b, ay*k.., ax*k.., c]) + vel_w[c]*v[b] + vel_b[c] )
// TO: element type of the token matrix (the transformer's residual stream: fp32 also in the bf16 mode, TT = bf16 features)
template <typename TT, typename TO = TT>
__global__ __launch_bounds__(NT) void tokens_fwd_kernel(FeatPtrsT<TT> feats, GroupMap gm, int n_modal, int B, int S, int C,
                                                        const float* __restrict__ pos, const float* __restrict__ vel_w,
                                                        const float* __restrict__ vel_b, const float* __restrict__ velocity,
                                                        TO* __restrict__ tok, float drop_p,
                                                        const uint64_t* __restrict__ rng_state, uint32_t rng_stream) {
  const int cq = C >> 2;
  const int k = S >> 3;
  const float inv = 1.0f / (float)(k * k);
  const int T = (gm.base[n_modal - 1] + gm.cnt[n_modal - 1]) * 64;
  const int64_t total = (int64_t)B * T * cq;
  uint64_t key = 0;
  if (drop_p > 0.f) key = mmfn_rng_key(rng_state, rng_stream);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    const int64_t row = i / cq;  // b*T + t
    const int t = (int)(row % T), b = (int)(row / T);
    const int grp = t >> 6, a = t & 63, ay = a >> 3, ax = a & 7;
    int m = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < n_modal && grp >= gm.base[i]) m = i;
    const int img = b * gm.cnt[m] + (grp - gm.base[m]);     // frame index inside modality m's (enlarged) batch
    const TT* f = feats.p[m] + ((size_t)(img * S + ay * k) * S + ax * k) * C + c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const f32x4 v = ldx4(f + ((size_t)dy * S + dx) * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
      }
    const float vb = velocity[b];
    const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (size_t)t * C + c4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pooled = (k == 1) ? s[e] : s[e] * inv;
      o[e] = (pe[e] + pooled) + (vel_w[c4 + e] * vb + vel_b[c4 + e]);
      if (drop_p > 0.f) o[e] *= mmfn_dropout_scale(key, (uint64_t)row * C + c4 + e, drop_p, 1.0f / (1.0f - drop_p));
    }
    stx4(tok + (size_t)row * C + c4, o);
  }
}

// gm = g * dropmask (in place into gtok), plus per-(b) partial sums for pos/vel gradients:
//   dpos[t,c]  = sum_b gm[b,t,c];  dvel_w[c] = sum_{b,t} gm*v[b];  dvel_b[c] = sum_{b,t} gm
// one block per token index t (all b): deterministic, no atomics; vel partials [T][2][C].
template <typename TT>
__global__ __launch_bounds__(NT) void tokens_bwd_kernel(TT* __restrict__ gtok, int B, int T, int C,
                                                        const float* __restrict__ velocity, float* __restrict__ dpos,
                                                        float* __restrict__ vel_partials, float drop_p,
                                                        const uint64_t* __restrict__ rng_state, uint32_t rng_stream) {
  const int t = blockIdx.x;
  uint64_t key = 0;
  if (drop_p > 0.f) key = mmfn_rng_key(rng_state, rng_stream);
  for (int c = threadIdx.x; c < C; c += NT) {
    float sp = 0.f, sw = 0.f;
    for (int b = 0; b < B; ++b) {
      const size_t off = ((size_t)b * T + t) * C + c;
      float g = ldx1(gtok + off);
      if (drop_p > 0.f) {
        g *= mmfn_dropout_scale(key, (uint64_t)off, drop_p, 1.0f / (1.0f - drop_p));
        stx1(gtok + off, g);
      }
      sp += g;
      sw += g * velocity[b];
    }
    dpos[(size_t)t * C + c] = sp;
    vel_partials[((size_t)t * 2 + 0) * C + c] = sw;
    vel_partials[((size_t)t * 2 + 1) * C + c] = sp;
  }
}

// 8 columns x 32 row-lanes per block (the one-thread-per-column loop over T = 192 partial rows this replaces took 46 us
// on 1-4 blocks - on the transformers' dependent chain); fixed summation order, fp64 accumulation.
constexpr int TF_COLS = 8, TF_LANES = 32;
__global__ __launch_bounds__(TF_COLS * TF_LANES) void tokens_bwd_finalize_kernel(const float* __restrict__ vel_partials, int T, int C,
                                                                                float* __restrict__ dvel_w, float* __restrict__ dvel_b) {
  __shared__ double sh[2][TF_LANES][TF_COLS];
  const int cl = threadIdx.x % TF_COLS, rl = threadIdx.x / TF_COLS;
  const int c = blockIdx.x * TF_COLS + cl;
  double sw = 0, sb = 0;
  if (c < C)
    for (int t = rl; t < T; t += TF_LANES) {
      sw += (double)vel_partials[((size_t)t * 2) * C + c];
      sb += (double)vel_partials[((size_t)t * 2 + 1) * C + c];
    }
  sh[0][rl][cl] = sw;
  sh[1][rl][cl] = sb;
  __syncthreads();
  if (rl == 0 && c < C) {
    double tw = 0, tb = 0;
    for (int k = 0; k < TF_LANES; ++k) { tw += sh[0][k][cl]; tb += sh[1][k][cl]; }
    dvel_w[c] = (float)tw;
    dvel_b[c] = (float)tb;
  }
}

// ---------------------------------------------------------------- bilinear upsample (align_corners) + add
// out[b,y,x,c] = F[b,y,x,c] + bilinear(tok[b, m*64 + 8x8 grid, c])   (tok row stride = C)
template <typename TT>
__global__ __launch_bounds__(NT) void upsample_add_fwd_kernel(const TT* __restrict__ feat, const TT* __restrict__ tok,
                                                              TT* __restrict__ out, int B, int S, int C, int T, int m, int frames) {
  // B: frames of this modality over the whole batch (samples * frames); token group of frame i: m + i % frames of sample i / frames
  const int cq = C >> 2;
  const int64_t total = (int64_t)B * S * S * cq;
  const float r = (S > 1) ? (float)(8 - 1) / (float)(S - 1) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    int64_t p = i / cq;
    const int x = (int)(p % S); p /= S;
    const int y = (int)(p % S);
    const int b = (int)(p / S);
    const f32x4 f = ldx4(feat + (size_t)(i / cq) * C + c4);
    f32x4 o;
    if (S == 8) {
      const f32x4 t = ldx4(tok + ((size_t)(b / frames) * T + (m + b % frames) * 64 + y * 8 + x) * C + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f[e] + t[e];
    } else {
      const float h1r = r * (float)y, w1r = r * (float)x;
      const int h1 = (int)h1r, w1 = (int)w1r;
      const int h1p = (h1 < 7) ? 1 : 0, w1p = (w1 < 7) ? 1 : 0;
      const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
      const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
      const TT* base = tok + ((size_t)(b / frames) * T + (m + b % frames) * 64 + h1 * 8 + w1) * C + c4;
      const f32x4 v00 = ldx4(base);
      const f32x4 v01 = ldx4(base + (size_t)w1p * C);
      const f32x4 v10 = ldx4(base + (size_t)h1p * 8 * C);
      const f32x4 v11 = ldx4(base + (size_t)(h1p * 8 + w1p) * C);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = f[e] + (h0l * (w0l * v00[e] + w1l * v01[e]) + h1l * (w0l * v10[e] + w1l * v11[e]));
    }
    stx4(out + (size_t)(i / cq) * C + c4, o);
  }
}

// adjoint: gtok[b, m*64 + a, c] = sum_{pixels} weight(a, pixel) * G[b, pixel, c]
// YS threads per (b, anchor, c4) share the rows of the anchor's bilinear footprint (consecutive lanes, combined with
// xor shuffles in a fixed order); with one thread per output the 64x64 maps ran 128 blocks of ~300 serial loads each.
template <int YS, typename TT>
__global__ __launch_bounds__(NT) void upsample_adj_kernel(const TT* __restrict__ G, TT* __restrict__ gtok, int B, int S,
                                                          int C, int T, int m, int frames) {
  const int cq = C >> 2;
  const int64_t total = (int64_t)B * 64 * cq * YS;
  const float r = (S > 1) ? (float)(8 - 1) / (float)(S - 1) : 0.f;
  // grid covers `total` rounded up to the block size, so every lane of a shuffle group is alive
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < total;
  const int64_t o = live ? i / YS : 0;
  const int ys = (int)(i % YS);
  const int c4 = (int)(o % cq) * 4;
  const int a = (int)((o / cq) % 64), b = (int)(o / cq / 64);
  const int ay = a >> 3, ax = a & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    if (S == 8) {
      if (ys == 0) acc = ldx4(G + ((size_t)(b * 8 + ay) * 8 + ax) * C + c4);
    } else {
      // pixels y with floor(r*y) in {ay-1, ay}
      const int step = (S - 1) / 7 + 2;
      const int ylo = max(0, (int)((float)(ay - 1) / r) - 1), yhi = min(S - 1, ylo + 2 * step + 2);
      const int xlo = max(0, (int)((float)(ax - 1) / r) - 1), xhi = min(S - 1, xlo + 2 * step + 2);
      for (int y = ylo + ys; y <= yhi; y += YS) {
        const float h1r = r * (float)y;
        const int h1 = (int)h1r;
        const int h1p = (h1 < 7) ? 1 : 0;
        const float h1l = h1r - (float)h1;
        float wy = 0.f;
        if (h1 == ay) wy += 1.f - h1l;
        if (h1 + h1p == ay) wy += h1l;  // (a clamped neighbour folds onto the same anchor)
        if (wy == 0.f) continue;
        for (int x = xlo; x <= xhi; ++x) {
          const float w1r = r * (float)x;
          const int w1 = (int)w1r;
          const int w1p = (w1 < 7) ? 1 : 0;
          const float w1l = w1r - (float)w1;
          float wx = 0.f;
          if (w1 == ax) wx += 1.f - w1l;
          if (w1 + w1p == ax) wx += w1l;
          if (wx == 0.f) continue;
          const f32x4 g = ldx4(G + ((size_t)(b * S + y) * S + x) * C + c4);
          const float wgt = wy * wx;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += wgt * g[e];
        }
      }
    }
  }
#pragma unroll
  for (int d = 1; d < YS; d <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], d, 64);
  if (live && ys == 0) stx4(gtok + ((size_t)(b / frames) * T + (m + b % frames) * 64 + a) * C + c4, acc);
}

// dF[b,y,x,c] = G[b,y,x,c] + gtok[b, m*64 + (y/k)*8 + x/k, c] / k^2      (avgpool adjoint + identity)
// TG: element type of the token gradient (fp32 residual stream of the bf16 mode: TT = bf16, TG = float)
template <typename TT, typename TG = TT>
__global__ __launch_bounds__(NT) void pool_bcast_add_kernel(const TT* __restrict__ G, const TG* __restrict__ gtok,
                                                            TT* __restrict__ dF, int B, int S, int C, int T, int m, int frames) {
  const int cq = C >> 2;
  const int k = S >> 3;
  const float inv = 1.0f / (float)(k * k);
  const int64_t total = (int64_t)B * S * S * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % cq) * 4;
    int64_t p = i / cq;
    const int x = (int)(p % S); p /= S;
    const int y = (int)(p % S);
    const int b = (int)(p / S);
    const f32x4 g = ldx4(G + (size_t)(i / cq) * C + c4);
    const f32x4 t = ldx4(gtok + ((size_t)(b / frames) * T + (m + b % frames) * 64 + (y / k) * 8 + x / k) * C + c4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] + t[e] * inv;
    stx4(dF + (size_t)(i / cq) * C + c4, o);
  }
}

// ---------------------------------------------------------------- global avgpool + branch sum
template <typename T>
__global__ __launch_bounds__(NT) void gap_sum_fwd_kernel(FeatPtrsT<T> feats, GroupMap gm, int n, int B, int P, int C, float* __restrict__ out) {
  const int64_t total = (int64_t)B * C;
  const float inv = 1.0f / (float)P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), b = (int)(i / C);
    float tot = 0.f;
    for (int m = 0; m < n; ++m)
      for (int j = 0; j < gm.cnt[m]; ++j) {   // every frame of the modality (model_vec.py:585-596: view(bz, frames, -1), sum)
        float s = 0.f;
        const T* f = feats.p[m] + (size_t)(b * gm.cnt[m] + j) * P * C + c;
        for (int p = 0; p < P; ++p) s += ldx1(f + (size_t)p * C);
        tot += s * inv;
      }
    out[i] = tot;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void gap_sum_bwd_kernel(const float* __restrict__ g, GradPtrsT<T> outs, GroupMap gm, int n, int B, int P, int C) {
  const float inv = 1.0f / (float)P;
  for (int m = 0; m < n; ++m) {
    const int64_t total = (int64_t)B * gm.cnt[m] * P * C;   // every frame of a sample receives the sample's gradient
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % C), img = (int)(i / ((int64_t)P * C));
      stx1(outs.p[m] + i, g[(size_t)(img / gm.cnt[m]) * C + c] * inv);
    }
  }
}

// ---------------------------------------------------------------- layout transposes (LDS tiled)
// in [B, R, Cc] -> out [B, Cc, R]   (NCHW->NHWC with R = C, Cc = H*W; NHWC->NCHW with R = H*W, Cc = C)
template <typename TI, typename TO>
__global__ void transpose_kernel(const TI* __restrict__ in, TO* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const TI* src = in + (size_t)b * R * Cc;
  TO* dst = out + (size_t)b * R * Cc;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < R && c < Cc) tile[j][threadIdx.x] = ldx1(src + (size_t)r * Cc + c);
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < R && c < Cc) stx1(dst + (size_t)c * R + r, tile[threadIdx.x][j]);
  }
}

int grid_for(int64_t total) { return (int)std::min<int64_t>(ceil_div64(total, NT), 16384); }
}  // namespace

namespace {
template <typename T>
int maxpool_fwd_launch(const T* x, T* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  if (C % 4) return MMFN_EINVAL;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for((int64_t)B * OH * OW * (C / 4))), dim3(NT), 0, (hipStream_t)stream, x, y,
                     idx, B, H, W, C, OH, OW);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int maxpool_bwd_launch(const T* gy, const uint8_t* idx, T* gx, int B, int H, int W, int C, void* stream) {
  if (C % 4) return MMFN_EINVAL;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(NT), 0, (hipStream_t)stream, gy, idx,
                     gx, B, H, W, C, OH, OW);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T, typename TO>
int tokens_fwd_launch(const T* const* feats, int n_modal, const int32_t* frames, int B, int S, int C, const float* pos, const float* vel_w,
                      const float* vel_b, const float* velocity, TO* tok, float drop_p, const uint64_t* rng_state,
                      uint32_t rng_stream, void* stream) {
  if (C % 4 || S % 8 || n_modal < 1 || n_modal > 4) return MMFN_EINVAL;
  FeatPtrsT<T> fp;
  for (int i = 0; i < 4; ++i) fp.p[i] = i < n_modal ? feats[i] : nullptr;
  const GroupMap gm = make_groups(n_modal, frames);
  const int groups = gm.base[n_modal - 1] + gm.cnt[n_modal - 1];
  hipLaunchKernelGGL((tokens_fwd_kernel<T, TO>), dim3(grid_for((int64_t)B * groups * 64 * (C / 4))), dim3(NT), 0, (hipStream_t)stream,
                     fp, gm, n_modal, B, S, C, pos, vel_w, vel_b, velocity, tok, drop_p, rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int tokens_bwd_launch(T* gtok, int B, int Tn, int C, const float* velocity, float* dpos, float* dvel_w, float* dvel_b, float drop_p,
                      const uint64_t* rng_state, uint32_t rng_stream, void* workspace, void* stream) {
  if (!workspace) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tokens_bwd_kernel<T>, dim3(Tn), dim3(NT), 0, s, gtok, B, Tn, C, velocity, dpos, (float*)workspace, drop_p,
                     rng_state, rng_stream);
  MMFN_LAUNCH_CHECK();
  hipLaunchKernelGGL(tokens_bwd_finalize_kernel, dim3(ceil_div(C, TF_COLS)), dim3(TF_COLS * TF_LANES), 0, s, (const float*)workspace, Tn, C, dvel_w,
                     dvel_b);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int upsample_add_fwd_launch(const T* feat, const T* tok, T* out, int B, int S, int C, int Tn, int m, int frames, void* stream) {
  if (C % 4 || frames < 1 || B % frames) return MMFN_EINVAL;
  hipLaunchKernelGGL(upsample_add_fwd_kernel<T>, dim3(grid_for((int64_t)B * S * S * (C / 4))), dim3(NT), 0, (hipStream_t)stream,
                     feat, tok, out, B, S, C, Tn, m, frames);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int upsample_adj_launch(const T* G, T* gtok, int B, int S, int C, int Tn, int m, int frames, void* stream) {
  if (C % 4 || frames < 1 || B % frames) return MMFN_EINVAL;
  const int64_t outs = (int64_t)B * 64 * (C / 4);
  if (S >= 32)
    hipLaunchKernelGGL((upsample_adj_kernel<8, T>), dim3((unsigned)((outs * 8 + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, G, gtok,
                       B, S, C, Tn, m, frames);
  else
    hipLaunchKernelGGL((upsample_adj_kernel<1, T>), dim3((unsigned)((outs + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, G, gtok, B,
                       S, C, Tn, m, frames);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T, typename TG>
int pool_bcast_add_launch(const T* G, const TG* gtok, T* dF, int B, int S, int C, int Tn, int m, int frames, void* stream) {
  if (C % 4 || S % 8 || frames < 1 || B % frames) return MMFN_EINVAL;
  hipLaunchKernelGGL((pool_bcast_add_kernel<T, TG>), dim3(grid_for((int64_t)B * S * S * (C / 4))), dim3(NT), 0, (hipStream_t)stream, G,
                     gtok, dF, B, S, C, Tn, m, frames);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int gap_sum_fwd_launch(const T* const* feats, int n, const int32_t* frames, int B, int P, int C, float* out, void* stream) {
  if (n < 1 || n > 4) return MMFN_EINVAL;
  FeatPtrsT<T> fp;
  for (int i = 0; i < 4; ++i) fp.p[i] = i < n ? feats[i] : nullptr;
  hipLaunchKernelGGL(gap_sum_fwd_kernel<T>, dim3(grid_for((int64_t)B * C)), dim3(NT), 0, (hipStream_t)stream, fp, make_groups(n, frames), n, B, P, C,
                     out);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename T>
int gap_sum_bwd_launch(const float* g, T* const* outs, int n, const int32_t* frames, int B, int P, int C, void* stream) {
  if (n < 1 || n > 4) return MMFN_EINVAL;
  GradPtrsT<T> gp;
  for (int i = 0; i < 4; ++i) gp.p[i] = i < n ? outs[i] : nullptr;
  const GroupMap gm = make_groups(n, frames);
  int most = 1;
  for (int m = 0; m < n; ++m) most = std::max(most, gm.cnt[m]);
  hipLaunchKernelGGL(gap_sum_bwd_kernel<T>, dim3(grid_for((int64_t)B * most * P * C)), dim3(NT), 0, (hipStream_t)stream, g, gp, gm, n, B, P, C);
  MMFN_LAUNCH_CHECK();
  return 0;
}
template <typename TI, typename TO>
int transpose_launch(const TI* in, TO* out, int B, int R, int Cc, void* stream) {
  dim3 grid(ceil_div(Cc, 32), ceil_div(R, 32), B);
  hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(32, 8), 0, (hipStream_t)stream, in, out, R, Cc);
  MMFN_LAUNCH_CHECK();
  return 0;
}
typedef const bf16_t* cbf;
typedef bf16_t* mbf;
}  // namespace

extern "C" int mmfn_maxpool3x3s2_fwd_f32(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  return maxpool_fwd_launch(x, y, idx, B, H, W, C, stream);
}
extern "C" int mmfn_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  return maxpool_fwd_launch((cbf)x, (mbf)y, idx, B, H, W, C, stream);
}
extern "C" int mmfn_maxpool3x3s2_bwd_f32(const float* gy, const uint8_t* idx, float* gx, int B, int H, int W, int C,
                                         void* stream) {
  return maxpool_bwd_launch(gy, idx, gx, B, H, W, C, stream);
}
extern "C" int mmfn_maxpool3x3s2_bwd_bf16(const void* gy, const uint8_t* idx, void* gx, int B, int H, int W, int C, void* stream) {
  return maxpool_bwd_launch((cbf)gy, idx, (mbf)gx, B, H, W, C, stream);
}

extern "C" int mmfn_tokens_fwd_f32(const float* const* feats, int n_modal, const int32_t* frames, int B, int S, int C, const float* pos,
                                   const float* vel_w, const float* vel_b, const float* velocity, float* tok, float drop_p,
                                   const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  return tokens_fwd_launch(feats, n_modal, frames, B, S, C, pos, vel_w, vel_b, velocity, tok, drop_p, rng_state, rng_stream, stream);
}
extern "C" int mmfn_tokens_fwd_bf16(const void* const* feats, int n_modal, const int32_t* frames, int B, int S, int C, const float* pos,
                                    const float* vel_w, const float* vel_b, const float* velocity, void* tok, int tok_is_f32,
                                    float drop_p, const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  if (tok_is_f32)
    return tokens_fwd_launch((const bf16_t* const*)feats, n_modal, frames, B, S, C, pos, vel_w, vel_b, velocity, (float*)tok, drop_p,
                             rng_state, rng_stream, stream);
  return tokens_fwd_launch((const bf16_t* const*)feats, n_modal, frames, B, S, C, pos, vel_w, vel_b, velocity, (mbf)tok, drop_p, rng_state,
                           rng_stream, stream);
}

extern "C" int64_t mmfn_tokens_bwd_workspace_bytes(int T, int C) { return (int64_t)T * 2 * C * (int64_t)sizeof(float); }

extern "C" int mmfn_tokens_bwd_f32(float* gtok, int B, int T, int C, const float* velocity, float* dpos, float* dvel_w,
                                   float* dvel_b, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                                   void* workspace, void* stream) {
  return tokens_bwd_launch(gtok, B, T, C, velocity, dpos, dvel_w, dvel_b, drop_p, rng_state, rng_stream, workspace, stream);
}
extern "C" int mmfn_tokens_bwd_bf16(void* gtok, int B, int T, int C, const float* velocity, float* dpos, float* dvel_w,
                                    float* dvel_b, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                                    void* workspace, void* stream) {
  return tokens_bwd_launch((mbf)gtok, B, T, C, velocity, dpos, dvel_w, dvel_b, drop_p, rng_state, rng_stream, workspace, stream);
}

extern "C" int mmfn_upsample_add_fwd_f32(const float* feat, const float* tok, float* out, int B, int S, int C, int T, int m, int frames,
                                       void* stream) {
  return upsample_add_fwd_launch(feat, tok, out, B, S, C, T, m, frames, stream);
}
extern "C" int mmfn_upsample_add_fwd_bf16(const void* feat, const void* tok, void* out, int B, int S, int C, int T, int m, int frames,
                                        void* stream) {
  return upsample_add_fwd_launch((cbf)feat, (cbf)tok, (mbf)out, B, S, C, T, m, frames, stream);
}

extern "C" int mmfn_upsample_adj_f32(const float* G, float* gtok, int B, int S, int C, int T, int m, int frames, void* stream) {
  return upsample_adj_launch(G, gtok, B, S, C, T, m, frames, stream);
}
extern "C" int mmfn_upsample_adj_bf16(const void* G, void* gtok, int B, int S, int C, int T, int m, int frames, void* stream) {
  return upsample_adj_launch((cbf)G, (mbf)gtok, B, S, C, T, m, frames, stream);
}

extern "C" int mmfn_pool_bcast_add_f32(const float* G, const float* gtok, float* dF, int B, int S, int C, int T, int m, int frames,
                                       void* stream) {
  return pool_bcast_add_launch(G, gtok, dF, B, S, C, T, m, frames, stream);
}
extern "C" int mmfn_pool_bcast_add_bf16(const void* G, const void* gtok, int gtok_is_f32, void* dF, int B, int S, int C, int T, int m,
                                        int frames, void* stream) {
  if (gtok_is_f32) return pool_bcast_add_launch((cbf)G, (const float*)gtok, (mbf)dF, B, S, C, T, m, frames, stream);
  return pool_bcast_add_launch((cbf)G, (cbf)gtok, (mbf)dF, B, S, C, T, m, frames, stream);
}

extern "C" int mmfn_gap_sum_fwd_f32(const float* const* feats, int n, const int32_t* frames, int B, int P, int C, float* out, void* stream) {
  return gap_sum_fwd_launch(feats, n, frames, B, P, C, out, stream);
}
extern "C" int mmfn_gap_sum_fwd_bf16(const void* const* feats, int n, const int32_t* frames, int B, int P, int C, float* out, void* stream) {
  return gap_sum_fwd_launch((const bf16_t* const*)feats, n, frames, B, P, C, out, stream);
}

extern "C" int mmfn_gap_sum_bwd_f32(const float* g, float* const* outs, int n, const int32_t* frames, int B, int P, int C, void* stream) {
  return gap_sum_bwd_launch(g, outs, n, frames, B, P, C, stream);
}
extern "C" int mmfn_gap_sum_bwd_bf16(const float* g, void* const* outs, int n, const int32_t* frames, int B, int P, int C, void* stream) {
  return gap_sum_bwd_launch(g, (bf16_t* const*)outs, n, frames, B, P, C, stream);
}

extern "C" int mmfn_transpose_f32(const float* in, float* out, int B, int R, int Cc, void* stream) {
  return transpose_launch(in, out, B, R, Cc, stream);
}
/* fp32 in -> bf16 out (VectorNet's fp32 output becoming the bf16 map feature) and bf16 in -> fp32 out (its gradient) */
extern "C" int mmfn_transpose_f32_to_bf16(const float* in, void* out, int B, int R, int Cc, void* stream) {
  return transpose_launch(in, (mbf)out, B, R, Cc, stream);
}
extern "C" int mmfn_transpose_bf16_to_f32(const void* in, float* out, int B, int R, int Cc, void* stream) {
  return transpose_launch((cbf)in, out, B, R, Cc, stream);
}
