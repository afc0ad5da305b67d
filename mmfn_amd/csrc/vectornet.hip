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
// ---- lane attention of VectorNet, query 0 only -----------------------------------------------------------------------
// MaskSelfAttention (model_vec.py:286-324) computes all L x L lane interactions, but VectornetEncoder.forward keeps only
// lane 0's fused token (`agent_token_fuse[:, 0, :]`, :412): every other query row is dead.  So the attention runs for
// query 0 alone - one wave per (sample, head), lanes over the keys, ANY number of lanes (the fused attention kernels stop
// at 256 tokens), O(L) instead of O(L^2).  Softmax semantics as in the reference: keys >= lane_num masked (-1e9), a sample
// without lanes degrades to uniform attention.  qkv: [B*L, 3*D] = [q | k | v], D = heads * HD; HD = 64.
constexpr int L0_HD = 64;

__global__ __launch_bounds__(64) void lane0_attn_fwd_kernel(const float* __restrict__ qkv, const int* __restrict__ kv_len, int L,
                                                            int heads, float scale, float* __restrict__ att0 /* [B, heads*HD] */,
                                                            float* __restrict__ prob /* [B, heads, L] */) {
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads, lane = threadIdx.x;
  const int D = heads * L0_HD, ld = 3 * D;
  const float* base = qkv + (size_t)b * L * ld;
  __shared__ float q0[L0_HD];
  q0[lane] = base[hd * L0_HD + lane];   // row 0 = lane 0 of the sample (blockDim == HD == 64)
  __syncthreads();
  const int n = kv_len ? min(L, kv_len[b]) : L;
  const bool nokeys = n <= 0;
  float* P = prob + ((size_t)b * heads + hd) * L;
  // pass 1: scores -> P (unnormalised exponent input), running max
  float mx = -INFINITY;
  for (int j = lane; j < L; j += 64) {
    const float* kr = base + (size_t)j * ld + D + hd * L0_HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < L0_HD; c += 4) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + c);
      s += q0[c] * kv[0] + q0[c + 1] * kv[1] + q0[c + 2] * kv[2] + q0[c + 3] * kv[3];
    }
    s = nokeys ? 0.f : (j < n ? s * scale : -INFINITY);
    P[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {   // every lane re-reads only what it wrote itself
    const float e = mmfn_exp(P[j] - mx);
    P[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < L; j += 64) P[j] *= inv;   // normalised probabilities, kept for the backward
  __syncthreads();   // one wave per block: makes the other lanes' P visible (stores drained) before the cross-lane reads
  // pass 2: o = sum_j P_j V_j; lane c accumulates output column c over all keys (rows are 256-byte coalesced reads)
  float o = 0.f;
  for (int j = 0; j < L; ++j) o += P[j] * base[(size_t)j * ld + 2 * D + hd * L0_HD + lane];
  att0[(size_t)b * D + hd * L0_HD + lane] = o;
}

// dqkv [B*L, 3*D] is written completely: dq for row 0, zero for the other query rows, dk / dv for every lane.
__global__ __launch_bounds__(64) void lane0_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ prob,
                                                            const float* __restrict__ g_att0, const int* __restrict__ kv_len,
                                                            int L, int heads, float scale, float* __restrict__ dqkv) {
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads, lane = threadIdx.x;
  const int D = heads * L0_HD, ld = 3 * D;
  const float* base = qkv + (size_t)b * L * ld;
  float* dbase = dqkv + (size_t)b * L * ld;
  const float* P = prob + ((size_t)b * heads + hd) * L;
  __shared__ float q0[L0_HD], go[L0_HD];
  q0[lane] = base[hd * L0_HD + lane];
  go[lane] = g_att0[(size_t)b * D + hd * L0_HD + lane];
  __syncthreads();
  const bool nokeys = (kv_len ? min(L, kv_len[b]) : L) <= 0;   // constant scores: no gradient to q, k
  // dP_j = go . V_j ;  delta = sum_j P_j dP_j
  float delta = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float* vr = base + (size_t)j * ld + 2 * D + hd * L0_HD;
    float dp = 0.f;
#pragma unroll
    for (int c = 0; c < L0_HD; c += 4) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + c);
      dp += go[c] * vv[0] + go[c + 1] * vv[1] + go[c + 2] * vv[2] + go[c + 3] * vv[3];
    }
    delta += P[j] * dp;
  }
  delta = wave_sum(delta);
  float dq = 0.f;
  for (int j = 0; j < L; ++j) {
    const float p = P[j];
    const float* kr = base + (size_t)j * ld + D + hd * L0_HD;
    const float* vr = base + (size_t)j * ld + 2 * D + hd * L0_HD;
    // recompute dP_j across the wave (lane c holds column c)
    float dp = wave_sum(go[lane] * vr[lane]);
    const float ds = nokeys ? 0.f : p * (dp - delta) * scale;
    dq += ds * kr[lane];
    float* dr = dbase + (size_t)j * ld;
    dr[D + hd * L0_HD + lane] = ds * q0[lane];          // dK_j
    dr[2 * D + hd * L0_HD + lane] = p * go[lane];       // dV_j
    if (j > 0) dr[hd * L0_HD + lane] = 0.f;             // dead query rows
  }
  dbase[hd * L0_HD + lane] = dq;
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

extern "C" int mmfn_lane0_attention_fwd_f32(const float* qkv, const int32_t* kv_len, int B, int L, int heads, int head_dim,
                                            float scale, float* att0, float* prob, void* stream) {
  if (!qkv || !att0 || !prob || B <= 0 || L <= 0 || heads <= 0 || head_dim != L0_HD) return MMFN_EINVAL;
  hipLaunchKernelGGL(lane0_attn_fwd_kernel, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, qkv, kv_len, L, heads, scale, att0,
                     prob);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_lane0_attention_bwd_f32(const float* qkv, const float* prob, const float* g_att0, const int32_t* kv_len, int B,
                                            int L, int heads, int head_dim, float scale, float* dqkv, void* stream) {
  if (!qkv || !prob || !g_att0 || !dqkv || B <= 0 || L <= 0 || heads <= 0 || head_dim != L0_HD) return MMFN_EINVAL;
  hipLaunchKernelGGL(lane0_attn_bwd_kernel, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, qkv, prob, g_att0, kv_len, L, heads,
                     scale, dqkv);
  MMFN_LAUNCH_CHECK();
  return 0;
}
