// Fused multi-head attention of the bf16 training mode on v_mfma_f32_32x32x16_bf16 (model_vec.py:92-109 SelfAttention, as
// torch.autocast(bfloat16) would run its two bmm's; softmax, its statistics and every accumulator stay fp32).
//
// The fusion transformers' shape is tiny for this instruction: T = 192 tokens, 4 heads of 16 / 32 / 64 / 128, 32 samples - a
// (sample, head) is 2 x 48 MFMAs per 32-query tile, ~1.3 us of matrix work.  So the kernels are built around keeping that work
// fed, not around FLOPs:
//   * one workgroup per (sample, head, half): 3 waves, each owning ONE 32-row tile (of queries in the forward and the dQ pass,
//     of keys in the dK/dV pass) against ALL tokens of the other side; 256 workgroups = one per CU, a whole SIMD's register
//     file per wave, so a tile's full score row (6 accumulator tiles) lives in registers;
//   * the two operand matrices every wave of the workgroup walks (K and V; Q and dO in the key-owned pass) are staged in LDS
//     ONCE per workgroup by global_load_lds (16 B per lane, lane-linear image, XOR slot swizzle on the SOURCE address);
//   * S^T = K Q^T puts keys in accumulator ROWS and the wave's queries in LANES: the softmax of a query is register-local
//     plus one exchange with lane ^ 32, and the probability tile is already laid out as the B operand (k = key) of the
//     second product - the 16 keys of a k-step are taken in the accumulator's own order (0-3, 8-11 | 4-7, 12-15 per half
//     wave), and the matching A operand (V^T, K^T, dO^T, Q^T: the staged matrix TRANSPOSED) is read with that key order by
//     ds_read_b64_tr_b16, the gfx950 LDS transpose read: no cross-lane traffic, no second LDS image;
//   * one slot swizzle serves both read kinds of a staged matrix (row fragments by ds_read_b128, transposed fragments by
//     tr_b16): slot ^= ((row & 3) << 2) | ((row >> 2) & 3) at head size 128, its analogues below.
// Dropout (attn_pdrop): the keep mask of element (b, h, query, key) is a 32-bit hash (lowbias32) of query * T + key, salted per
// (rng state, stream, b, h) - the same function in all three kernels.  (The engine's 64-bit splitmix counter RNG costs ~25
// integer instructions per element, a dozen of them quarter-rate multiplies: 96 elements per lane made it a third of the
// backward's time at these shapes; two 32-bit multiplies are enough for a dropout mask.)
#include <stdlib.h>

#include "attention_args.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0,
                                   0);
}

// 16-byte slot swizzle of a staged [rows][HS] bf16 matrix (CPR = HS / 8 slots per row): conflict-free for the row fragments
// (16 rows distinct mod 16 at one logical slot) AND for the transposed fragments (4 consecutive rows x 4 consecutive slots)
template <int HS>
__device__ __forceinline__ int swz(int row) {
  if (HS == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
  if (HS == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  if (HS == 32) return (row >> 2) & 3;
  return (row >> 3) & 1;
}

// stage rows [0, T) of a [*, ld] bf16 matrix (head slice of HS columns) into LDS, lane-linear 1 KB pieces
template <int HS, int T, int NW>
__device__ __forceinline__ void stage(const bf16_t* src, size_t ld, unsigned char* dst, int wave, int lane) {
  constexpr int CPR = HS / 8, RPP = 64 / CPR;          // slots per row, rows per 1 KB piece
  constexpr int PIECES = T / RPP;
  const int rl = lane / CPR, ps = lane % CPR;
  for (int p = wave; p < PIECES; p += NW) {
    const int row = p * RPP + rl;
    glds16(src + (size_t)row * ld + ((ps ^ swz<HS>(row)) << 3), dst + p * 1024);
  }
}

// A operand, rows = staged rows row0 .. row0+31 (lane l & 31), k = columns 16 * ks + 8 * (l >> 5) .. + 7
template <int HS>
__device__ __forceinline__ bf16x8 rowfrag(const unsigned char* m, int row0, int ks, int l31, int h) {
  const int row = row0 + l31;
  return *reinterpret_cast<const bf16x8*>(m + row * (HS * 2) + (((ks * 2 + h) ^ swz<HS>(row)) << 4));
}

// A operand = the staged matrix TRANSPOSED: rows = columns col0 .. col0+31 of it (lane l & 31), k = the 8 staged rows
// k0 + 4 * (l >> 5) + {0,1,2,3, 8,9,10,11}: the order in which a 32x32 accumulator holds the 16 rows of a k-step
template <int HS>
__device__ __forceinline__ bf16x8 trfrag(const unsigned char* m, int col0, int k0, int lane) {
  const int w = lane & 15, g2 = (lane >> 4) & 1, hh = lane >> 5;
  const int col = (col0 + g2 * 16 + (w & 3) * 4) & (HS - 1);   // (head size 16: the upper 16 rows of the tile are never stored)
  bf16x8 out;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = k0 + 4 * hh + 8 * q + (w >> 2);
    const unsigned char* p = m + k * (HS * 2) + ((((col >> 3) ^ swz<HS>(k)) << 4) | ((col & 7) << 1));
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
#pragma unroll
    for (int e = 0; e < 4; ++e) out[q * 4 + e] = v[e];
  }
  return out;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[8 * s + j];
  return o;
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int accrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// store a transposed result tile set: acc[dt][r] = X^T[d = 32 dt + accrow(r)][row l31]  ->  X[row][d] (bf16), 4 consecutive d per store
template <int HS>
__device__ __forceinline__ void store_rows(const f32x16* acc, bf16_t* dst /* this lane's row, head column 0 */, int h, float mul) {
  constexpr int ND = (HS + 31) / 32;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * h;
      if (d < HS) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[dt][4 * g + e] * mul;
        stx4(dst + d, v);
      }
    }
}

struct Ids { int b, hd, part; };
// bx of nb blocks -> (part fastest): the halves of a (sample, head) are neighbours; XCD-aware order keeps them on one L2
__device__ __forceinline__ Ids block_ids(int NH, int parts, int nb, int bx) {
  const int xcd = bx & 7, q = nb >> 3, r = nb & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bx >> 3);
  Ids o;
  o.part = id % parts;
  o.hd = (id / parts) % NH;
  o.b = id / (parts * NH);
  return o;
}

// ------------------------------------------------------------------------------------------ forward
template <int HS, int NKT, int NQT>
__global__ __launch_bounds__(64 * NQT) void attn16_fwd_kernel(const AttnArgs a) {
  const int nb = gridDim.x, bx = blockIdx.x;
  constexpr int T = 32 * NKT, NKS = HS / 16, ND = (HS + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + T * HS * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const Ids id = block_ids(a.NH, NKT / NQT, nb, bx);
  const size_t rowbase = (size_t)id.b * T;
  const bf16_t* q = reinterpret_cast<const bf16_t*>(a.q) + rowbase * a.ld + id.hd * HS;
  const bf16_t* k = reinterpret_cast<const bf16_t*>(a.k) + rowbase * a.ld + id.hd * HS;
  const bf16_t* v = reinterpret_cast<const bf16_t*>(a.v) + rowbase * a.ld + id.hd * HS;
  stage<HS, T, NQT>(k, a.ld, Ks, wave, lane);
  stage<HS, T, NQT>(v, a.ld, Vs, wave, lane);
  const int query = (id.part * NQT + wave) * 32 + l31;
  bf16x8 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(q + (size_t)query * a.ld + ks * 16 + h * 8);
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): the staged pieces and the Q fragments
  __syncthreads();
  const int kvlen = a.kv_len ? min(T, a.kv_len[id.b]) : T;
  f32x16 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) s[kt] = mfma(rowfrag<HS>(Ks, kt * 32, ks, l31, h), qf[ks], s[kt]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + accrow(r, h);
      // no valid key (kv_len == 0): the reference's masked_fill(-1e9) + softmax degrades to uniform attention over all T keys
      const float x = kvlen <= 0 ? 0.f : (key < kvlen ? s[kt][r] * a.scale : -INFINITY);
      s[kt][r] = x;
      mx = fmaxf(mx, x);
    }
  mx = xhalf_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = mmfn_exp(s[kt][r] - mx);
      s[kt][r] = p;
      sum += p;
    }
  sum = xhalf_sum(sum);
  if (a.lse && h == 0) a.lse[((size_t)id.b * a.NH + id.hd) * T + query] = mx + logf(sum);
  Drop dr;
  dr.init(a, id.b, id.hd);
  if (dr.on) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] *= dr.scale(query, kt * 32 + accrow(r, h), T);
  }
  f32x16 o[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int sgrp = 0; sgrp < 2; ++sgrp) {
      const bf16x8 pb = pack8(s[kt], sgrp);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) o[dt] = mfma(trfrag<HS>(Vs, dt * 32, kt * 32 + 16 * sgrp, lane), pb, o[dt]);
    }
  store_rows<HS>(o, reinterpret_cast<bf16_t*>(a.o) + (rowbase + query) * a.ldo + id.hd * HS, h, 1.0f / sum);
}

// ------------------------------------------------------------------------------------------ backward, query-owned: dQ (+ delta)
template <int HS, int NKT, int NQT>
__device__ __forceinline__ void attn16_dq_body(const AttnArgs& a, unsigned char* smem, int nb, int bx) {
  constexpr int T = 32 * NKT, NKS = HS / 16, ND = (HS + 31) / 32;
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + T * HS * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const Ids id = block_ids(a.NH, NKT / NQT, nb, bx);
  const size_t rowbase = (size_t)id.b * T;
  const bf16_t* q = reinterpret_cast<const bf16_t*>(a.q) + rowbase * a.ld + id.hd * HS;
  const bf16_t* k = reinterpret_cast<const bf16_t*>(a.k) + rowbase * a.ld + id.hd * HS;
  const bf16_t* v = reinterpret_cast<const bf16_t*>(a.v) + rowbase * a.ld + id.hd * HS;
  const bf16_t* o = reinterpret_cast<const bf16_t*>(a.o) + rowbase * a.ldo + id.hd * HS;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dO) + rowbase * a.ldo + id.hd * HS;
  stage<HS, T, NQT>(k, a.ld, Ks, wave, lane);
  stage<HS, T, NQT>(v, a.ld, Vs, wave, lane);
  const int query = (id.part * NQT + wave) * 32 + l31;
  bf16x8 qf[NKS], dof[NKS];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(q + (size_t)query * a.ld + ks * 16 + h * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dO + (size_t)query * a.ldo + ks * 16 + h * 8);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + (size_t)query * a.ldo + ks * 16 + h * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[j];
  }
  delta = xhalf_sum(delta);   // = sum_k P dP (the dropout mask sits inside O)
  const size_t stat = ((size_t)id.b * a.NH + id.hd) * T + query;
  if (h == 0) a.delta[stat] = delta;
  const float lse = a.lse[stat];
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  const int kvlen = a.kv_len ? min(T, a.kv_len[id.b]) : T;
  Drop dr;
  dr.init(a, id.b, id.hd);
  f32x16 dq[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      s = mfma(rowfrag<HS>(Ks, kt * 32, ks, l31, h), qf[ks], s);
      dp = mfma(rowfrag<HS>(Vs, kt * 32, ks, l31, h), dof[ks], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + accrow(r, h);
      const float p = kvlen <= 0 ? 1.0f / (float)T : (key < kvlen ? mmfn_exp(s[r] * a.scale - lse) : 0.f);
      float g = dp[r];
      if (dr.on) g *= dr.scale(query, key, T);
      s[r] = (kvlen <= 0 ? 0.f : p * (g - delta)) * a.scale;   // (no valid key: the scores are constants, no gradient reaches q / k)
    }
#pragma unroll
    for (int sgrp = 0; sgrp < 2; ++sgrp) {
      const bf16x8 b = pack8(s, sgrp);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) dq[dt] = mfma(trfrag<HS>(Ks, dt * 32, kt * 32 + 16 * sgrp, lane), b, dq[dt]);
    }
  }
  store_rows<HS>(dq, reinterpret_cast<bf16_t*>(a.dq) + (rowbase + query) * a.ldg + id.hd * HS, h, 1.0f);
}
template <int HS, int NKT, int NQT>
__global__ __launch_bounds__(64 * NQT) void attn16_dq_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn16_dq_body<HS, NKT, NQT>(a, smem, gridDim.x, blockIdx.x);
}

// ------------------------------------------------------------------------------------------ backward, key-owned: dK, dV
// OWN_DELTA: delta of every query is formed here from dO and O (the sums in the query-owned pass's order: bit-identical to what
// that pass writes to a.delta) instead of read back - the two passes then have no dependence and share one launch (attn16_bwd_kernel)
template <int HS, int NKT, int NQT, bool OWN_DELTA>
__device__ __forceinline__ void attn16_dkv_body(const AttnArgs& a, unsigned char* smem, int nb, int bx) {
  constexpr int T = 32 * NKT, NKS = HS / 16, ND = (HS + 31) / 32;
  unsigned char* Qs = smem;
  unsigned char* Gs = smem + T * HS * 2;   // dO
  float* stats = reinterpret_cast<float*>(smem + 2 * T * HS * 2);   // [2][T]: lse, delta of every query (read 16 + 16 times per tile per lane)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const Ids id = block_ids(a.NH, NKT / NQT, nb, bx);
  const size_t rowbase = (size_t)id.b * T;
  const bf16_t* q = reinterpret_cast<const bf16_t*>(a.q) + rowbase * a.ld + id.hd * HS;
  const bf16_t* k = reinterpret_cast<const bf16_t*>(a.k) + rowbase * a.ld + id.hd * HS;
  const bf16_t* v = reinterpret_cast<const bf16_t*>(a.v) + rowbase * a.ld + id.hd * HS;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dO) + rowbase * a.ldo + id.hd * HS;
  stage<HS, T, NQT>(q, a.ld, Qs, wave, lane);
  stage<HS, T, NQT>(dO, a.ldo, Gs, wave, lane);
  const size_t stat0 = ((size_t)id.b * a.NH + id.hd) * T;
  for (int t = threadIdx.x; t < T; t += 64 * NQT) {
    stats[t] = a.lse[stat0 + t];
    if (OWN_DELTA) {
      const bf16_t* o = reinterpret_cast<const bf16_t*>(a.o) + rowbase * a.ldo + id.hd * HS;
      float dh[2] = {0.f, 0.f};
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 df = *reinterpret_cast<const bf16x8*>(dO + (size_t)t * a.ldo + ks * 16 + hh * 8);
          const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + (size_t)t * a.ldo + ks * 16 + hh * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) dh[hh] += (float)df[j] * (float)of[j];
        }
      stats[T + t] = dh[0] + dh[1];
    } else {
      stats[T + t] = a.delta[stat0 + t];
    }
  }
  const int key = (id.part * NQT + wave) * 32 + l31;
  bf16x8 kf[NKS], vf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(k + (size_t)key * a.ld + ks * 16 + h * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(v + (size_t)key * a.ld + ks * 16 + h * 8);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  const int kvlen = a.kv_len ? min(T, a.kv_len[id.b]) : T;
  const bool valid = key < kvlen;
  Drop dr;
  dr.init(a, id.b, id.hd);
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      s = mfma(rowfrag<HS>(Qs, qt * 32, ks, l31, h), kf[ks], s);     // S[query][key]: queries in accumulator rows, this wave's keys in lanes
      dp = mfma(rowfrag<HS>(Gs, qt * 32, ks, l31, h), vf[ks], dp);
    }
    f32x16 pd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int query = qt * 32 + accrow(r, h);
      const float lse = stats[query], delta = stats[T + query];
      const float p = kvlen <= 0 ? 1.0f / (float)T : (valid ? mmfn_exp(s[r] * a.scale - lse) : 0.f);
      float m = 1.f;
      if (dr.on) m = dr.scale(query, key, T);
      pd[r] = p * m;
      s[r] = (kvlen <= 0 ? 0.f : p * (dp[r] * m - delta)) * a.scale;
    }
#pragma unroll
    for (int sgrp = 0; sgrp < 2; ++sgrp) {
      const bf16x8 pb = pack8(pd, sgrp), sb = pack8(s, sgrp);
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        dv[dt] = mfma(trfrag<HS>(Gs, dt * 32, qt * 32 + 16 * sgrp, lane), pb, dv[dt]);
        dk[dt] = mfma(trfrag<HS>(Qs, dt * 32, qt * 32 + 16 * sgrp, lane), sb, dk[dt]);
      }
    }
  }
  store_rows<HS>(dv, reinterpret_cast<bf16_t*>(a.dv) + (rowbase + key) * a.ldg + id.hd * HS, h, 1.0f);
  store_rows<HS>(dk, reinterpret_cast<bf16_t*>(a.dk) + (rowbase + key) * a.ldg + id.hd * HS, h, 1.0f);
}
template <int HS, int NKT, int NQT>
__global__ __launch_bounds__(64 * NQT) void attn16_dkv_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn16_dkv_body<HS, NKT, NQT, false>(a, smem, gridDim.x, blockIdx.x);
}

// Both backward passes in ONE launch of twice the blocks: block role = query-owned (dQ, delta) or key-owned (dK, dV).  The passes
// are independent once the key-owned one forms delta itself, so at head sizes up to 64 (two or more workgroups per CU) they run side
// by side instead of one after the other, and every length saves a launch on the transformers' dependent chain.  Blocks 16 j .. 16 j + 7
// take the query-owned role and 16 j + 8 .. 16 j + 15 the key-owned role of the SAME eight ids: a pair shares its XCD (block b runs on
// XCD b % 8) and therefore the L2 lines of q / k / v / dO.
template <int HS, int NKT, int NQT>
__global__ __launch_bounds__(64 * NQT) void attn16_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = gridDim.x >> 1, gx = blockIdx.x;
  int role, bx;
  if ((nb & 7) == 0) { role = (gx >> 3) & 1; bx = (gx & 7) | ((gx >> 4) << 3); }
  else { role = gx >= nb; bx = gx - role * nb; }
  if (role == 0) attn16_dq_body<HS, NKT, NQT>(a, smem, nb, bx);
  else attn16_dkv_body<HS, NKT, NQT, true>(a, smem, nb, bx);
}

template <int HS, int NKT>
int launch16(int which, const AttnArgs& a, hipStream_t s) {
  constexpr int NQT = (NKT % 2 == 0) ? NKT / 2 : NKT;
  constexpr int smem = 2 * 32 * NKT * HS * 2 + 2 * 32 * NKT * 4;   // two staged matrices (+ lse / delta in the key-owned pass)
  const dim3 grid(a.B * a.NH * (NKT / NQT)), block(64 * NQT);
  static bool ready[4] = {false, false, false, false};
  const void* fn = which == 0 ? reinterpret_cast<const void*>(&attn16_fwd_kernel<HS, NKT, NQT>)
                 : which == 1 ? reinterpret_cast<const void*>(&attn16_dq_kernel<HS, NKT, NQT>)
                 : which == 2 ? reinterpret_cast<const void*>(&attn16_dkv_kernel<HS, NKT, NQT>)
                              : reinterpret_cast<const void*>(&attn16_bwd_kernel<HS, NKT, NQT>);
  if (!ready[which]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return MMFN_EINVAL;
    ready[which] = true;
  }
  if (which == 0) hipLaunchKernelGGL((attn16_fwd_kernel<HS, NKT, NQT>), grid, block, smem, s, a);
  else if (which == 1) hipLaunchKernelGGL((attn16_dq_kernel<HS, NKT, NQT>), grid, block, smem, s, a);
  else if (which == 2) hipLaunchKernelGGL((attn16_dkv_kernel<HS, NKT, NQT>), grid, block, smem, s, a);
  else hipLaunchKernelGGL((attn16_bwd_kernel<HS, NKT, NQT>), dim3(2 * grid.x), block, smem, s, a);
  MMFN_LAUNCH_CHECK();
  return 0;
}

template <int HS>
int by_tokens16(int which, const AttnArgs& a, hipStream_t s) {
  switch (a.T) {
    case 64: return launch16<HS, 2>(which, a, s);
    case 128: return launch16<HS, 4>(which, a, s);
    case 192: return launch16<HS, 6>(which, a, s);
    case 256: return launch16<HS, 8>(which, a, s);   // the rad variant's deepest fusion / two views: 131 KB of staged K + V at head size 128
  }
  return -1;
}

}  // namespace

// which: 0 forward, 1 dQ (+ delta), 2 dK / dV (after 1: reads delta), 3 both backward passes in one launch.  -1: shape not covered (the caller falls back to the fp32-arithmetic kernels).
int mmfn_attn16_launch(int which, int hs, const AttnArgs& a, hipStream_t s) {
  if ((a.ld & 7) || (a.ldo & 7) || (which > 0 && (a.ldg & 7))) return -1;
  // one launch for both backward passes where two workgroups share a CU (measured in the step's shapes, dQ + dK/dV -> merged:
  // head size 16: 17.2 + 16.6 -> 21.3 us, 32: 18.6 + 18.1 -> 23.3, 64: 20.8 + 20.9 -> 29.4); at head size 128 one workgroup's K + V
  // fill the LDS, the roles run in two rounds and the launch is slower than the two it replaces (26.2 + 28.1 -> 59.5)
  if (which == 3 && hs > 64) return -1;
  switch (hs) {
    case 16: return by_tokens16<16>(which, a, s);
    case 32: return by_tokens16<32>(which, a, s);
    case 64: return by_tokens16<64>(which, a, s);
    case 128: return by_tokens16<128>(which, a, s);
  }
  return -1;
}
