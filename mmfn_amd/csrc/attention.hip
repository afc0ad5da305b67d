// Fused multi-head attention for the GPT fusion transformers and VectorNet's lane attention,
// fp32 on v_mfma_f32_32x32x2_f32 (model_vec.py:92-109 SelfAttention, :301-324 MaskSelfAttention).
//
// Shapes are tiny by GEMM standards (T = 192 or 256 tokens, head size 16..128, B*heads = 128
// problems), so the unit of work is one wave = one 32-query tile against ALL keys:
//   S^T[key, q] = K . Q^T      keys land in accumulator ROWS, queries in LANES, so the softmax
//                              of a query is a register-local reduction (+ one cross-half swap),
//   O^T[d, q]   = V^T . P^T    the accumulator tile of P^T is already laid out as the MFMA B
//                              operand (row index <-> k slot), so probabilities never leave
//                              registers: no LDS, no barriers, no [B,h,T,T] tensor in HBM.
// K/V/Q fragments stream straight from L2 (the whole qkv of a sample is 1.2 MB).  The backward
// runs two such passes (query-owned: dQ; key-owned: dK, dV) that recompute P from the saved
// row log-sum-exp instead of storing it; dropout masks are regenerated from the counter RNG.
#include <stdlib.h>

#include "common.h"
#include "attention_args.h"

namespace {

__device__ __forceinline__ int rowmap(int r) { return (r & 3) + 8 * (r >> 2); }
__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Row fragments of the second GEMM of each pass (V, K, dO, Q rows indexed by head dimension d): lane l31 owns the
// VW = HS/32 consecutive columns d = VW*l31 .. VW*l31+VW-1 and column j feeds output tile j, so a row costs one
// 4/8/16-byte load per lane and the result tile stores back as one vector per lane (rows of O/dQ/dK/dV are
// written as contiguous 128..512-byte runs instead of 64 scattered dwords per instruction).
template <int VW>
__device__ __forceinline__ void ldrow(const float* p, bool ok, float* dst) {
  if (VW == 4) {
    const f32x4 t = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
    dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3];
  } else if (VW == 2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 t = ok ? *reinterpret_cast<const f32x2*>(p) : f32x2{0.f, 0.f};
    dst[0] = t[0]; dst[1] = t[1];
  } else {
    dst[0] = ok ? p[0] : 0.f;
  }
}
// Fragments both waves of a block consume (the query tile's Q / dO rows, or the key tile's K / V rows) live in LDS
// rather than in each wave's registers - that is what lets two waves share a SIMD.  Row pitch HS+4 floats: the 16-byte
// slots of consecutive rows fall into distinct banks, so the per-lane ds_read_b128 of row l31 is conflict-free.
template <int HS>
__device__ __forceinline__ void stage_rows(float* sm, const float* src, size_t ld, int row0, int T, int tid) {
  constexpr int Q4 = HS / 4;
#pragma unroll
  for (int u = tid; u < 32 * Q4; u += 128) {
    const int r = u / Q4, c4 = u % Q4;
    *reinterpret_cast<f32x4*>(&sm[r * (HS + 4) + 4 * c4]) = ld4g(src + (size_t)min(row0 + r, T - 1) * ld + 4 * c4);
  }
}

// Partial result tiles of the two waves of a block -> their sum, each wave finishing half of the accumulator rows.
// W is the compile-time wave id so every register index stays static.  smem: [2][ND*8*64] floats.
template <int ND, int W, typename F>
__device__ __forceinline__ void merge_rows(const f32x16* acc, float* smem, int lane, F&& emit) {
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) smem[W * ND * 512 + (dt * 8 + r8) * 64 + lane] = acc[dt][(1 - W) * 8 + r8];
  __syncthreads();
#pragma unroll
  for (int r8 = 0; r8 < 8; ++r8) {
    float t[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) t[dt] = acc[dt][W * 8 + r8] + smem[(1 - W) * ND * 512 + (dt * 8 + r8) * 64 + lane];
    emit(W * 8 + r8, t);
  }
}
template <int ND, typename F>
__device__ __forceinline__ void merge_store(const f32x16* acc, float* smem, int w, int lane, F&& emit) {
  if (w == 0) merge_rows<ND, 0>(acc, smem, lane, emit);
  else merge_rows<ND, 1>(acc, smem, lane, emit);
}

template <int VW>
__device__ __forceinline__ void strow(float* p, const float* src) {
  if (VW == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{src[0], src[1], src[2], src[3]};
  } else if (VW == 2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f32x2*>(p) = f32x2{src[0], src[1]};
  } else {
    p[0] = src[0];
  }
}

// Register budget: up to 8 key tiles (256 keys) the kernels fit 256 VGPRs at two waves per SIMD.  The 12-tile instantiations
// (257-384 keys: several frames per sample) keep 2 x 6 accumulator tiles of S^T / dP^T live and spilled 132-368 bytes per lane to
// scratch in rounds 1-3; they are compiled for ONE wave per SIMD instead (amdgpu_waves_per_eu(1, 1)): the unified register file
// then gives a wave 512 registers and the overflow lives in AGPRs - ScratchSize 0 for every kernel of this file
// (-Rpass-analysis=kernel-resource-usage).  Their grids (12 query tiles x heads x batch blocks of two waves) are about one wave
// per SIMD anyway.
// Two waves per 32-query tile, each owning half of the key tiles: the per-wave state halves (S^T is NKT/2
// accumulator tiles), so two waves fit on a SIMD and one wave's softmax / load latency hides under the other's
// MFMAs.  The halves meet through LDS: (row max, row sum) first, then the partial O tiles, merged flash-style.
template <int HS, int NKT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NKT > 8 ? 1 : 2, NKT > 8 ? 1 : 2))) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int NC = HS / 8;
  constexpr int ND = (HS + 31) / 32;
  constexpr int NK2 = NKT / 2;
  __shared__ float sm_ml[2][2][32];
  __shared__ float sm_o[2][ND * 8 * 64];
  const int qt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int ktb = w * NK2;
  const int T = a.T;
  const int q = qt * 32 + l31;
  const bool qvalid = q < T;
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const size_t rowbase = (size_t)b * T;
  const float* Qp = a.q + (rowbase + min(q, T - 1)) * a.ld + hd * HS + 4 * h;
  f32x4 qf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) qf[c] = ld4g(Qp + 8 * c);

  f32x16 s[NK2];
#pragma unroll
  for (int kt = 0; kt < NK2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
  // L2 latency is hidden inside the wave as well: the K fragments are software-pipelined PF chunks (8 d-columns
  // each) ahead of the MFMAs that consume them.  hipcc does not pipeline this on its own (it emits load;
  // s_waitcnt vmcnt(0); 4 MFMAs), hence the explicit ring and the sched_barrier pins.
  {
    constexpr int PF = (NC < 8 ? NC : 8), TOT = NK2 * NC;
    f32x4 ring[PF];
    auto kptr = [&](int i) {
      const int kt = ktb + i / NC, c = i % NC;
      return a.k + (rowbase + min(kt * 32 + l31, T - 1)) * a.ld + hd * HS + 4 * h + 8 * c;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = ld4g(kptr(i));
#pragma unroll
    for (int i = 0; i < TOT; ++i) {
      const f32x4 kf = ring[i % PF];
      if (i + PF < TOT) ring[i % PF] = ld4g(kptr(i + PF));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s[i / NC] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[i % NC][j], s[i / NC], 0, 0, 0);
    }
  }
  // ---- softmax over keys (rows of S^T): register-local + one cross-half exchange, then the two key halves
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NK2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (ktb + kt) * 32 + rowmap(r) + 4 * h;
      // kv_len == 0 (a sample without lanes): the reference's masked_fill(-1e9) + softmax degrades to UNIFORM attention
      // over all T keys (model_vec.py:315-317) and stays finite; -inf everywhere would give 0/0
      const float v = nokeys ? (key < T ? 0.f : -INFINITY) : (key < kvlen ? s[kt][r] * a.scale : -INFINITY);
      s[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float msafe = mx > -INFINITY ? mx : 0.f;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NK2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = mmfn_exp(s[kt][r] - msafe);  // branch-free; masked keys and fully masked halves give exp(-inf) = 0
      s[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  if (h == 0) { sm_ml[w][0][l31] = mx; sm_ml[w][1][l31] = sum; }
  __syncthreads();
  float pscale;
  {
    const float mo = sm_ml[w ^ 1][0][l31], lo = sm_ml[w ^ 1][1][l31];
    const float m = fmaxf(mx, mo);
    const float fw = mmfn_exp(mx - m);
    const float fo = mmfn_exp(mo - m);
    const float l = sum * fw + lo * fo;
    if (w == 0 && qvalid && h == 0 && a.lse) a.lse[((size_t)b * a.NH + hd) * T + q] = m + logf(l);
    pscale = fw / l;
  }
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
  const uint64_t pbase = (((uint64_t)b * a.NH + hd) * T + q) * (uint64_t)T;
#pragma unroll
  for (int kt = 0; kt < NK2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = s[kt][r] * pscale;
      if (drop) p *= mmfn_dropout_scale(key64, pbase + (uint64_t)((ktb + kt) * 32 + rowmap(r) + 4 * h), a.drop_p, inv_keep);
      s[kt][r] = p;
    }
  // ---- O = P . V   (P^T's accumulator tile read as the A operand: row = query lane, k slot = key row)
  const bool dok = ND * l31 < HS;
  f32x16 o[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  {
    // V rows are consumed one accumulator row (r) at a time; prefetch them in groups of 4 rows, one group ahead
    constexpr int G = NK2 * 4;
    float vv[2][4][ND];
    auto vload = [&](int g, float (*dst)[ND]) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int kt = ktb + (g >> 2), r = (g & 3) * 4 + rr;
        const int key = min(kt * 32 + rowmap(r) + 4 * h, T - 1);
        ldrow<ND>(a.v + (rowbase + key) * a.ld + hd * HS + ND * l31, dok, dst[rr]);
      }
    };
    vload(0, vv[0]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) vload(g + 1, vv[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[g >> 2][(g & 3) * 4 + rr], vv[g & 1][rr][dt], o[dt], 0, 0, 0);
    }
  }
  // ---- merge: wave w finishes accumulator rows [8w, 8w+8); it hands the other eight rows to its partner
  merge_store<ND>(o, sm_o[0], w, lane, [&](int r, const float* t) {
    const int qr = qt * 32 + rowmap(r) + 4 * h;
    if (qr < T && dok) strow<ND>(a.o + (rowbase + qr) * a.ldo + hd * HS + ND * l31, t);
  });
}

// query-owned backward pass: delta = sum_j P_j dP_j, dQ.  Same two-wave key split as the forward: the halves
// exchange their partial (sum P dP, sum P) before forming dS, and their partial dQ tiles at the end.
template <int HS, int NKT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NKT > 8 ? 1 : 2, NKT > 8 ? 1 : 2))) void attn_bwd_dq_kernel(const AttnArgs a) {
  constexpr int NC = HS / 8;
  constexpr int ND = (HS + 31) / 32;
  constexpr int NK2 = NKT / 2;
  constexpr int P = HS + 4;
  __shared__ float sm_dl[2][2][32];
  __shared__ float sm_o[2][ND * 8 * 64];
  __shared__ __attribute__((aligned(16))) float sm_q[32 * P];
  __shared__ __attribute__((aligned(16))) float sm_do[32 * P];
  const int qt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int ktb = w * NK2;
  const int T = a.T;
  const int q = qt * 32 + l31;
  const bool qvalid = q < T;
  const int qc = min(q, T - 1);
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;  // uniform attention whose scores are constants: P = 1/T, dS = 0 (see attn_fwd_kernel)
  const size_t rowbase = (size_t)b * T;
  stage_rows<HS>(sm_q, a.q + rowbase * a.ld + hd * HS, a.ld, qt * 32, T, threadIdx.x);
  stage_rows<HS>(sm_do, a.dO + rowbase * a.ldo + hd * HS, a.ldo, qt * 32, T, threadIdx.x);
  __syncthreads();
  const float* qrow = sm_q + l31 * P + 4 * h;
  const float* dorow = sm_do + l31 * P + 4 * h;
  const size_t statoff = ((size_t)b * a.NH + hd) * T + qc;
  const float lse = a.lse[statoff];
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
  const uint64_t pbase = (((uint64_t)b * a.NH + hd) * T + q) * (uint64_t)T;

  // P^T and dP^T tiles stay in registers; delta = sum_j P_j dP_j is formed from them directly
  // (the same cancellation structure as softmax_backward: more accurate than rowsum(dO*O) when
  // attention is near-uniform and dS is a small difference of large terms)
  f32x16 ds[NK2], dpt[NK2];
  float dl = 0.f, psum = 0.f;
  {
    // chunk-granular software pipeline of the K and V fragments (see attn_fwd_kernel)
    constexpr int PF = (NC < 4 ? NC : 4), TOT = NK2 * NC;
    f32x4 rk[PF], rv[PF];
    auto off = [&](int i) {
      const int kt = ktb + i / NC, c = i % NC;
      return (rowbase + min(kt * 32 + l31, T - 1)) * a.ld + hd * HS + 4 * h + 8 * c;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) { rk[i] = ld4g(a.k + off(i)); rv[i] = ld4g(a.v + off(i)); }
#pragma unroll
    for (int kt = 0; kt < NK2; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = kt * NC + c;
        const f32x4 kf = rk[i % PF], vf = rv[i % PF];
        if (i + PF < TOT) { rk[i % PF] = ld4g(a.k + off(i + PF)); rv[i % PF] = ld4g(a.v + off(i + PF)); }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 qfc = *reinterpret_cast<const f32x4*>(qrow + 8 * c);
        const f32x4 dofc = *reinterpret_cast<const f32x4*>(dorow + 8 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qfc[j], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], dofc[j], dp, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (ktb + kt) * 32 + rowmap(r) + 4 * h;
        const float ex = mmfn_exp((nokeys ? 0.f : st[r] * a.scale) - lse);
        const float p = (nokeys ? key < T : key < kvlen) ? ex : 0.f;
        float dpv = dp[r];
        if (drop) dpv *= mmfn_dropout_scale(key64, pbase + (uint64_t)key, a.drop_p, inv_keep);
        ds[kt][r] = p;
        dpt[kt][r] = dpv;
        dl += p * dpv;
        psum += p;
      }
    }
  }
  // P is recomputed from the rounded log-sum-exp, so sum_j P_j = 1 + O(1e-6); dividing by it keeps
  // sum_j dS_j = 0 to rounding (otherwise a common-mode term eps*delta*P leaks into dQ/dK, which
  // dominates when attention is near-uniform and |dP - delta| << |delta|)
  dl += __shfl_xor(dl, 32, 64);
  psum += __shfl_xor(psum, 32, 64);
  if (h == 0) { sm_dl[w][0][l31] = dl; sm_dl[w][1][l31] = psum; }
  __syncthreads();
  // both waves add the halves in the same order so they agree on delta bit for bit
  const float delta = (sm_dl[0][0][l31] + sm_dl[1][0][l31]) / (sm_dl[0][1][l31] + sm_dl[1][1][l31]);
  if (w == 0 && qvalid && h == 0) a.delta[statoff] = delta;
#pragma unroll
  for (int kt = 0; kt < NK2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[kt][r] = nokeys ? 0.f : ds[kt][r] * (dpt[kt][r] - delta) * a.scale;
  const bool dok = ND * l31 < HS;
  f32x16 dq[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  {
    constexpr int G = NK2 * 4;
    float kk[2][4][ND];
    auto kload = [&](int g, float (*dst)[ND]) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int kt = ktb + (g >> 2), r = (g & 3) * 4 + rr;
        const int key = min(kt * 32 + rowmap(r) + 4 * h, T - 1);
        ldrow<ND>(a.k + (rowbase + key) * a.ld + hd * HS + ND * l31, dok, dst[rr]);
      }
    };
    kload(0, kk[0]);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) kload(g + 1, kk[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[g >> 2][(g & 3) * 4 + rr], kk[g & 1][rr][dt], dq[dt], 0, 0, 0);
    }
  }
  merge_store<ND>(dq, sm_o[0], w, lane, [&](int r, const float* t) {
    const int qr = qt * 32 + rowmap(r) + 4 * h;
    if (qr < T && dok) strow<ND>(a.dq + (rowbase + qr) * a.ldg + hd * HS + ND * l31, t);
  });
}

// key-owned backward pass: dK, dV.  Two waves per 32-key tile, each owning half of the query tiles; the partial
// dK / dV tiles are summed through LDS.
template <int HS, int NKT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NKT > 8 ? 1 : 2, NKT > 8 ? 1 : 2))) void attn_bwd_dkv_kernel(const AttnArgs a) {
  constexpr int NC = HS / 8;
  constexpr int ND = (HS + 31) / 32;
  constexpr int NK2 = NKT / 2;
  constexpr int P = HS + 4;
  // K / V fragments of the key tile while the query loop runs; the dK / dV merge buffers afterwards (same bytes)
  constexpr int STAGE = 2 * 32 * P, MERGE = 4 * ND * 8 * 64;
  __shared__ __attribute__((aligned(16))) float smem[STAGE > MERGE ? STAGE : MERGE];
  float* sm_kf = smem;
  float* sm_vf = smem + 32 * P;
  float* sm_k = smem;
  float* sm_v = smem + 2 * ND * 8 * 64;
  const int kt0 = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int T = a.T;
  const int key = kt0 * 32 + l31;
  const bool kvalid = key < T;
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const bool kin = nokeys ? kvalid : key < kvlen;
  const size_t rowbase = (size_t)b * T;
  stage_rows<HS>(sm_kf, a.k + rowbase * a.ld + hd * HS, a.ld, kt0 * 32, T, threadIdx.x);
  stage_rows<HS>(sm_vf, a.v + rowbase * a.ld + hd * HS, a.ld, kt0 * 32, T, threadIdx.x);
  __syncthreads();
  const float* krow = sm_kf + l31 * P + 4 * h;
  const float* vrow = sm_vf + l31 * P + 4 * h;
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
  const bool dok = ND * l31 < HS;
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const size_t statbase = ((size_t)b * a.NH + hd) * T;
  for (int qi = 0; qi < NK2; ++qi) {
    const int qt = w * NK2 + qi;
    if (qt * 32 >= T) break;
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
    const int qa = min(qt * 32 + l31, T - 1);
    const float* Qp = a.q + (rowbase + qa) * a.ld + hd * HS + 4 * h;
    const float* dOp = a.dO + (rowbase + qa) * a.ldo + hd * HS + 4 * h;
    {
      constexpr int PF = (ND == 4 ? 2 : (NC < 4 ? NC : 4));
      f32x4 rq[PF], rd[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) { rq[i] = ld4g(Qp + 8 * i); rd[i] = ld4g(dOp + 8 * i); }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 qf = rq[c % PF], dof = rd[c % PF];
        if (c + PF < NC) { rq[c % PF] = ld4g(Qp + 8 * (c + PF)); rd[c % PF] = ld4g(dOp + 8 * (c + PF)); }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 kfc = *reinterpret_cast<const f32x4*>(krow + 8 * c);
        const f32x4 vfc = *reinterpret_cast<const f32x4*>(vrow + 8 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[j], kfc[j], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[j], vfc[j], dp, 0, 0, 0);
        }
      }
    }
    // per accumulator row: probabilities / dS, then rank-1-per-row updates of dV and dK; the dO and Q rows
    // they need are prefetched in groups of GR rows (2 at head size 128, where registers decide the occupancy)
    constexpr int GR = (ND == 4 ? 2 : 4), NG = 16 / GR;
    float a1[2][GR][ND], a2[2][GR][ND], lsev[2][GR], dlt[2][GR];
    auto rload = [&](int g, int buf) {
#pragma unroll
      for (int rr = 0; rr < GR; ++rr) {
        const int r = g * GR + rr;
        const int qc = min(qt * 32 + rowmap(r) + 4 * h, T - 1);
        lsev[buf][rr] = a.lse[statbase + qc];
        dlt[buf][rr] = a.delta[statbase + qc];
        ldrow<ND>(a.dO + (rowbase + qc) * a.ldo + hd * HS + ND * l31, dok, a1[buf][rr]);
        ldrow<ND>(a.q + (rowbase + qc) * a.ld + hd * HS + ND * l31, dok, a2[buf][rr]);
      }
    };
    rload(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) rload(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < GR; ++rr) {
        const int r = g * GR + rr;
        const int qrow = qt * 32 + rowmap(r) + 4 * h;
        const bool valid = qrow < T;
        const int qc = min(qrow, T - 1);
        const float ex = mmfn_exp((nokeys ? 0.f : st[r] * a.scale) - lsev[g & 1][rr]);
        const float p = (valid && kin) ? ex : 0.f;
        float msc = 1.f;
        if (drop) msc = mmfn_dropout_scale(key64, (statbase + qc) * (uint64_t)T + (uint64_t)key, a.drop_p, inv_keep);
        const float pd = p * msc;
        const float dsv = nokeys ? 0.f : p * (dp[r] * msc - dlt[g & 1][rr]) * a.scale;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd, a1[g & 1][rr][dt], dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dsv, a2[g & 1][rr][dt], dk[dt], 0, 0, 0);
        }
      }
    }
  }
  auto emit_to = [&](float* base) {
    return [=](int r, const float* t) {
      const int kr = kt0 * 32 + rowmap(r) + 4 * h;
      if (kr < T && dok) strow<ND>(base + (rowbase + kr) * a.ldg + hd * HS + ND * l31, t);
    };
  };
  __syncthreads();  // both waves are done with the staged K / V before the bytes are reused
  merge_store<ND>(dk, sm_k, w, lane, emit_to(a.dk));
  merge_store<ND>(dv, sm_v, w, lane, emit_to(a.dv));
}

template <int HS, int NKT>
int launch_attn(int which, const AttnArgs& a, hipStream_t s) {
  dim3 grid(ceil_div(a.T, 32), a.NH, a.B);
  if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<HS, NKT>), grid, dim3(128), 0, s, a);
  else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<HS, NKT>), grid, dim3(128), 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HS, NKT>), grid, dim3(128), 0, s, a);
  MMFN_LAUNCH_CHECK();
  return 0;
}

template <int HS>
int dispatch_nkt(int which, const AttnArgs& a, hipStream_t s) {
  const int nkt = ceil_div(a.T, 32);
  if (nkt <= 2) return launch_attn<HS, 2>(which, a, s);
  if (nkt <= 4) return launch_attn<HS, 4>(which, a, s);
  if (nkt <= 6) return launch_attn<HS, 6>(which, a, s);
  if (nkt <= 8) return launch_attn<HS, 8>(which, a, s);
  if (nkt <= 12) return launch_attn<HS, 12>(which, a, s);   // seq_len / n_views > 1: (n_views + 2) * seq_len * 64 tokens
  return MMFN_EINVAL;
}

// (rounds 2-3 had environment switches here that forced the one-tile-per-block kernels of this file, or the fp32-arithmetic
// kernels on bf16 tensors, for A/B runs; the measurements are in tools/experiments/NOTES.md)
constexpr bool tile_kernels_only() { return false; }
constexpr bool bf16_mfma_off() { return false; }

int dispatch(int which, int hs, const AttnArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.T <= 0 || a.T > 384 || a.NH <= 0) return MMFN_EINVAL;
  if ((a.ld & 3) || (a.ldo & 3) || ((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15) || ((uintptr_t)a.v & 15)) return MMFN_EINVAL;
  if (a.io_bf16 && !bf16_mfma_off()) {   // bf16 mode: bf16 MFMA kernels (attention16.hip)
    const int rc = mmfn_attn16_launch(which, hs, a, s);
    if (rc >= 0) return rc;
  }
  if ((!tile_kernels_only() || a.io_bf16) && !(a.ldg & 3)) {
    // T = 64 / 128 / 192 (the fusion transformers: 192 tokens): one workgroup per (sample, head, half), attention_wg.hip
    const int rc = mmfn_attn_wg_launch(which, hs, a, s);
    if (rc >= 0) return rc;
  }
  if (a.io_bf16) return MMFN_EINVAL;   // bf16 activations: only the workgroup-per-half kernels read them
  switch (hs) {
    case 16: return dispatch_nkt<16>(which, a, s);
    case 32: return dispatch_nkt<32>(which, a, s);
    case 64: return dispatch_nkt<64>(which, a, s);
    case 128: return dispatch_nkt<128>(which, a, s);
  }
  return MMFN_EINVAL;
}

}  // namespace

extern "C" int mmfn_attention_fwd_f32(const float* q, const float* k, const float* v, int ld, float* o, int ldo, float* lse,
                                      int B, int T, int NH, int HS, float scale, const int32_t* kv_len, float drop_p,
                                      const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.kv_len = kv_len; a.rng_state = rng_state;
  a.B = B; a.T = T; a.NH = NH; a.ld = ld; a.ldo = ldo; a.ldg = ld;
  a.scale = scale; a.drop_p = drop_p; a.rng_stream = rng_stream;
  if (drop_p > 0.f && !rng_state) return MMFN_EINVAL;
  return dispatch(0, HS, a, (hipStream_t)stream);
}

extern "C" int mmfn_attention_bwd_f32(const float* q, const float* k, const float* v, int ld, const float* o, const float* dO,
                                      int ldo, const float* lse, float* delta, float* dq, float* dk, float* dv, int ldg, int B,
                                      int T, int NH, int HS, float scale, const int32_t* kv_len, float drop_p,
                                      const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o = const_cast<float*>(o); a.lse = const_cast<float*>(lse); a.dO = dO; a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv; a.kv_len = kv_len; a.rng_state = rng_state;
  a.B = B; a.T = T; a.NH = NH; a.ld = ld; a.ldo = ldo; a.ldg = ldg;
  a.scale = scale; a.drop_p = drop_p; a.rng_stream = rng_stream;
  if (drop_p > 0.f && !rng_state) return MMFN_EINVAL;
  if ((ldg & 3) || !lse || !delta) return MMFN_EINVAL;
  int rc = dispatch(1, HS, a, (hipStream_t)stream);
  if (rc) return rc;
  return dispatch(2, HS, a, (hipStream_t)stream);
}

/* bf16 mode: q k v o dO dq dk dv are bf16 (strides in elements), lse / delta fp32; T = 64 / 128 / 192 (the workgroup-per-half
 * kernels: operands are widened to fp32 on their way into LDS, arithmetic as in the fp32 path) */
extern "C" int mmfn_attention_fwd_bf16(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                                       int B, int T, int NH, int HS, float scale, const int32_t* kv_len, float drop_p,
                                       const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  AttnArgs a = {};
  a.q = (const float*)q; a.k = (const float*)k; a.v = (const float*)v; a.o = (float*)o; a.lse = lse; a.kv_len = kv_len;
  a.rng_state = rng_state;
  a.B = B; a.T = T; a.NH = NH; a.ld = ld; a.ldo = ldo; a.ldg = ld;
  a.scale = scale; a.drop_p = drop_p; a.rng_stream = rng_stream; a.io_bf16 = 1;
  if (drop_p > 0.f && !rng_state) return MMFN_EINVAL;
  return dispatch(0, HS, a, (hipStream_t)stream);
}

extern "C" int mmfn_attention_bwd_bf16(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO,
                                       int ldo, const float* lse, float* delta, void* dq, void* dk, void* dv, int ldg, int B,
                                       int T, int NH, int HS, float scale, const int32_t* kv_len, float drop_p,
                                       const uint64_t* rng_state, uint32_t rng_stream, void* stream) {
  AttnArgs a = {};
  a.q = (const float*)q; a.k = (const float*)k; a.v = (const float*)v; a.o = (float*)const_cast<void*>(o);
  a.lse = const_cast<float*>(lse); a.dO = (const float*)dO; a.delta = delta;
  a.dq = (float*)dq; a.dk = (float*)dk; a.dv = (float*)dv; a.kv_len = kv_len; a.rng_state = rng_state;
  a.B = B; a.T = T; a.NH = NH; a.ld = ld; a.ldo = ldo; a.ldg = ldg;
  a.scale = scale; a.drop_p = drop_p; a.rng_stream = rng_stream; a.io_bf16 = 1;
  if (drop_p > 0.f && !rng_state) return MMFN_EINVAL;
  if ((ldg & 3) || !lse || !delta) return MMFN_EINVAL;
  if (a.B > 0 && a.T > 0 && a.NH > 0 && !((uintptr_t)a.q & 15) && !((uintptr_t)a.k & 15) && !((uintptr_t)a.v & 15)) {
    // both passes in one launch (attention16.hip attn16_bwd_kernel): the key-owned pass forms delta itself
    const int both = mmfn_attn16_launch(3, HS, a, (hipStream_t)stream);
    if (both >= 0) return both;
  }
  int rc = dispatch(1, HS, a, (hipStream_t)stream);
  if (rc) return rc;
  return dispatch(2, HS, a, (hipStream_t)stream);
}
