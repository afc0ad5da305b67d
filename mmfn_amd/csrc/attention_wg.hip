// Fused multi-head attention, workgroup-per-(sample, head, half) form (model_vec.py:92-109 SelfAttention; fp32 on
// v_mfma_f32_16x16x4_f32).
//
// Why a second form.  attention.hip gives every 32-query tile its own 2-wave block: 768 blocks x 2 waves = 1.5 waves per
// SIMD at the fusion-transformer shape (B = 32, 4 heads, T = 192), so a quarter of the matrix pipes idle by construction
// (the makespan is two waves' work on the SIMDs that got two), and every block re-streams all of K / V from L2 (6x).
// An fp32 MFMA occupies its SIMD's matrix pipe for 32 (16x16x4) or 64 (32x32x2) cycles, so what matters here is not
// operand bandwidth but that all 1024 pipes hold the SAME amount of work and never wait:
//   * grid = (2 halves, heads, batch) = 256 workgroups of 8 waves = one workgroup per CU, two waves per SIMD;
//   * 16x16 tiles give the granularity that divides: a workgroup owns T/2 queries (forward, dQ pass) or T/2 keys (dK/dV
//     pass) against all T of the other side; wave (g, s) takes the 16*NT x 16*NT sub-block of group g (2 per half) and
//     contraction slice s (4 per T), NT = T/64 - exactly 1/8 of the workgroup's MFMAs each, in every phase;
//   * operands go L2 -> registers directly, 16 B per lane, prefetched one 16-column chunk ahead: each wave reads its
//     rows once, a K / V row is read by 4 waves per (sample, head) instead of 12;
//   * the S^T / dP^T accumulator tile IS the A operand of the second product (accumulator row 4*(lane>>4)+r <-> k slot
//     lane>>4, step r), so probabilities never leave registers;
//   * the four slices meet through LDS three times: row max, row sum (delta in the backward), and the partial output
//     tiles - each wave finishes the quarter of the head dimension it owns, so the merge is balanced as well and the
//     result leaves as 64..128-byte contiguous runs per row.
#include "attention_args.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int HS, int NT>
struct Shape {
  static constexpr int NC = HS / 16;                  // 16-column chunks of the head dimension (first product)
  static constexpr int NDT = HS / 16;                 // 16-wide output tiles of the second product
  static constexpr int NOWN = NDT >= 4 ? 4 : NDT;     // waves (of the 4 slices) that own output tiles in the merge
  static constexpr int W = NDT / NOWN;                // tiles per owner = consecutive floats per lane
  static constexpr int G = 16 * NT;                   // rows per wave group
  static constexpr int NFOREIGN = NDT - W;            // tiles a wave hands to other owners (owners), NDT for non-owners
  // merge buffer: [group 2][slice 4][tile slot NDT][row tile NT][reg 4][lane 64] floats; owners never write their own
  static constexpr int SLOTS = NDT >= 4 ? NDT - W : NDT;
  static constexpr int MERGE_FLOATS = 2 * 4 * SLOTS * NT * 4 * 64;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// column of output tile j held by lane column n: owner-contiguous (tile group j / W covers 16*W consecutive columns)
template <int W>
__device__ __forceinline__ int dcol(int n, int j) { return (j / W) * 16 * W + n * W + (j % W); }

// acc[p][x][y] += sum_d A_p[16x + l15][d] * B_p[16y + l15][d] over the HS columns of this head, NP products at once.
// Lane (l15, l4) loads the float4 at columns 16c + 4*l4 of its row: element e feeds MFMA step e, whose four k slots are
// then the columns {16c + 4*slot + e}: a permutation of the chunk's columns, the same one on both operands.
template <int HS, int NT, int NP>
__device__ __forceinline__ void product_phase(const float* const* Abase, const size_t* lda, const float* const* Bbase,
                                              const size_t* ldb, int l15, int l4, f32x4 (*acc)[NT][NT]) {
  constexpr int NC = HS / 16;
  f32x4 fa[2][NP][NT], fb[2][NP][NT];
  auto load = [&](int c, int buf) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        fa[buf][p][t] = ld4(Abase[p] + (size_t)(16 * t + l15) * lda[p] + 16 * c + 4 * l4);
        fb[buf][p][t] = ld4(Bbase[p] + (size_t)(16 * t + l15) * ldb[p] + 16 * c + 4 * l4);
      }
  };
  load(0, 0);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + 1 < NC) load(c + 1, (c + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
          for (int y = 0; y < NT; ++y) acc[p][x][y] = mfma16(fa[c & 1][p][x][e], fb[c & 1][p][y][e], acc[p][x][y]);
  }
}

// out[y][j] += sum over this wave's contraction rows r of P[r][row 16y + ..] * R[r][dcol(.., j)]:  P is the accumulator
// of product_phase ([x = contraction tile][y = output-row tile]), R rows start at `rows` with stride ld.
template <int HS, int NT>
__device__ __forceinline__ void second_phase(const f32x4 (*P)[NT], const float* rows, size_t ld, int l15, int l4,
                                             f32x4 (*out)[HS / 16]) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, W = S::W, NG = NDT / W;
  float rv[2][NDT];
  auto load = [&](int step, int buf) {
    const int x = step >> 2, e = step & 3;
    const float* p = rows + (size_t)(16 * x + 4 * l4 + e) * ld + l15 * W;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (W == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p + 32 * g);
        rv[buf][2 * g] = t[0]; rv[buf][2 * g + 1] = t[1];
      } else {
        rv[buf][g] = p[16 * g];
      }
    }
  };
  load(0, 0);
#pragma unroll
  for (int step = 0; step < 4 * NT; ++step) {
    if (step + 1 < 4 * NT) load(step + 1, (step + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int x = step >> 2, e = step & 3;
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int j = 0; j < NDT; ++j) out[y][j] = mfma16(P[x][y][e], rv[step & 1][j], out[y][j]);
  }
}

// Partial output tiles of the four slices of a group -> their sum, written to dst (row stride ldd, rows of this group).
// Slice s owns tiles [s*W, s*W+W) (the first NOWN slices when the head has fewer than four tiles); everybody parks the
// tiles it does not own in LDS, one barrier, owners add the three foreign copies in slice order (deterministic).
template <int HS, int NT>
__device__ __forceinline__ void merge_store(const f32x4 (*acc)[HS / 16], float* sm, int grp, int slice, int lane, int l15,
                                            int l4, float* dst, size_t ldd) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, W = S::W, NOWN = S::NOWN, SLOTS = S::SLOTS;
  auto slab = [&](int src, int slot, int y, int r) {
    return sm + ((((size_t)(grp * 4 + src) * SLOTS + slot) * NT + y) * 4 + r) * 64 + lane;
  };
  const bool owner = slice < NOWN;
  // slot of tile j in slice src's parking area: its own tiles are skipped when it is an owner of a full (NDT >= 4) head
  auto slot_of = [&](int src, int j) { return (NDT >= 4 && j >= (src + 1) * W) ? j - W : j; };
#pragma unroll
  for (int j = 0; j < NDT; ++j) {
    const bool mine = owner && (j / W) == slice;
    if (!mine) {
      const int sl = slot_of(slice, j);
#pragma unroll
      for (int y = 0; y < NT; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) *slab(slice, sl, y, r) = acc[y][j][r];
    }
  }
  __syncthreads();
  if (owner) {
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t[W];
#pragma unroll
        for (int jj = 0; jj < W; ++jj) {
          const int j = slice * W + jj;   // NDT < 4: W == 1 and slice < NDT, so j = slice
          float v = 0.f;
          bool first = true;
#pragma unroll
          for (int src = 0; src < 4; ++src) {   // slice order, own copy taken from registers at its place in the order
            float part;
            if (src == slice) {
              // select acc[y][j][r] with a compile-time-indexable form: j depends on slice (wave-uniform) -> unrolled compare
              part = 0.f;
#pragma unroll
              for (int jc = 0; jc < NDT; ++jc)
                if (jc == j) part = acc[y][jc][r];
            } else {
              part = *slab(src, slot_of(src, j), y, r);
            }
            v = first ? part : v + part;
            first = false;
          }
          t[jj] = v;
        }
        float* p = dst + (size_t)(16 * y + 4 * l4 + r) * ldd + slice * 16 * W + l15 * W;
        if (W == 2) *reinterpret_cast<f32x2*>(p) = f32x2{t[0], t[1]};
        else p[0] = t[0];
      }
  }
}

__device__ __forceinline__ float quad_max(float v) {   // over the four lanes l4 = 0..3 that share l15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------------------- forward
template <int HS, int NT>
__global__ __launch_bounds__(512) void attn_wg_fwd_kernel(const AttnArgs a) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, G = S::G, T = 64 * NT;
  __shared__ float sm_merge[S::MERGE_FLOATS];
  __shared__ float sm_stat[2][2][4][G];   // [max | sum][query group][key slice][query]
  const int half = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int qg = w >> 2, ks = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int q0 = half * 2 * G + qg * G;   // first query of this wave's group
  const int k0 = ks * G;                  // first key of this wave's slice
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const size_t ld = a.ld;
  f32x4 s[1][NT][NT];   // S^T: [key tile][query tile], acc row = key 4*l4 + r, lane column = query l15
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) s[0][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* A[1] = {a.k + (rowbase + k0) * ld + hd * HS};
    const float* Bq[1] = {a.q + (rowbase + q0) * ld + hd * HS};
    const size_t l1[1] = {ld};
    product_phase<HS, NT, 1>(A, l1, Bq, l1, l15, l4, s);
  }
  // ---- softmax over keys: local (12 values per query) -> quad -> the four key slices through LDS
  float mx[NT], sum[NT];
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    float m = -INFINITY;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * x + 4 * l4 + r;
        // kv_len == 0: the reference's masked_fill(-1e9) + softmax is uniform attention over all keys (model_vec.py:315-317)
        const float v = nokeys ? 0.f : (key < kvlen ? s[0][x][y][r] * a.scale : -INFINITY);
        s[0][x][y][r] = v;
        m = fmaxf(m, v);
      }
    mx[y] = quad_max(m);
    if (l4 == 0) sm_stat[0][qg][ks][16 * y + l15] = mx[y];
  }
  __syncthreads();
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    float m = sm_stat[0][qg][0][16 * y + l15];
#pragma unroll
    for (int o = 1; o < 4; ++o) m = fmaxf(m, sm_stat[0][qg][o][16 * y + l15]);
    mx[y] = m;   // finite: at least one key of the row is unmasked (kv_len >= 1 or the uniform case)
    float t = 0.f;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = s[0][x][y][r] > -INFINITY ? expf(s[0][x][y][r] - m) : 0.f;
        s[0][x][y][r] = e;
        t += e;
      }
    sum[y] = quad_sum(t);
    if (l4 == 0) sm_stat[1][qg][ks][16 * y + l15] = sum[y];
  }
  __syncthreads();
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int qi = 16 * y + l15;
    const float l = (sm_stat[1][qg][0][qi] + sm_stat[1][qg][1][qi]) + (sm_stat[1][qg][2][qi] + sm_stat[1][qg][3][qi]);
    const int q = q0 + qi;
    if (ks == 0 && l4 == 0 && a.lse) a.lse[((size_t)b * a.NH + hd) * T + q] = mx[y] + logf(l);
    const float inv = 1.0f / l;
    const uint64_t pbase = (((uint64_t)b * a.NH + hd) * T + q) * (uint64_t)T;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = s[0][x][y][r] * inv;
        if (drop) p *= mmfn_dropout_scale(key64, pbase + (uint64_t)(k0 + 16 * x + 4 * l4 + r), a.drop_p, inv_keep);
        s[0][x][y][r] = p;
      }
  }
  // ---- O = P . V over this wave's key slice, then the four slices are summed
  f32x4 o[NT][NDT];
#pragma unroll
  for (int y = 0; y < NT; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) o[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT>(s[0], a.v + (rowbase + k0) * ld + hd * HS, ld, l15, l4, o);
  merge_store<HS, NT>(o, sm_merge, qg, ks, lane, l15, l4, a.o + (rowbase + q0) * a.ldo + hd * HS, a.ldo);
}

// ---------------------------------------------------------------------------------------- backward, query-owned: dQ, delta
template <int HS, int NT>
__global__ __launch_bounds__(512) void attn_wg_dq_kernel(const AttnArgs a) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, G = S::G, T = 64 * NT;
  __shared__ float sm_merge[S::MERGE_FLOATS];
  __shared__ float sm_stat[2][2][4][G];   // [sum P dP | sum P][query group][key slice][query]
  const int half = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int qg = w >> 2, ks = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int q0 = half * 2 * G + qg * G;
  const int k0 = ks * G;
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;   // constant scores: P = 1/T, dS = 0
  const size_t ld = a.ld;
  f32x4 acc[2][NT][NT];   // [0] S^T = K Q^T, [1] dP^T = V dO^T
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) acc[p][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* A[2] = {a.k + (rowbase + k0) * ld + hd * HS, a.v + (rowbase + k0) * ld + hd * HS};
    const float* Bq[2] = {a.q + (rowbase + q0) * ld + hd * HS, a.dO + (rowbase + q0) * a.ldo + hd * HS};
    const size_t la[2] = {ld, ld}, lb[2] = {ld, (size_t)a.ldo};
    product_phase<HS, NT, 2>(A, la, Bq, lb, l15, l4, acc);
  }
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
  const size_t statbase = ((size_t)b * a.NH + hd) * T;
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int q = q0 + 16 * y + l15;
    const float lse = a.lse[statbase + q];
    const uint64_t pbase = (statbase + q) * (uint64_t)T;
    float dl = 0.f, ps = 0.f;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * x + 4 * l4 + r;
        const float p = nokeys ? expf(-lse) : (key < kvlen ? expf(acc[0][x][y][r] * a.scale - lse) : 0.f);
        float dpv = acc[1][x][y][r];
        if (drop) dpv *= mmfn_dropout_scale(key64, pbase + (uint64_t)key, a.drop_p, inv_keep);
        acc[0][x][y][r] = p;
        acc[1][x][y][r] = dpv;
        dl += p * dpv;
        ps += p;
      }
    dl = quad_sum(dl);
    ps = quad_sum(ps);
    if (l4 == 0) { sm_stat[0][qg][ks][16 * y + l15] = dl; sm_stat[1][qg][ks][16 * y + l15] = ps; }
  }
  __syncthreads();
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int qi = 16 * y + l15;
    // P is recomputed from the rounded log-sum-exp, so sum_j P_j = 1 + O(1e-6); dividing by it keeps sum_j dS_j = 0 to
    // rounding (attention.hip).  All slices add in the same order, so they agree on delta bit for bit.
    const float num = (sm_stat[0][qg][0][qi] + sm_stat[0][qg][1][qi]) + (sm_stat[0][qg][2][qi] + sm_stat[0][qg][3][qi]);
    const float den = (sm_stat[1][qg][0][qi] + sm_stat[1][qg][1][qi]) + (sm_stat[1][qg][2][qi] + sm_stat[1][qg][3][qi]);
    const float delta = num / den;
    if (ks == 0 && l4 == 0) a.delta[statbase + q0 + qi] = delta;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[0][x][y][r] = nokeys ? 0.f : acc[0][x][y][r] * (acc[1][x][y][r] - delta) * a.scale;
  }
  f32x4 dq[NT][NDT];
#pragma unroll
  for (int y = 0; y < NT; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) dq[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT>(acc[0], a.k + (rowbase + k0) * ld + hd * HS, ld, l15, l4, dq);
  merge_store<HS, NT>(dq, sm_merge, qg, ks, lane, l15, l4, a.dq + (rowbase + q0) * a.ldg + hd * HS, a.ldg);
}

// ------------------------------------------------------------------------------------------ backward, key-owned: dK, dV
template <int HS, int NT>
__global__ __launch_bounds__(512) void attn_wg_dkv_kernel(const AttnArgs a) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, G = S::G, T = 64 * NT;
  __shared__ float sm_merge[S::MERGE_FLOATS];
  const int half = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int kg = w >> 2, qs = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int k0 = half * 2 * G + kg * G;   // first key of this wave's group
  const int q0 = qs * G;                  // first query of this wave's slice
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const size_t ld = a.ld;
  f32x4 acc[2][NT][NT];   // [0] S = Q K^T, [1] dP = dO V^T: [query tile][key tile], acc row = query 4*l4 + r, column = key
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) acc[p][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* A[2] = {a.q + (rowbase + q0) * ld + hd * HS, a.dO + (rowbase + q0) * a.ldo + hd * HS};
    const float* Bk[2] = {a.k + (rowbase + k0) * ld + hd * HS, a.v + (rowbase + k0) * ld + hd * HS};
    const size_t la[2] = {ld, (size_t)a.ldo}, lb[2] = {ld, ld};
    product_phase<HS, NT, 2>(A, la, Bk, lb, l15, l4, acc);
  }
  const bool drop = a.drop_p > 0.f;
  uint64_t key64 = 0;
  float inv_keep = 1.f;
  if (drop) { key64 = mmfn_rng_key(a.rng_state, a.rng_stream); inv_keep = 1.0f / (1.0f - a.drop_p); }
  const size_t statbase = ((size_t)b * a.NH + hd) * T;
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 16 * x + 4 * l4 + r;
      const float lse = a.lse[statbase + q], dlt = a.delta[statbase + q];
#pragma unroll
      for (int y = 0; y < NT; ++y) {
        const int key = k0 + 16 * y + l15;
        const bool kin = nokeys || key < kvlen;
        const float p = kin ? expf((nokeys ? 0.f : acc[0][x][y][r] * a.scale) - lse) : 0.f;
        float msc = 1.f;
        if (drop) msc = mmfn_dropout_scale(key64, (statbase + q) * (uint64_t)T + (uint64_t)key, a.drop_p, inv_keep);
        acc[0][x][y][r] = p * msc;                                                           // dV = (P o mask)^T dO
        acc[1][x][y][r] = nokeys ? 0.f : p * (acc[1][x][y][r] * msc - dlt) * a.scale;         // dK = dS^T Q
      }
    }
  f32x4 g[NT][NDT];
#pragma unroll
  for (int y = 0; y < NT; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) g[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT>(acc[0], a.dO + (rowbase + q0) * a.ldo + hd * HS, a.ldo, l15, l4, g);
  merge_store<HS, NT>(g, sm_merge, kg, qs, lane, l15, l4, a.dv + (rowbase + k0) * a.ldg + hd * HS, a.ldg);
#pragma unroll
  for (int y = 0; y < NT; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) g[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT>(acc[1], a.q + (rowbase + q0) * ld + hd * HS, ld, l15, l4, g);
  __syncthreads();   // every owner has read the dV copies before the parking area is reused
  merge_store<HS, NT>(g, sm_merge, kg, qs, lane, l15, l4, a.dk + (rowbase + k0) * a.ldg + hd * HS, a.ldg);
}

template <int HS, int NT>
int launch(int which, const AttnArgs& a, hipStream_t s) {
  dim3 grid(2, a.NH, a.B);
  if (which == 0) hipLaunchKernelGGL((attn_wg_fwd_kernel<HS, NT>), grid, dim3(512), 0, s, a);
  else if (which == 1) hipLaunchKernelGGL((attn_wg_dq_kernel<HS, NT>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((attn_wg_dkv_kernel<HS, NT>), grid, dim3(512), 0, s, a);
  MMFN_LAUNCH_CHECK();
  return 0;
}

template <int HS>
int by_tokens(int which, const AttnArgs& a, hipStream_t s) {
  switch (a.T) {
    case 64: return launch<HS, 1>(which, a, s);
    case 128: return launch<HS, 2>(which, a, s);
    case 192: return launch<HS, 3>(which, a, s);
  }
  return -1;
}

}  // namespace

int mmfn_attn_wg_launch(int which, int hs, const AttnArgs& a, hipStream_t s) {
  switch (hs) {
    case 16: return by_tokens<16>(which, a, s);
    case 32: return by_tokens<32>(which, a, s);
    case 64: return by_tokens<64>(which, a, s);
    case 128: return by_tokens<128>(which, a, s);
  }
  return -1;
}
