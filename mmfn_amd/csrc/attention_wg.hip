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
//   * operands are staged through LDS by the whole workgroup (16-byte coalesced row loads, one copy per workgroup):
//     with L2 -> register operand loads every wave re-read the rows its seven siblings also need (580 KB per workgroup
//     at head size 128 against 240 KB of distinct data) and the CU's vector-memory path, not the matrix pipe, set the
//     pace (51 cycles per MFMA measured where the instruction needs 32); the MFMA phases now read LDS only;
//   * the S^T / dP^T accumulator tile IS the A operand of the second product (accumulator row 4*(lane>>4)+r <-> k slot
//     lane>>4, step r), so probabilities never leave registers;
//   * the four slices meet through LDS three times: row max, row sum (delta in the backward), and the partial output
//     tiles - each wave finishes the quarter of the head dimension it owns, so the merge is balanced as well and the
//     result leaves as 64..128-byte contiguous runs per row;
//   * the key-owned pass keeps BOTH accumulator sets (P o mask for dV, dS for dK: 72 registers at T = 192) across its two
//     second products; at head size 128 (96 more for the output tiles, 48 for the operand in flight) the recomputed
//     probabilities are pinned where they are formed (an empty asm per element) - left to itself the compiler sinks those
//     expressions to their MFMA uses and keeps every element's inputs alive instead: 312 bytes of scratch per lane and
//     62.6 us up to round 5, none and 54 us now.  build.check_scratch() fails the build on scratch in any kernel.
#include <stdlib.h>

#include <type_traits>

#include "attention_args.h"
#include "gpt_block.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NTHR = 512;   // 8 waves: two per SIMD

// NT: 16-row tiles per contraction slice (T = 4 slices x 16 NT rows); NY: tiles per wave group on the OWNED side (a workgroup owns
// 2 groups = 32 NY rows: T/2 with NY = NT - the (sample, head, half) form - or T/4 with NY = NT/2: four workgroups per (sample, head),
// for T = 256 where the halves' accumulator sets and operands no longer fit and B = 16 needs the extra workgroups to fill the chip)
template <int HS, int NT, int NY = NT>
struct Shape {
  static constexpr int NC = HS / 16;                  // 16-column chunks of the head dimension (first product)
  static constexpr int NDT = HS / 16;                 // 16-wide output tiles of the second product
  static constexpr int NOWN = NDT >= 4 ? 4 : NDT;     // waves (of the 4 slices) that own output tiles in the merge
  static constexpr int W = NDT / NOWN;                // tiles per owner = consecutive floats per lane
  static constexpr int G = 16 * NY;                   // owned rows per wave group
  static constexpr int GX = 16 * NT;                  // rows per contraction slice
  static constexpr int PARTS = 2 * NT / NY;           // workgroups per (sample, head)
  static constexpr int NFOREIGN = NDT - W;            // tiles a wave hands to other owners (owners), NDT for non-owners
  // merge buffer: [group 2][slice 4][tile slot NDT][row tile NT][reg 4][lane 64] floats; owners never write their own
  static constexpr int SLOTS = NDT >= 4 ? NDT - W : NDT;
  static constexpr int MERGE_FLOATS_PER_GROUP = 4 * SLOTS * NY * 4 * 64;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// column of output tile j held by lane column n: owner-contiguous (tile group j / W covers 16*W consecutive columns)
template <int W>
__device__ __forceinline__ int dcol(int n, int j) { return (j / W) * 16 * W + n * W + (j % W); }

// ---- operand staging.  An operand matrix in LDS is [rows][HS] with row pitch HS + 4 floats: the 16-byte fragment
// reads of 16 consecutive rows then fall into 16 distinct bank quads.
template <int HS>
struct Pitch { static constexpr int P = HS + 4; };

// Cooperative copy of NROWS rows x NCOLS columns (columns col0 .. col0 + NCOLS of an HS-wide operand, global row stride
// ld) into LDS: issue() puts every thread's 16-byte pieces in flight into registers, commit() writes them to LDS - the
// caller places work (MFMAs, softmax, a barrier) in between.
template <int HS, int NROWS, int NCOLS, int NTHR>
struct Stager {
  static constexpr int Q = NCOLS / 4;
  static constexpr int UNITS = NROWS * Q;
  static constexpr int PER = (UNITS + NTHR - 1) / NTHR;
  f32x4 r[PER];
  template <typename TIO>
  __device__ __forceinline__ void issue(const TIO* src, size_t ld, int tid, int col0 = 0) {   // TIO = float or bf16_t (bf16 mode)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NTHR;
      if (UNITS % NTHR == 0 || u < UNITS) r[i] = ldx4(src + (size_t)(u / Q) * ld + col0 + 4 * (u % Q));
    }
  }
  __device__ __forceinline__ void commit(float* dst, int tid, int col0 = 0) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NTHR;
      if (UNITS % NTHR == 0 || u < UNITS)
        *reinterpret_cast<f32x4*>(dst + (u / Q) * Pitch<HS>::P + col0 + 4 * (u % Q)) = r[i];
    }
  }
  // the same columns as a matrix of their own: [NROWS][NCOLS] with row pitch NCOLS + 4 (a column-group window, product_chain_w)
  __device__ __forceinline__ void commit_window(float* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NTHR;
      if (UNITS % NTHR == 0 || u < UNITS) *reinterpret_cast<f32x4*>(dst + (u / Q) * (NCOLS + 4) + 4 * (u % Q)) = r[i];
    }
  }
};

// Column groups of the first products: the operands arrive in NS groups of HS / NS columns; group g+1 is in flight
// (global -> registers) while the MFMAs of group g run, so only the first group's latency is exposed.  (Cold operands
// stream at HBM speed: 150 KB per workgroup x 256 workgroups at head size 128 is ~6 us that used to precede any MFMA.)
template <int HS>
struct Groups {
  static constexpr int NS = HS >= 128 ? 4 : (HS >= 64 ? 2 : 1);
  static constexpr int COLS = HS / NS;
};

// acc[x][y] += sum_d A[16x + l15][d] * B[16y + l15][d] over columns [col0, col0 + NCOLS) of this head; A, B point at the
// first row of this wave's tiles inside staged LDS matrices.  Lane (l15, l4) reads the float4 at columns 16c + 4*l4 of its
// row: element e feeds MFMA step e, whose four k slots are then the columns {16c + 4*slot + e}: a permutation of the
// chunk's columns, the same one on both operands.
template <int HS, int NT, int NCOLS, int NY = NT>
__device__ __forceinline__ void product_phase(const float* A, const float* B, int col0, int l15, int l4, f32x4 (*acc)[NY]) {
  constexpr int NC = NCOLS / 16, P = Pitch<HS>::P;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x4 fa[NT], fb[NY];
#pragma unroll
    for (int t = 0; t < NT; ++t) fa[t] = *reinterpret_cast<const f32x4*>(A + (16 * t + l15) * P + col0 + 16 * c + 4 * l4);
#pragma unroll
    for (int t = 0; t < NY; ++t) fb[t] = *reinterpret_cast<const f32x4*>(B + (16 * t + l15) * P + col0 + 16 * c + 4 * l4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[x][y] = mfma16(fa[x][e], fb[y][e], acc[x][y]);
  }
}

// A chain of N first products, each over a (full T-row, half T/2-row) operand pair that arrives in column groups:
//   src(i) -> (pointer / stride of the full operand, of the half operand) of product i;  prod(i, col0) runs its MFMAs;
//   tail() is called right before the last group's MFMAs (the caller issues the next phase's loads there).
// Every group is one barrier.  With NS >= 2 the group being written (columns of group g+1) was last read NS groups ago, one
// or more barriers back; with NS == 1 the only group is overwritten, so a barrier separates its last reader from the write.
template <int HS, int NT, int NPROD, typename TIO, int NY = NT, typename Src, typename Prod, typename Tail>
__device__ __forceinline__ void product_chain(float* sFull, float* sHalf, int tid, Src&& src, Prod&& prod, Tail&& tail) {
  constexpr int T = 64 * NT, NS = Groups<HS>::NS, COLS = Groups<HS>::COLS, N = NPROD * NS;
  Stager<HS, T, COLS, NTHR> stF;
  Stager<HS, 32 * NY, COLS, NTHR> stH;   // the workgroup's own rows: two groups of 16 NY
  const TIO* pf; const TIO* ph; size_t lf, lh;
  src(0, pf, lf, ph, lh);
  stF.issue(pf, lf, tid, 0);
  stH.issue(ph, lh, tid, 0);
  stF.commit(sFull, tid, 0);
  stH.commit(sHalf, tid, 0);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int col0 = (i % NS) * COLS;
    if (i + 1 < N) {
      src((i + 1) / NS, pf, lf, ph, lh);
      stF.issue(pf, lf, tid, ((i + 1) % NS) * COLS);
      stH.issue(ph, lh, tid, ((i + 1) % NS) * COLS);
    } else {
      tail();
    }
    prod(i / NS, col0);
    if (i + 1 < N) {
      if (NS == 1) __syncthreads();
      stF.commit(sFull, tid, ((i + 1) % NS) * COLS);
      stH.commit(sHalf, tid, ((i + 1) % NS) * COLS);
      __syncthreads();
    }
  }
}

// ---- windowed first products (T = 256 at head size 128).  The all-T operand of a first product is only ever read one column
// group at a time, so it need not be resident as a [T][HS] matrix (135 KB: with the workgroup's own rows 169 KB > LDS): its
// column groups pass through a two-slot ring of [T][COLS + 4] windows (2 x 37 KB), the own-row operand keeps its full-width
// layout.  Slot (g+1) % 2 is rewritten while group g is multiplied; it was last read by group g-1, one barrier back.
template <int HS, int NT, int NCOLS, int NY>
__device__ __forceinline__ void product_phase_w(const float* Aw, const float* B, int col0, int l15, int l4, f32x4 (*acc)[NY]) {
  constexpr int NC = NCOLS / 16, P = Pitch<HS>::P, PW = NCOLS + 4;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x4 fa[NT], fb[NY];
#pragma unroll
    for (int t = 0; t < NT; ++t) fa[t] = *reinterpret_cast<const f32x4*>(Aw + (16 * t + l15) * PW + 16 * c + 4 * l4);
#pragma unroll
    for (int t = 0; t < NY; ++t) fb[t] = *reinterpret_cast<const f32x4*>(B + (16 * t + l15) * P + col0 + 16 * c + 4 * l4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[x][y] = mfma16(fa[x][e], fb[y][e], acc[x][y]);
  }
}

template <int HS, int NT>
struct Window {
  static constexpr int COLS = Groups<HS>::COLS, PW = COLS + 4, SLOT = 64 * NT * PW, RING = 2 * SLOT;
};

// prod(i, window, col0): window = the ring slot holding columns [col0, col0 + COLS) of product i's all-T operand
template <int HS, int NT, int NPROD, typename TIO, int NY, typename Src, typename Prod, typename Tail>
__device__ __forceinline__ void product_chain_w(float* sRing, float* sHalf, int tid, Src&& src, Prod&& prod, Tail&& tail) {
  constexpr int T = 64 * NT, NS = Groups<HS>::NS, COLS = Groups<HS>::COLS, N = NPROD * NS, SLOT = Window<HS, NT>::SLOT;
  static_assert(NS >= 2, "a window ring needs at least two column groups");
  Stager<HS, T, COLS, NTHR> stF;
  Stager<HS, 32 * NY, COLS, NTHR> stH;
  const TIO* pf; const TIO* ph; size_t lf, lh;
  src(0, pf, lf, ph, lh);
  stF.issue(pf, lf, tid, 0);
  stH.issue(ph, lh, tid, 0);
  stF.commit_window(sRing, tid);
  stH.commit(sHalf, tid, 0);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int col0 = (i % NS) * COLS;
    if (i + 1 < N) {
      src((i + 1) / NS, pf, lf, ph, lh);
      stF.issue(pf, lf, tid, ((i + 1) % NS) * COLS);
      stH.issue(ph, lh, tid, ((i + 1) % NS) * COLS);
    } else {
      tail();
    }
    prod(i / NS, sRing + (i & 1) * SLOT, col0);
    if (i + 1 < N) {
      stF.commit_window(sRing + ((i + 1) & 1) * SLOT, tid);
      stH.commit(sHalf, tid, ((i + 1) % NS) * COLS);   // (a product's own-row columns are new; the next product's overwrite group 0 last read NS - 1 barriers ago)
      __syncthreads();
    }
  }
}

// out[y][j] += sum over this wave's contraction rows r of P[r][row 16y + ..] * R[r][dcol(.., j)]:  P is the accumulator
// of product_phase ([x = contraction tile][y = output-row tile]), R the staged LDS matrix at the wave's first row.
template <int HS, int NT, int NY = NT>
__device__ __forceinline__ void second_phase(const f32x4 (*Pm)[NY], const float* rows, int l15, int l4, f32x4 (*out)[HS / 16]) {
  using S = Shape<HS, NT, NY>;
  constexpr int NDT = S::NDT, W = S::W, NGR = NDT / W, P = Pitch<HS>::P;
#pragma unroll
  for (int step = 0; step < 4 * NT; ++step) {
    const int x = step >> 2, e = step & 3;
    const float* p = rows + (16 * x + 4 * l4 + e) * P + l15 * W;
    float rv[NDT];
#pragma unroll
    for (int g = 0; g < NGR; ++g) {
      if (W == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p + 32 * g);
        rv[2 * g] = t[0]; rv[2 * g + 1] = t[1];
      } else {
        rv[g] = p[16 * g];
      }
    }
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int j = 0; j < NDT; ++j) out[y][j] = mfma16(Pm[x][y][e], rv[j], out[y][j]);
  }
}

// Partial output tiles of the four slices of a group -> their sum, written to dst (row stride ldd, rows of this group).
// Slice s owns tiles [s*W, s*W+W) (the first NOWN slices when the head has fewer than four tiles); everybody parks the
// tiles it does not own in LDS, one barrier, owners add the three foreign copies in slice order (deterministic).
template <int HS, int NT, typename TIO>
__device__ __forceinline__ void merge_store(const f32x4 (*acc)[HS / 16], float* sm, int grp, int slice, int lane, int l15,
                                            int l4, TIO* dst, size_t ldd) {
  using S = Shape<HS, NT>;
  constexpr int NDT = S::NDT, W = S::W, NOWN = S::NOWN, SLOTS = S::SLOTS;
  // one 16-byte unit per (tile, row tile, lane): the four accumulator rows of a lane sit together, lane-linear across the wave
  auto slab = [&](int src, int slot, int y) {
    return reinterpret_cast<f32x4*>(sm) + (((size_t)(grp * 4 + src) * SLOTS + slot) * NT + y) * 64 + lane;
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
      for (int y = 0; y < NT; ++y) *slab(slice, sl, y) = acc[y][j];
    }
  }
  __syncthreads();
  // the owner's part, with the slice as a compile-time constant (slice is wave-uniform: one scalar branch, static registers)
  auto own = [&](auto SL) {
    constexpr int sl = decltype(SL)::value;
#pragma unroll
    for (int y = 0; y < NT; ++y) {
      f32x4 t[W];
#pragma unroll
      for (int jj = 0; jj < W; ++jj) {
        const int j = sl * W + jj;   // NDT < 4: W == 1 and slice < NDT, so j = slice
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src = 0; src < 4; ++src) {   // slice order, own copy taken from registers at its place in the order
          const f32x4 part = src == sl ? acc[y][j < NDT ? j : 0] : *slab(src, slot_of(src, j), y);
          v = src == 0 ? part : v + part;
        }
        t[jj] = v;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        TIO* p = dst + (size_t)(16 * y + 4 * l4 + r) * ldd + sl * 16 * W + l15 * W;
        if (W == 2) VecIO<2, TIO>::st(p, f32x2{t[0][r], t[W - 1][r]});
        else stx1(p, t[0][r]);
      }
    }
  };
  if (owner) {
    switch (slice) {
      case 0: own(std::integral_constant<int, 0>{}); break;
      case 1: if constexpr (NOWN > 1) own(std::integral_constant<int, 1>{}); break;
      case 2: if constexpr (NOWN > 2) own(std::integral_constant<int, 2>{}); break;
      default: if constexpr (NOWN > 3) own(std::integral_constant<int, 3>{}); break;
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

// LDS plan of a workgroup (8 waves, 512 threads), in floats.  The operand area holds, at different times, two staged
// operand matrices for the products (T + T/2 rows), one T-row matrix for the second product, and the merge parking area.
template <int HS, int NT, int NY = NT>
struct Lds {
  using S = Shape<HS, NT, NY>;
  static constexpr int T = 64 * NT, P = Pitch<HS>::P;
  static constexpr int FULL = T * P, HALF = (32 * NY) * P;   // an operand with all T rows / with this workgroup's own rows
  static constexpr int MERGE = 2 * S::MERGE_FLOATS_PER_GROUP;
  // WIN: the resident form (all-T operand + own rows) would not fit the CU's 160 KB beside the statistics: first products through
  // the window ring (product_chain_w); the second products' all-T operand (FULL) then has the area to itself
  static constexpr bool WIN = (FULL + HALF + 16 * S::GX) * 4 > 160 * 1024;
  static constexpr int RING = WIN ? Window<HS, NT>::RING : 0;
  static constexpr int OWN = WIN ? RING : FULL;            // offset of the own-row operand
  static constexpr int OPERANDS = WIN ? (RING + HALF > FULL ? RING + HALF : FULL) : FULL + HALF;
  static constexpr int AREA = OPERANDS > MERGE ? OPERANDS : MERGE;
};

// XCD-aware block order.  Workgroup L runs on XCD L % 8 (each XCD has its own L2).  The two halves of a (sample, head)
// pair read the same K / V (forward, dQ pass) or Q / dO (dK/dV pass): they get ids that agree mod 8 and are adjacent in that
// XCD's dispatch order, so the pair's shared operands come from HBM once and the second half hits L2.
__device__ __forceinline__ bool decode_block(int NH, int B, int& half, int& hd, int& b) {
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int pair = (slot >> 1) * 8 + xcd;
  half = slot & 1;
  hd = pair % NH;
  b = pair / NH;
  return pair < NH * B;
}
// PARTS workgroups per (sample, head), adjacent in their XCD's dispatch order like the two halves
template <int PARTS>
__device__ __forceinline__ bool decode_parts(int NH, int B, int& part, int& hd, int& b) {
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int pair = (slot / PARTS) * 8 + xcd;
  part = slot % PARTS;
  hd = pair % NH;
  b = pair / NH;
  return pair < NH * B;
}

// dev instrumentation: s_memtime stamps of waves 0 and 7 of workgroup (0,0,0) at the phase boundaries
__device__ __forceinline__ void stamp(const AttnArgs& a, int idx) {
  if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 &&
      ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7))
    a.dbg[((threadIdx.x >> 6) ? 16 : 0) + idx] = (long long)__builtin_amdgcn_s_memtime();
}

// ---------------------------------------------------------------------------------------------------------- forward
template <int HS, int NT, typename TIO, int NY = NT>
__global__ __launch_bounds__(NTHR) void attn_wg_fwd_kernel(const AttnArgs a) {
  // activations in the bf16 mode are bf16 in HBM: they are widened to fp32 on their way into LDS, the arithmetic is unchanged
  const TIO* io_q = reinterpret_cast<const TIO*>(a.q); const TIO* io_k = reinterpret_cast<const TIO*>(a.k);
  const TIO* io_v = reinterpret_cast<const TIO*>(a.v); const TIO* io_dO = reinterpret_cast<const TIO*>(a.dO);
  TIO* io_o = reinterpret_cast<TIO*>(a.o); TIO* io_dq = reinterpret_cast<TIO*>(a.dq); TIO* io_dk = reinterpret_cast<TIO*>(a.dk);
  TIO* io_dv = reinterpret_cast<TIO*>(a.dv);
  (void)io_q; (void)io_k; (void)io_v; (void)io_dO; (void)io_o; (void)io_dq; (void)io_dk; (void)io_dv;
  using S = Shape<HS, NT, NY>;
  using L = Lds<HS, NT, NY>;
  constexpr int NDT = S::NDT, G = S::G, GX = S::GX, T = 64 * NT, P = L::P;
  __shared__ __attribute__((aligned(16))) float sm[L::AREA];
  __shared__ float sm_stat[2][2][4][G];   // [slice max | slice sum][query group][key slice][query]
  int half, hd, b;
  if (!decode_parts<S::PARTS>(a.NH, a.B, half, hd, b)) return;   // half: this workgroup's part of the queries
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int qg = w >> 2, ks = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int q0 = half * 2 * G + qg * G;   // first query of this wave's group
  const int k0 = ks * GX;                 // first key of this wave's slice
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const size_t ld = a.ld;
  float* sK = sm;               // [T][P] (windowed form: the ring of K's column groups)
  float* sQ = sm + L::OWN;      // [2G][P]: the queries of this workgroup
  stamp(a, 0);
  f32x4 s[NT][NY];   // S^T: [key tile][query tile], acc row = key 4*l4 + r, lane column = query l15
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NY; ++y) s[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  Stager<HS, T, HS, NTHR> stV;   // V goes into flight under the last MFMAs of the first product, lands in LDS (over K / Q)
  auto src = [&](int, const TIO*& pf, size_t& lf, const TIO*& ph, size_t& lh) {
    pf = io_k + rowbase * ld + hd * HS; lf = ld;
    ph = io_q + (rowbase + half * 2 * G) * ld + hd * HS; lh = ld;
  };
  auto tail = [&]() { stV.issue(io_v + rowbase * ld + hd * HS, ld, tid); };
  if constexpr (L::WIN) {
    constexpr int PW = Window<HS, NT>::PW;
    product_chain_w<HS, NT, 1, TIO, NY>(
        sm, sQ, tid, src,
        [&](int, const float* win, int col0) { product_phase_w<HS, NT, Groups<HS>::COLS, NY>(win + k0 * PW, sQ + qg * G * P, col0, l15, l4, s); },
        tail);
  } else {
    product_chain<HS, NT, 1, TIO, NY>(
        sK, sQ, tid, src,
        [&](int, int col0) { product_phase<HS, NT, Groups<HS>::COLS, NY>(sK + k0 * P, sQ + qg * G * P, col0, l15, l4, s); },
        tail);
  }
  stamp(a, 2);
  // ---- softmax over keys, flash-style across the four key slices: every slice normalises by its OWN row maximum, the
  // (max, sum) pairs meet in LDS once, then each slice rescales by exp(m_slice - m) / l
  float mx[NY];
  const bool masked = a.kv_len != nullptr;   // the fusion transformers pass no key mask: no per-element key test there (block-uniform)
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    float m = -INFINITY;
    if (!masked) {
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = s[x][y][r] * a.scale;
          s[x][y][r] = v;
          m = fmaxf(m, v);
        }
    } else {
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + 16 * x + 4 * l4 + r;
          // kv_len == 0: the reference's masked_fill(-1e9) + softmax is uniform attention over all keys (model_vec.py:315-317)
          const float v = nokeys ? 0.f : (key < kvlen ? s[x][y][r] * a.scale : -INFINITY);
          s[x][y][r] = v;
          m = fmaxf(m, v);
        }
    }
    m = quad_max(m);
    const float msafe = m > -INFINITY ? m : 0.f;   // a fully masked slice: every exponent is exp(-inf - 0) = 0
    float t = 0.f;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = mmfn_exp(s[x][y][r] - msafe);   // masked keys: exp(-inf) = 0
        s[x][y][r] = e;
        t += e;
      }
    mx[y] = m;
    t = quad_sum(t);
    if (l4 == 0) { sm_stat[0][qg][ks][16 * y + l15] = m; sm_stat[1][qg][ks][16 * y + l15] = t; }
  }
  __syncthreads();   // statistics visible; K / Q no longer read
  stV.commit(sm, tid);
  Drop dr;
  dr.init(a, b, hd);
  const bool drop = dr.on;
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    const int qi = 16 * y + l15;
    float m = sm_stat[0][qg][0][qi];
#pragma unroll
    for (int o = 1; o < 4; ++o) m = fmaxf(m, sm_stat[0][qg][o][qi]);   // finite: some key of the row is unmasked
    float l = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {   // slice order: every slice computes the same l bit for bit
      l += sm_stat[1][qg][o][qi] * mmfn_exp(sm_stat[0][qg][o][qi] - m);   // masked slice: sum 0 * exp(-inf) = 0
    }
    const int q = q0 + qi;
    if (ks == 0 && l4 == 0 && a.lse) a.lse[((size_t)b * a.NH + hd) * T + q] = m + logf(l);
    const float fac = mmfn_exp(mx[y] - m) / l;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = s[x][y][r] * fac;
        if (drop) p *= dr.scale(q, k0 + 16 * x + 4 * l4 + r, T);
        s[x][y][r] = p;
      }
  }
  __syncthreads();   // V staged
  stamp(a, 3);
  // ---- O = P . V over this wave's key slice, then the four slices are summed
  f32x4 o[NY][NDT];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) o[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT, NY>(s, sm + k0 * P, l15, l4, o);
  stamp(a, 4);
  __syncthreads();   // V no longer read: the area becomes the merge parking space
  merge_store<HS, NY, TIO>(o, sm, qg, ks, lane, l15, l4, io_o + (rowbase + q0) * a.ldo + hd * HS, a.ldo);
  stamp(a, 5);
}

// ------------------------------------------------------------------------ forward with the projections as its prologue
// The narrow fusion transformers (n_embd 64 / 128, head size 16 / 32): ln1 -> key / query / value of this head -> attention in
// ONE launch (gpt_block.h; include/mmfn_hip.h mmfn_gpt_block_attn_fwd_f32).  A sample's 192 x C token rows fit LDS, so the
// workgroup of (sample, head, query half) normalises them itself (8 workgroups per sample repeat that: 24 rows per wave),
// projects K and V of its head for all T tokens and Q for its half on the matrix pipes - weights as the MFMA A operand, so a
// lane holds 4 consecutive head columns of a token: 16-byte stores into the staged operand layout and into the packed qkv tensor
// the backward reads - and continues exactly as attn_wg_fwd_kernel.  Waves 0-2: K (4 token tiles each), 3-5: V, 6-7: Q (3 each).
// Saved tensors: this workgroup writes its (query half) x (head columns) piece of a = ln1(x) and of key | query | value.
template <int C, int HS>
__global__ __launch_bounds__(NTHR) void gpt_attn_fwd_kernel(const GptArgs g) {
  constexpr int NT = 3, T = 64 * NT, NH = C / HS, PX = C + 4, NHT = HS / 16, NCHX = C / 64;
  using S = Shape<HS, NT>;
  using L = Lds<HS, NT>;
  constexpr int NDT = S::NDT, G = S::G, P = L::P;
  constexpr int XF = T * PX, KQV = (2 * T + T / 2) * P;
  constexpr int AREA = XF > KQV ? (XF > L::MERGE ? XF : L::MERGE) : (KQV > L::MERGE ? KQV : L::MERGE);
  __shared__ __attribute__((aligned(16))) float sm[AREA];
  __shared__ float sm_stat[2][2][4][G];
  int half, hd, b;
  if (!decode_block(NH, g.B, half, hd, b)) return;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const size_t rowbase = (size_t)b * T;
  // the projection weights of this wave (waves 0-2: key, 3-5: value, 6-7: query) start travelling before the LayerNorm phase
  const int sel = w < 3 ? 0 : (w < 6 ? 2 : 1);               // packed order: key | query | value
  GptWRing<C, NHT, 4, false> ring;
  ring.start(g.wqkv + (size_t)(sel * C + hd * HS) * C, C, l15, l4);
  // ---- x rows of the sample -> LDS
  {
    constexpr int Q = C / 4, UNITS = T * Q;
    const float* src = g.x + rowbase * C;
#pragma unroll
    for (int u0 = 0; u0 < UNITS; u0 += NTHR) {
      const int u = u0 + tid;
      *reinterpret_cast<f32x4*>(sm + (u / Q) * PX + 4 * (u % Q)) = ld4(src + (size_t)(u / Q) * C + 4 * (u % Q));
    }
  }
  gpt_barrier();
  // ---- a = ln1(x) in place: 4 rows per wave pass (16 lanes per row), 6 passes
  {
    f32x4 wv[NCHX], bv[NCHX];
#pragma unroll
    for (int i = 0; i < NCHX; ++i) { wv[i] = ld4(g.ln1_w + 64 * i + 4 * l15); bv[i] = ld4(g.ln1_b + 64 * i + 4 * l15); }
#pragma unroll
    for (int it = 0; it < T / 32; ++it) {
      const int t = 32 * it + 4 * w + l4;
      f32x4 v[NCHX];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NCHX; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(sm + t * PX + 64 * i + 4 * l15);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
      const float mu = gpt_row16_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NCHX; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mu; q += d * d; }
      const float rs = 1.0f / sqrtf(gpt_row16_sum(q) / (float)C + g.eps);
      const bool mine = (t / (T / 2)) == half;
#pragma unroll
      for (int i = 0; i < NCHX; ++i) {
        const int n = 64 * i + 4 * l15;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mu) * rs * wv[i][r] + bv[i][r];
        *reinterpret_cast<f32x4*>(sm + t * PX + n) = o;
        if (mine && n / HS == hd) *reinterpret_cast<f32x4*>(g.a + (rowbase + t) * C + n) = o;
      }
      if (mine && hd == 0 && l15 == 0) { g.mu1[rowbase + t] = mu; g.rs1[rowbase + t] = rs; }
    }
  }
  gpt_barrier();
  // ---- key / value (all T tokens) and query (this half) of head hd: [HS] x [token tile] accumulators, in registers until
  // every wave has finished reading a (their destination overlays it)
  const int tile0 = w < 6 ? 4 * (w % 3) : (T / 32) * half + 3 * (w - 6);   // first 16-token tile of this wave
  f32x4 pacc[NHT][4];
#pragma unroll
  for (int i = 0; i < NHT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) pacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    if (w < 6) {
      ring.template run<4>(sm + 16 * tile0 * PX, l15, l4, pacc);
    } else {
      f32x4 q3[NHT][3];
#pragma unroll
      for (int i = 0; i < NHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) q3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      ring.template run<3>(sm + 16 * tile0 * PX, l15, l4, q3);
#pragma unroll
      for (int i = 0; i < NHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) pacc[i][j] = q3[i][j];
    }
  }
  gpt_barrier();   // a is no longer read
  float* sK = sm;                    // [T][P]
  float* sQ = sm + T * P;            // [T/2][P]
  float* sV = sm + (T + T / 2) * P;  // [T][P]
  {
    float* dstbase = sel == 0 ? sK : (sel == 2 ? sV : sQ - (T / 2) * half * P);   // (query rows are stored relative to the half)
    const int ntile = w < 6 ? 4 : 3;
#pragma unroll
    for (int i = 0; i < NHT; ++i) {
      const f32x4 bias = ld4(g.bqkv + sel * C + hd * HS + 16 * i + 4 * l4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntile) {
          const int t = 16 * (tile0 + j) + l15, n = 16 * i + 4 * l4;
          const f32x4 v = pacc[i][j] + bias;
          *reinterpret_cast<f32x4*>(dstbase + t * P + n) = v;
          if ((t / (T / 2)) == half) *reinterpret_cast<f32x4*>(g.qkv + (rowbase + t) * 3 * C + sel * C + hd * HS + n) = v;
        }
    }
  }
  gpt_barrier();
  // ---- attention (attn_wg_fwd_kernel without a key mask; K, Q, V are staged already)
  const int qg = w >> 2, ks = w & 3;
  const int q0 = half * 2 * G + qg * G;
  const int k0 = ks * G;
  const float scale = 1.0f / sqrtf((float)HS);
  f32x4 s[NT][NT];
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int y = 0; y < NT; ++y) s[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  product_phase<HS, NT, HS>(sK + k0 * P, sQ + qg * G * P, 0, l15, l4, s);
  float mx[NT];
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    float m = -INFINITY;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = s[x][y][r] * scale;
        s[x][y][r] = v;
        m = fmaxf(m, v);
      }
    m = quad_max(m);
    float t = 0.f;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = mmfn_exp(s[x][y][r] - m);
        s[x][y][r] = e;
        t += e;
      }
    mx[y] = m;
    t = quad_sum(t);
    if (l4 == 0) { sm_stat[0][qg][ks][16 * y + l15] = m; sm_stat[1][qg][ks][16 * y + l15] = t; }
  }
  gpt_barrier();
  Drop dr;
  dr.init(g.attn_pdrop, g.rng_state, g.rng_stream, NH, b, hd);
  const bool drop = dr.on;
#pragma unroll
  for (int y = 0; y < NT; ++y) {
    const int qi = 16 * y + l15;
    float m = sm_stat[0][qg][0][qi];
#pragma unroll
    for (int o = 1; o < 4; ++o) m = fmaxf(m, sm_stat[0][qg][o][qi]);
    float l = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) l += sm_stat[1][qg][o][qi] * mmfn_exp(sm_stat[0][qg][o][qi] - m);
    const int q = q0 + qi;
    if (ks == 0 && l4 == 0) g.lse[((size_t)b * NH + hd) * T + q] = m + logf(l);
    const float fac = mmfn_exp(mx[y] - m) / l;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = s[x][y][r] * fac;
        if (drop) p *= dr.scale(q, k0 + 16 * x + 4 * l4 + r, T);
        s[x][y][r] = p;
      }
  }
  f32x4 o[NT][NDT];
#pragma unroll
  for (int y = 0; y < NT; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) o[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT>(s, sV + k0 * P, l15, l4, o);
  gpt_barrier();   // K, Q, V no longer read: the area becomes the merge parking space
  merge_store<HS, NT, float>(o, sm, qg, ks, lane, l15, l4, g.o + (rowbase + q0) * C + hd * HS, C);
}

// ---------------------------------------------------------------------------------------- backward, query-owned: dQ, delta
template <int HS, int NT, typename TIO, int NY = NT>
__global__ __launch_bounds__(NTHR) void attn_wg_dq_kernel(const AttnArgs a) {
  // activations in the bf16 mode are bf16 in HBM: they are widened to fp32 on their way into LDS, the arithmetic is unchanged
  const TIO* io_q = reinterpret_cast<const TIO*>(a.q); const TIO* io_k = reinterpret_cast<const TIO*>(a.k);
  const TIO* io_v = reinterpret_cast<const TIO*>(a.v); const TIO* io_dO = reinterpret_cast<const TIO*>(a.dO);
  TIO* io_o = reinterpret_cast<TIO*>(a.o); TIO* io_dq = reinterpret_cast<TIO*>(a.dq); TIO* io_dk = reinterpret_cast<TIO*>(a.dk);
  TIO* io_dv = reinterpret_cast<TIO*>(a.dv);
  (void)io_q; (void)io_k; (void)io_v; (void)io_dO; (void)io_o; (void)io_dq; (void)io_dk; (void)io_dv;
  using S = Shape<HS, NT, NY>;
  using L = Lds<HS, NT, NY>;
  constexpr int NDT = S::NDT, G = S::G, GX = S::GX, T = 64 * NT, P = L::P;
  __shared__ __attribute__((aligned(16))) float sm[L::AREA];
  __shared__ float sm_stat[2][2][4][G];   // [sum P dP | sum P][query group][key slice][query]
  int half, hd, b;
  if (!decode_parts<S::PARTS>(a.NH, a.B, half, hd, b)) return;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int qg = w >> 2, ks = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int q0 = half * 2 * G + qg * G;
  const int k0 = ks * GX;
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;   // constant scores: P = 1/T, dS = 0
  const size_t ld = a.ld;
  const size_t statbase = ((size_t)b * a.NH + hd) * T;
  float* sA = sm;               // [T][P]: K, then V, then K again
  float* sB = sm + L::OWN;      // [2G][P]: Q, then dO of this workgroup's queries
  float lse_y[NY];   // issued before the products: used right after them
#pragma unroll
  for (int y = 0; y < NY; ++y) lse_y[y] = a.lse[statbase + q0 + 16 * y + l15];
  f32x4 acc[2][NT][NY];   // [0] S^T = K Q^T, [1] dP^T = V dO^T
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NY; ++y) acc[p][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  Stager<HS, T, HS, NTHR> stK;   // K again, for dQ = dS K: in flight under the last MFMAs, committed once V is no longer read
  auto src = [&](int i, const TIO*& pf, size_t& lf, const TIO*& ph, size_t& lh) {
    if (i == 0) {
      pf = io_k + rowbase * ld + hd * HS; lf = ld;
      ph = io_q + (rowbase + half * 2 * G) * ld + hd * HS; lh = ld;
    } else {
      pf = io_v + rowbase * ld + hd * HS; lf = ld;
      ph = io_dO + (rowbase + half * 2 * G) * a.ldo + hd * HS; lh = a.ldo;
    }
  };
  auto tail = [&]() { stK.issue(io_k + rowbase * ld + hd * HS, ld, tid); };
  if constexpr (L::WIN) {
    constexpr int PW = Window<HS, NT>::PW;
    product_chain_w<HS, NT, 2, TIO, NY>(
        sm, sB, tid, src,
        [&](int i, const float* win, int col0) { product_phase_w<HS, NT, Groups<HS>::COLS, NY>(win + k0 * PW, sB + qg * G * P, col0, l15, l4, acc[i]); },
        tail);
  } else {
    product_chain<HS, NT, 2, TIO, NY>(
        sA, sB, tid, src,
        [&](int i, int col0) { product_phase<HS, NT, Groups<HS>::COLS, NY>(sA + k0 * P, sB + qg * G * P, col0, l15, l4, acc[i]); },
        tail);
  }
  __syncthreads();
  stK.commit(sA, tid);
  Drop dr;
  dr.init(a, b, hd);
  const bool drop = dr.on;
  const bool masked = a.kv_len != nullptr;
#pragma unroll
  for (int y = 0; y < NY; ++y) {
    const int q = q0 + 16 * y + l15;
    const float lse = lse_y[y];
    float dl = 0.f, ps = 0.f;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * x + 4 * l4 + r;
        float p;
        if (!masked) p = mmfn_exp(acc[0][x][y][r] * a.scale - lse);   // (block-uniform: no key mask was passed)
        else {
          const float ex = mmfn_exp((nokeys ? 0.f : acc[0][x][y][r] * a.scale) - lse);
          p = (nokeys || key < kvlen) ? ex : 0.f;
        }
        float dpv = acc[1][x][y][r];
        if (drop) dpv *= dr.scale(q, key, T);
        acc[0][x][y][r] = p;
        acc[1][x][y][r] = dpv;
        dl += p * dpv;
        ps += p;
      }
    dl = quad_sum(dl);
    ps = quad_sum(ps);
    if (l4 == 0) { sm_stat[0][qg][ks][16 * y + l15] = dl; sm_stat[1][qg][ks][16 * y + l15] = ps; }
  }
  __syncthreads();   // statistics visible, K staged
#pragma unroll
  for (int y = 0; y < NY; ++y) {
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
  f32x4 dq[NY][NDT];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) dq[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT, NY>(acc[0], sA + k0 * P, l15, l4, dq);
  __syncthreads();
  merge_store<HS, NY, TIO>(dq, sm, qg, ks, lane, l15, l4, io_dq + (rowbase + q0) * a.ldg + hd * HS, a.ldg);
}

// ------------------------------------------------------------------------------------------ backward, key-owned: dK, dV
template <int HS, int NT, typename TIO, int NY = NT>
__global__ __launch_bounds__(NTHR) void attn_wg_dkv_kernel(const AttnArgs a) {
  // activations in the bf16 mode are bf16 in HBM: they are widened to fp32 on their way into LDS, the arithmetic is unchanged
  const TIO* io_q = reinterpret_cast<const TIO*>(a.q); const TIO* io_k = reinterpret_cast<const TIO*>(a.k);
  const TIO* io_v = reinterpret_cast<const TIO*>(a.v); const TIO* io_dO = reinterpret_cast<const TIO*>(a.dO);
  TIO* io_o = reinterpret_cast<TIO*>(a.o); TIO* io_dq = reinterpret_cast<TIO*>(a.dq); TIO* io_dk = reinterpret_cast<TIO*>(a.dk);
  TIO* io_dv = reinterpret_cast<TIO*>(a.dv);
  (void)io_q; (void)io_k; (void)io_v; (void)io_dO; (void)io_o; (void)io_dq; (void)io_dk; (void)io_dv;
  using S = Shape<HS, NT, NY>;
  using L = Lds<HS, NT, NY>;
  constexpr int NDT = S::NDT, G = S::G, GX = S::GX, T = 64 * NT, P = L::P;
  __shared__ __attribute__((aligned(16))) float sm[L::AREA];
  __shared__ float sm_rows[8][2][GX];   // per wave: log-sum-exp and delta of the GX queries of its slice
  int half, hd, b;
  if (!decode_parts<S::PARTS>(a.NH, a.B, half, hd, b)) return;   // half: this workgroup's part of the keys
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int kg = w >> 2, qs = w & 3;
  const size_t rowbase = (size_t)b * T;
  const int k0 = half * 2 * G + kg * G;   // first key of this wave's group
  const int q0 = qs * GX;                 // first query of this wave's slice
  const int kvlen = a.kv_len ? min(T, a.kv_len[b]) : T;
  const bool nokeys = kvlen <= 0;
  const size_t ld = a.ld;
  const size_t statbase = ((size_t)b * a.NH + hd) * T;
  float* sA = sm;               // [T][P]: Q, then dO (kept for dV), then Q again
  float* sB = sm + L::OWN;      // [2G][P]: K, then V of this workgroup's keys
  // the wave's G per-query statistics go to a wave-private LDS strip now (one coalesced load each), and are picked up
  // after the products: no register cost, no global-load latency between the phases
  if (lane < GX) {
    sm_rows[w][0][lane] = a.lse[statbase + q0 + lane];
    sm_rows[w][1][lane] = a.delta[statbase + q0 + lane];
  }
  f32x4 acc[2][NT][NY];   // [0] S = Q K^T, [1] dP = dO V^T: [query tile][key tile], acc row = query 4*l4 + r, column = key
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NY; ++y) acc[p][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  Stager<HS, T, HS, NTHR> stA;   // Q again (for dK): issued before the dV product, committed after the dV merge
  Stager<HS, T / 2, HS, NTHR> stHalf;   // windowed form: the all-T operands of the second products arrive in two row halves (32 registers)
  auto src = [&](int i, const TIO*& pf, size_t& lf, const TIO*& ph, size_t& lh) {
    if (i == 0) {
      pf = io_q + rowbase * ld + hd * HS; lf = ld;
      ph = io_k + (rowbase + half * 2 * G) * ld + hd * HS; lh = ld;
    } else {
      pf = io_dO + rowbase * a.ldo + hd * HS; lf = a.ldo;
      ph = io_v + (rowbase + half * 2 * G) * ld + hd * HS; lh = ld;
    }
  };
  if constexpr (L::WIN) {
    // the products saw dO one window at a time: the whole matrix (for dV) goes into flight under the last group's MFMAs and the
    // softmax recomputation, and lands once nobody reads the ring any more
    constexpr int PW = Window<HS, NT>::PW;
    product_chain_w<HS, NT, 2, TIO, NY>(
        sm, sB, tid, src,
        [&](int i, const float* win, int col0) { product_phase_w<HS, NT, Groups<HS>::COLS, NY>(win + q0 * PW, sB + kg * G * P, col0, l15, l4, acc[i]); },
        [&]() { stHalf.issue(io_dO + rowbase * a.ldo + hd * HS, a.ldo, tid); });
  } else {
    product_chain<HS, NT, 2, TIO, NY>(
        sA, sB, tid, src,
        [&](int i, int col0) { product_phase<HS, NT, Groups<HS>::COLS, NY>(sA + q0 * P, sB + kg * G * P, col0, l15, l4, acc[i]); },
        [&]() {});
  }
  Drop dr;
  dr.init(a, b, hd);
  const bool drop = dr.on;
  const bool masked = a.kv_len != nullptr;
#pragma unroll
  for (int x = 0; x < NT; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = 16 * x + 4 * l4 + r;
      const int q = q0 + qi;
      const float lse = sm_rows[w][0][qi], dlt = sm_rows[w][1][qi];   // written by this wave (same lanes < G): in order
#pragma unroll
      for (int y = 0; y < NY; ++y) {
        const int key = k0 + 16 * y + l15;
        float p;
        if (!masked) p = mmfn_exp(acc[0][x][y][r] * a.scale - lse);
        else {
          const bool kin = nokeys || key < kvlen;
          const float ex = mmfn_exp((nokeys ? 0.f : acc[0][x][y][r] * a.scale) - lse);
          p = kin ? ex : 0.f;
        }
        float msc = 1.f;
        if (drop) msc = dr.scale(q, key, T);
        acc[0][x][y][r] = p * msc;                                                           // dV = (P o mask)^T dO
        acc[1][x][y][r] = nokeys ? 0.f : p * (acc[1][x][y][r] * msc - dlt) * a.scale;         // dK = dS^T Q
        if constexpr (HS == 128) {
          // materialise both values HERE: left alone, the compiler sinks these expressions to their uses (the second products'
          // MFMA operands), which keeps p, the mask scale, the log-sum-exp and delta of every element alive across the dV phase
          // (~60-80 registers: at head size 128 the pass then spills - 312 bytes per lane at T = 192 up to round 5)
          asm volatile("" : "+v"(acc[0][x][y][r]), "+v"(acc[1][x][y][r]));
        }
      }
    }
  f32x4 g[NY][NDT];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) g[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (L::WIN) {
    __syncthreads();        // the window ring and the own rows are no longer read
    stHalf.commit(sA, tid);    // dO for dV: rows 0 .. T/2 (in flight since the last group), then the rest
    stHalf.issue(io_dO + (rowbase + T / 2) * a.ldo + hd * HS, a.ldo, tid);
    stHalf.commit(sA + (T / 2) * P, tid);
    __syncthreads();
  } else {
    stA.issue(io_q + rowbase * ld + hd * HS, ld, tid);          // Q again (for dK), in flight under the dV product
  }
  second_phase<HS, NT, NY>(acc[0], sA + q0 * P, l15, l4, g);     // dO is staged
  if constexpr (L::WIN) {   // Q (for dK): its first half travels under the dV merge - and not earlier (registers): the index is laundered
    int t2 = tid;           // so that the loads cannot be hoisted above the product's MFMAs
    asm volatile("" : "+v"(t2));
    stHalf.issue(io_q + rowbase * ld + hd * HS, ld, t2);
  }
  __syncthreads();
  merge_store<HS, NY, TIO>(g, sm, kg, qs, lane, l15, l4, io_dv + (rowbase + k0) * a.ldg + hd * HS, a.ldg);
  __syncthreads();   // every owner has read the dV copies before the area is reused
  if constexpr (L::WIN) {
    stHalf.commit(sA, tid);
    stHalf.issue(io_q + (rowbase + T / 2) * ld + hd * HS, ld, tid);
    stHalf.commit(sA + (T / 2) * P, tid);
  } else {
    stA.commit(sA, tid);
  }
  __syncthreads();
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int j = 0; j < NDT; ++j) g[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  second_phase<HS, NT, NY>(acc[1], sA + q0 * P, l15, l4, g);
  __syncthreads();
  merge_store<HS, NY, TIO>(g, sm, kg, qs, lane, l15, l4, io_dk + (rowbase + k0) * a.ldg + hd * HS, a.ldg);
}

#ifdef MMFN_ATTN_STAMPS   // experiment builds (tools/experiments/attn_phases.sh): phase time stamps of the forward kernel
__device__ long long g_attn_dbg[32];
long long* debug_buffer() {
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_attn_dbg)) == hipSuccess ? (long long*)p : nullptr;
}
#else
long long* debug_buffer() { return nullptr; }   // (phase time stamps of the forward kernel: a development build option, off)
#endif

template <int HS, int NT, typename TIO, int NY>
int launch_io(int which, const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  a.dbg = which == 0 ? debug_buffer() : nullptr;
  dim3 grid(Shape<HS, NT, NY>::PARTS * 8 * ceil_div(a.NH * a.B, 8));
  if (which == 0) hipLaunchKernelGGL((attn_wg_fwd_kernel<HS, NT, TIO, NY>), grid, dim3(NTHR), 0, s, a);
  else if (which == 1) hipLaunchKernelGGL((attn_wg_dq_kernel<HS, NT, TIO, NY>), grid, dim3(NTHR), 0, s, a);
  else hipLaunchKernelGGL((attn_wg_dkv_kernel<HS, NT, TIO, NY>), grid, dim3(NTHR), 0, s, a);
  MMFN_LAUNCH_CHECK();
  return 0;
}

template <int HS, int NT, int NY = NT>
int launch(int which, const AttnArgs& a, hipStream_t s) {
  return a.io_bf16 ? launch_io<HS, NT, bf16_t, NY>(which, a, s) : launch_io<HS, NT, float, NY>(which, a, s);
}

template <int HS, int NT, int NY = NT>
constexpr bool fits_lds() {   // the largest of the three kernels: operand area + sm_stat[2][2][4][G] / sm_rows[8][2][GX]
  return (Lds<HS, NT, NY>::AREA + 16 * Shape<HS, NT, NY>::GX) * 4 <= 160 * 1024;
}

template <int HS>
int by_tokens(int which, const AttnArgs& a, hipStream_t s) {
  switch (a.T) {
    case 64: return launch<HS, 1>(which, a, s);
    case 128: return launch<HS, 2>(which, a, s);
    case 192: return launch<HS, 3>(which, a, s);
    // the rad variant's 256 tokens (and several frames / views per sample): FOUR workgroups per (sample, head), each owning T/4 = 64
    // queries (forward, dQ) or keys (dK/dV) against all 256 of the other side - at B = 16 that is 256 workgroups (the halves were
    // 128: half the chip), the key-owned pass holds two 4 x 2-tile accumulator sets (64 registers) instead of two 4 x 4 (128: it
    // spilled 20-75 registers at head sizes 16 / 32), and the operands ((256 + 64) rows x (HS + 4) floats) fit LDS up to head size
    // 64; at head size 128 (169 KB) the first products run through a ring of column-group windows (product_chain_w).
    case 256: return launch<HS, 4, 2>(which, a, s);
  }
  return -1;
}

}  // namespace

extern "C" int mmfn_gpt_block_attn_fwd_f32(const mmfn_gpt_block_desc* d, void* stream) {
  if (!d || mmfn_gpt_block_supported(d->C, d->NH, d->T) != 0 || d->B <= 0) return MMFN_EINVAL;
  if (d->attn_pdrop < 0.f || d->attn_pdrop >= 1.f || (d->attn_pdrop > 0.f && !d->rng_state)) return MMFN_EINVAL;
  const dim3 grid(2 * 8 * ceil_div(d->NH * d->B, 8));
  if (d->C == 64) hipLaunchKernelGGL((gpt_attn_fwd_kernel<64, 16>), grid, dim3(NTHR), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL((gpt_attn_fwd_kernel<128, 32>), grid, dim3(NTHR), 0, (hipStream_t)stream, *d);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_attn_debug_read(int64_t* out64) {   // dev only: copies the 32 stamps to host memory
  long long* p = debug_buffer();
  if (!p) return MMFN_EINVAL;
  return (int)hipMemcpy(out64, p, 32 * sizeof(long long), hipMemcpyDeviceToHost);
}

int mmfn_attn_wg_launch(int which, int hs, const AttnArgs& a, hipStream_t s) {
  switch (hs) {
    case 16: return by_tokens<16>(which, a, s);
    case 32: return by_tokens<32>(which, a, s);
    case 64: return by_tokens<64>(which, a, s);
    case 128: return by_tokens<128>(which, a, s);
  }
  return -1;
}
