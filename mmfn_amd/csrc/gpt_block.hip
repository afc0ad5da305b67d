// Row-block kernels of the fused GPT block for the narrow fusion transformers (n_embd 64 / 128): model_vec.py:112-133.
//
// Why.  At B = 32 a block of transformer 1 / 2 holds 0.9 / 3.4 GFLOP forward; as separate launches (ln, addmm x 4, attention,
// ...) every one of its 8 forward and 9 backward chain kernels sits at the 9-18 us launch floor (profiles/r05_gpt_tile_sweep.txt):
// 7.5 ms of the fp32 step for 10 % of its FLOPs.  Everything here is local to a token row except the attention itself, so a
// workgroup takes 32 rows through the whole row-local chain with the intermediate tensors in LDS:
//   forward   o -> proj (+bias, dropout, + x) -> x1 -> ln2 -> a2 -> mlp.0 (+bias, ReLU) -> h -> mlp.2 (+bias, dropout, + x1) -> x2
//   backward  [dqkv -> . Wqkv -> ln1 backward (+ g1) -> g, dropout mask -> gd]  (the block above in execution order)
//             [gd -> . W2, ReLU mask from h -> gh -> . W1 -> ln2 backward (+ g) -> g1, dropout mask -> gd2 -> . Wproj -> go]
// M / 32 = 192 workgroups of 8 waves.  The matrix work is v_mfma_f32_16x16x4_f32 with the WEIGHT rows as the A operand and the
// 32 token rows as the B operand (gpt_block.h), weights streamed from L2 (9 C^2 floats per workgroup forward: 147 / 590 KB;
// every workgroup reads the same matrices), activations from LDS.  A wave owns 1/8 of a GEMM's output columns for both 16-row
// tiles (or one 16 x 16 tile when the output has only four column tiles), so the three GEMMs of a phase chain are balanced over
// the 4 SIMDs by construction.  Saved tensors (a2, h, x1, gh, gd2, ...) still go to HBM: the weight gradients are separate GEMMs
// off the dependent chain.
#include "gpt_block.h"

namespace {

constexpr int NTHR = 512;
constexpr int R = 32;   // token rows per workgroup

#ifdef MMFN_GPT_STAMPS   // experiment builds (tools/experiments/gpt_phases.sh): s_memtime at the phase boundaries, workgroup 0, waves 0 / 7
__device__ long long g_gpt_dbg[64];
#define GPT_STAMP(i)                                                                                       \
  do {                                                                                                     \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 7)) \
      g_gpt_dbg[((threadIdx.x >> 6) ? 32 : 0) + (i)] = (long long)__builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define GPT_STAMP(i) do { } while (0)
#endif

// Which output tiles a wave owns in a [R = 32 rows] x [N columns] product: N / 16 >= 8 column tiles -> NWT = N / 128 adjacent
// column tiles x both row tiles; 4 column tiles (N = 64) -> one column tile x one row tile.
template <int N>
struct Own {
  static constexpr int NT_N = N / 16;
  static constexpr bool WIDE = NT_N >= 8;
  static constexpr int NWT = WIDE ? NT_N / 8 : 1;
  static constexpr int NTT = WIDE ? 2 : 1;
  static __device__ __forceinline__ int nt0(int w) { return WIDE ? w * NWT : (w & 3); }
  static __device__ __forceinline__ int tt0(int w) { return WIDE ? 0 : (w >> 2); }
};

template <int NWT, int NTT>
__device__ __forceinline__ void zero(f32x4 (*acc)[NTT]) {
#pragma unroll
  for (int i = 0; i < NWT; ++i)
#pragma unroll
    for (int j = 0; j < NTT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

struct DropKey {
  uint64_t key;
  float p, inv_keep;
  bool on;
  __device__ __forceinline__ void init(const uint64_t* state, uint32_t stream, float drop_p) {
    on = drop_p > 0.f;
    p = drop_p;
    key = on ? mmfn_rng_key(state, stream) : 0;
    inv_keep = on ? 1.0f / (1.0f - drop_p) : 1.f;
  }
  __device__ __forceinline__ f32x4 apply(f32x4 v, uint64_t idx) const {
    if (on) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= mmfn_dropout_scale(key, idx + r, p, inv_keep);
    }
    return v;
  }
};

// cooperative copy of R rows x W elements (global row pitch ld) into LDS with pitch W + (one 16-byte unit), 16 bytes per thread
template <int W, typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, size_t ld, T* dst, int tid) {
  constexpr int E = 16 / sizeof(T), Q = W / E, UNITS = R * Q;
#pragma unroll
  for (int u0 = 0; u0 < UNITS; u0 += NTHR) {
    const int u = u0 + tid;
    if (UNITS % NTHR == 0 || u < UNITS)
      *reinterpret_cast<f32x4*>(dst + (u / Q) * (W + E) + E * (u % Q)) = *reinterpret_cast<const f32x4*>(src + (size_t)(u / Q) * ld + E * (u % Q));
  }
}

// ---------------------------------------------------------------------------------------------------------- forward
// BF: the bf16 mode - o, a2, h and the weight shadows (wproj / w1 / w2 point at the [out][in] bf16 shadows) are bf16; x, x1, x2 fp32.
template <int C, bool BF>
__global__ __launch_bounds__(NTHR) void gpt_mlp_fwd_kernel(const GptArgs a) {
  typedef GptPrec<BF> PR;
  typedef typename PR::act A;
  constexpr int H = 4 * C, PC = C + 4, PA = C + PR::EPU, PH = H + PR::EPU, NCH = C / 64;
  __shared__ __attribute__((aligned(16))) A sO[R * PA];        // o rows, then a2
  __shared__ __attribute__((aligned(16))) float sX1[R * PC];   // x1
  __shared__ __attribute__((aligned(16))) A sH[R * PH];        // hidden activations
  const A* wproj = reinterpret_cast<const A*>(a.wproj); const A* w1 = reinterpret_cast<const A*>(a.w1);
  const A* w2 = reinterpret_cast<const A*>(a.w2);
  A* io_a2 = reinterpret_cast<A*>(a.a2); A* io_h = reinterpret_cast<A*>(a.h);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const size_t row0 = (size_t)blockIdx.x * R;
  using OC = Own<C>;
  using OH = Own<H>;
  const int nt0 = OC::nt0(w), tt0 = OC::tt0(w), hn0 = OH::nt0(w), ht0 = OH::tt0(w);
  GPT_STAMP(0);
  GptWRing<C, OC::NWT, (BF ? 2 : 4), false, BF> rproj;
  rproj.start(wproj + (size_t)(16 * nt0) * C, C, l15, l4);
  stage_rows<C>(reinterpret_cast<const A*>(a.o) + row0 * C, C, sO, tid);
  gpt_barrier();
  GPT_STAMP(1);
  // ---- x1 = x + drop(o . Wproj^T + b)
  GptWRing<C, OH::NWT, (OH::NWT >= 4 ? 2 : 4), false, BF> rfc1;
  {
    // (epilogue operands are requested before the product: their latency would otherwise be exposed after the last MFMA)
    f32x4 bias[OC::NWT], xres[OC::NWT][OC::NTT];
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i) {
      bias[i] = gpt_ld4(a.bproj + 16 * (nt0 + i) + 4 * l4);
#pragma unroll
      for (int j = 0; j < OC::NTT; ++j) xres[i][j] = gpt_ld4(a.x + (row0 + 16 * (tt0 + j) + l15) * C + 16 * (nt0 + i) + 4 * l4);
    }
    f32x4 acc[OC::NWT][OC::NTT];
    zero<OC::NWT, OC::NTT>(acc);
    rproj.template run<OC::NTT>(sO + 16 * tt0 * PA, l15, l4, acc);
    GPT_STAMP(2);
    rfc1.start(w1 + (size_t)(16 * hn0) * C, C, l15, l4);   // mlp.0's first chunks travel under the epilogue and the LayerNorm
    DropKey dk;
    dk.init(a.rng_state, a.rng_stream + 1, a.resid_pdrop);
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OC::NTT; ++j) {
        const int n = 16 * (nt0 + i) + 4 * l4, t = 16 * (tt0 + j) + l15;
        const size_t off = (row0 + t) * C + n;
        f32x4 v = acc[i][j] + bias[i];
        v = dk.apply(v, off);
        v += xres[i][j];
        gpt_st4(a.x1 + off, v);
        *reinterpret_cast<f32x4*>(sX1 + t * PC + n) = v;
      }
  }
  GPT_STAMP(3);
  gpt_barrier();
  GPT_STAMP(4);
  // ---- a2 = ln2(x1): wave w takes rows 4w .. 4w+3, 16 lanes per row
  {
    const int t = 4 * w + l4;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(sX1 + t * PC + 64 * i + 4 * l15);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = gpt_row16_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(gpt_row16_sum(q) / (float)C + a.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int n = 64 * i + 4 * l15;
      const f32x4 wv = gpt_ld4(a.ln2_w + n), bv = gpt_ld4(a.ln2_b + n);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mu) * rs * wv[r] + bv[r];
      stx4(io_a2 + (row0 + t) * C + n, o);
      stx4(sO + t * PA + n, o);
    }
    if (l15 == 0) { a.mu2[row0 + t] = mu; a.rs2[row0 + t] = rs; }
  }
  GPT_STAMP(5);
  gpt_barrier();
  GPT_STAMP(6);
  // ---- h = relu(a2 . W1^T + b1)
  GptWRing<H, OC::NWT, 6, false, BF> rfc2;
  {
    f32x4 bias[OH::NWT];
#pragma unroll
    for (int i = 0; i < OH::NWT; ++i) bias[i] = gpt_ld4(a.b1 + 16 * (hn0 + i) + 4 * l4);
    f32x4 acc[OH::NWT][OH::NTT];
    zero<OH::NWT, OH::NTT>(acc);
    rfc1.template run<OH::NTT>(sO + 16 * ht0 * PA, l15, l4, acc);
    GPT_STAMP(7);
    rfc2.start(w2 + (size_t)(16 * nt0) * H, H, l15, l4);
#pragma unroll
    for (int i = 0; i < OH::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OH::NTT; ++j) {
        const int n = 16 * (hn0 + i) + 4 * l4, t = 16 * (ht0 + j) + l15;
        f32x4 v = acc[i][j] + bias[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        stx4(io_h + (row0 + t) * H + n, v);
        stx4(sH + t * PH + n, v);
      }
  }
  GPT_STAMP(8);
  gpt_barrier();
  GPT_STAMP(9);
  // ---- x2 = x1 + drop(h . W2^T + b2)
  {
    f32x4 bias[OC::NWT];
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i) bias[i] = gpt_ld4(a.b2 + 16 * (nt0 + i) + 4 * l4);
    f32x4 acc[OC::NWT][OC::NTT];
    zero<OC::NWT, OC::NTT>(acc);
    rfc2.template run<OC::NTT>(sH + 16 * tt0 * PH, l15, l4, acc);
    GPT_STAMP(10);
    DropKey dk;
    dk.init(a.rng_state, a.rng_stream + 2, a.resid_pdrop);
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OC::NTT; ++j) {
        const int n = 16 * (nt0 + i) + 4 * l4, t = 16 * (tt0 + j) + l15;
        const size_t off = (row0 + t) * C + n;
        f32x4 v = acc[i][j] + bias[i];
        v = dk.apply(v, off);
        v += *reinterpret_cast<const f32x4*>(sX1 + t * PC + n);
        gpt_st4(a.x2 + off, v);
      }
  }
  GPT_STAMP(11);
}

// --------------------------------------------------------------------------------------------------------- backward
// LayerNorm backward over the workgroup's 32 rows, wave w rows 4w .. 4w+3, 16 lanes per row.
//   sGy: LDS [R][C+4] gradient of the LayerNorm output;  x / mean / rstd / lnw: the forward's input rows and statistics;
//   res(t, n): the residual gradient added to dx;  emit(t, n, dx): consumes the result (stores);  returns the dropped / plain
//   value whose column sums are wanted through emit's return value.
// Partial rows of the workgroup (dweight, dbias[, column sums]) -> part[blockIdx.x][rows][C].
template <int C, typename Res, typename Emit>
__device__ __forceinline__ void ln_bwd_rows(const float* sGy, const float* __restrict__ x, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, const float* __restrict__ lnw, size_t row0, int w,
                                            int l15, int l4, float* red /* [3][8][C] */, float* part, int part_rows, int tid,
                                            Res&& res, Emit&& emit) {
  constexpr int PC = C + 4, NCH = C / 64;
  const int t = 4 * w + l4;
  const float mu = mean[row0 + t], rs = rstd[row0 + t];
  f32x4 av[NCH], xh[NCH], dw[NCH], db[NCH], ds[NCH];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int n = 64 * i + 4 * l15;
    const f32x4 gy = *reinterpret_cast<const f32x4*>(sGy + t * PC + n);
    const f32x4 xv = gpt_ld4(x + (row0 + t) * C + n), wv = gpt_ld4(lnw + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xhat = (xv[r] - mu) * rs;
      dw[i][r] = gy[r] * xhat;
      db[i][r] = gy[r];
      const float aj = gy[r] * wv[r];
      xh[i][r] = xhat;
      av[i][r] = aj;
      c1 += aj;
      c2 += aj * xhat;
    }
  }
  c1 = gpt_row16_sum(c1) / (float)C;
  c2 = gpt_row16_sum(c2) / (float)C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int n = 64 * i + 4 * l15;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rs * (av[i][r] - c1 - xh[i][r] * c2);
    o += res(t, n);
    ds[i] = emit(t, n, o);
  }
  // rows of the wave (l4 = 0..3) -> one, then the 8 waves through LDS
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = dw[i][r], v1 = db[i][r], v2 = ds[i][r];
      v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);
      v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
      v2 += __shfl_xor(v2, 16, 64); v2 += __shfl_xor(v2, 32, 64);
      if (l4 == 0) {
        const int n = 64 * i + 4 * l15 + r;
        red[(0 * 8 + w) * C + n] = v0;
        red[(1 * 8 + w) * C + n] = v1;
        red[(2 * 8 + w) * C + n] = v2;
      }
    }
  gpt_barrier();
  for (int c = tid; c < part_rows * C; c += NTHR) {
    const int which = c / C, n = c % C;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[(which * 8 + k) * C + n];
    part[((size_t)blockIdx.x * part_rows + which) * C + n] = s;
  }
}

// BF: the bf16 mode - dqkv, gd / gd_below, gh, gd2, go and h are bf16, and wqkv / w2 / w1 / wproj point at the TRANSPOSED bf16
// shadows ([in][out]: every data gradient is an NT product over them); x, x1, g, g1, g_below and the LayerNorm arithmetic stay fp32.
template <int C, bool BF>
__global__ __launch_bounds__(NTHR) void gpt_bwd_rows_kernel(const GptArgs up, const GptArgs lo, int has_up, int has_lo) {
  typedef GptPrec<BF> PR;
  typedef typename PR::act A;
  constexpr int E = PR::EPU;
  constexpr int H = 4 * C, PC = C + 4, PA = C + E, PH = H + E, PQ = 3 * C + E;
  constexpr int BIG = R * PH > R * PQ ? R * PH : R * PQ;
  __shared__ __attribute__((aligned(16))) A sBig[BIG];           // dqkv rows (upper), then gh rows (lower)
  __shared__ __attribute__((aligned(16))) float sGy[R * PC];     // gradient entering a LayerNorm backward (ga, then ga2)
  __shared__ __attribute__((aligned(16))) float sG[R * PC];      // g: gradient at the lower block's output (residual path)
  __shared__ __attribute__((aligned(16))) A sGd[R * PA];         // gd (mlp.2's operand), later gd2 (proj's operand)
  __shared__ float red[3 * 8 * C];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const size_t row0 = (size_t)blockIdx.x * R;
  using OC = Own<C>;
  using OH = Own<H>;
  const int nt0 = OC::nt0(w), tt0 = OC::tt0(w), hn0 = OH::nt0(w), ht0 = OH::tt0(w);
  constexpr bool NN = !BF;
  // first element of this wave's weight tiles for a product with NOUT outputs contracting over NIN: fp32 - column 16 * tile of the
  // [NIN][NOUT] matrix (row pitch NOUT); bf16 - row 16 * tile of its transposed shadow [NOUT][NIN] (row pitch NIN)
  auto wbase = [](const float* wgt, int tile0, int nin) {
    return BF ? reinterpret_cast<const A*>(wgt) + (size_t)(16 * tile0) * nin : reinterpret_cast<const A*>(wgt) + 16 * tile0;
  };
  GptWRing<C, OH::NWT, (BF ? 2 : (OH::NWT >= 4 ? 5 : 4)), NN, BF> rw2;   // (started before the phase that precedes its product, gpt_block.h)
  if (has_up) {
    // ---- ga = dqkv . Wqkv  (contraction over the 3C outputs of the packed projection)
    GptWRing<3 * C, OC::NWT, (BF ? 4 : 10), NN, BF> rqkv;
    rqkv.start(wbase(up.wqkv, nt0, 3 * C), BF ? 3 * C : C, l15, l4);
    stage_rows<3 * C>(reinterpret_cast<const A*>(up.dqkv) + row0 * 3 * C, 3 * C, sBig, tid);
    gpt_barrier();
    {
      f32x4 acc[OC::NWT][OC::NTT];
      zero<OC::NWT, OC::NTT>(acc);
      rqkv.template run<OC::NTT>(sBig + 16 * tt0 * PQ, l15, l4, acc);
      if (has_lo) rw2.start(wbase(lo.w2, hn0, C), BF ? C : H, l15, l4);
#pragma unroll
      for (int i = 0; i < OC::NWT; ++i)
#pragma unroll
        for (int j = 0; j < OC::NTT; ++j)
          *reinterpret_cast<f32x4*>(sGy + (16 * (tt0 + j) + l15) * PC + 16 * (nt0 + i) + 4 * l4) = acc[i][j];
    }
    gpt_barrier();
    // ---- g = ln1 backward(ga) + g1;  gd = g under the mask of the block below (bf16 mode: also the rounding to the operand type)
    DropKey dk;
    dk.init(up.rng_state, up.rng_stream_below + 2, up.gd_below ? up.resid_pdrop : 0.f);
    A* gd_below = reinterpret_cast<A*>(up.gd_below);
    ln_bwd_rows<C>(
        sGy, up.x, up.mu1, up.rs1, up.ln1_w, row0, w, l15, l4, red, up.part_ln1, up.below_colsum ? 3 : 2, tid,
        [&](int t, int n) { return gpt_ld4(up.g1 + (row0 + t) * C + n); },
        [&](int t, int n, f32x4 o) {
          const size_t off = (row0 + t) * C + n;
          gpt_st4(up.g_below + off, o);
          *reinterpret_cast<f32x4*>(sG + t * PC + n) = o;
          f32x4 od = o;
          if (gd_below) {
            od = dk.apply(o, off);
            stx4(gd_below + off, od);
          }
          stx4(sGd + t * PA + n, od);
          return od;
        });
    if (!has_lo) return;
    gpt_barrier();
  } else {
    rw2.start(wbase(lo.w2, hn0, C), BF ? C : H, l15, l4);
    stage_rows<C>(lo.g + row0 * C, C, sG, tid);
    stage_rows<C>(reinterpret_cast<const A*>(lo.gd ? lo.gd : lo.g) + row0 * C, C, sGd, tid);
    gpt_barrier();
  }
  // ---- gh = (gd . W2) masked by h > 0   (W2 [C][4C]: contraction over its rows)
  GptWRing<H, OC::NWT, (BF ? 6 : 12), NN, BF> rw1;
  {
    const A* io_h = reinterpret_cast<const A*>(lo.h);
    A* io_gh = reinterpret_cast<A*>(lo.gh);
    f32x4 acc[OH::NWT][OH::NTT];
    zero<OH::NWT, OH::NTT>(acc);
    rw2.template run<OH::NTT>(sGd + 16 * ht0 * PA, l15, l4, acc);
    rw1.start(wbase(lo.w1, nt0, H), BF ? H : C, l15, l4);
#pragma unroll
    for (int i = 0; i < OH::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OH::NTT; ++j) {
        const int n = 16 * (hn0 + i) + 4 * l4, t = 16 * (ht0 + j) + l15;
        const f32x4 hv = ldx4(io_h + (row0 + t) * H + n);
        f32x4 v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = hv[r] > 0.f ? v[r] : 0.f;
        stx4(io_gh + (row0 + t) * H + n, v);
        stx4(sBig + t * PH + n, v);
      }
  }
  gpt_barrier();
  // ---- ga2 = gh . W1  (W1 [4C][C])
  GptWRing<C, OC::NWT, (BF ? 2 : 8), NN, BF> rproj;
  {
    f32x4 acc[OC::NWT][OC::NTT];
    zero<OC::NWT, OC::NTT>(acc);
    rw1.template run<OC::NTT>(sBig + 16 * tt0 * PH, l15, l4, acc);
    rproj.start(wbase(lo.wproj, nt0, C), C, l15, l4);
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OC::NTT; ++j)
        *reinterpret_cast<f32x4*>(sGy + (16 * (tt0 + j) + l15) * PC + 16 * (nt0 + i) + 4 * l4) = acc[i][j];
  }
  gpt_barrier();
  // ---- g1 = ln2 backward(ga2) + g;  gd2 = g1 under proj's mask
  {
    DropKey dk;
    dk.init(lo.rng_state, lo.rng_stream + 1, lo.gd2 ? lo.resid_pdrop : 0.f);
    A* gd2 = reinterpret_cast<A*>(lo.gd2);
    ln_bwd_rows<C>(
        sGy, lo.x1, lo.mu2, lo.rs2, lo.ln2_w, row0, w, l15, l4, red, lo.part_ln2, 3, tid,
        [&](int t, int n) { return *reinterpret_cast<const f32x4*>(sG + t * PC + n); },
        [&](int t, int n, f32x4 o) {
          const size_t off = (row0 + t) * C + n;
          gpt_st4(lo.g1 + off, o);
          f32x4 od = o;
          if (gd2) {
            od = dk.apply(o, off);
            stx4(gd2 + off, od);
          }
          stx4(sGd + t * PA + n, od);   // (gd was last read two barriers ago)
          return od;
        });
  }
  gpt_barrier();
  // ---- go = gd2 . Wproj
  {
    A* io_go = reinterpret_cast<A*>(lo.go);
    f32x4 acc[OC::NWT][OC::NTT];
    zero<OC::NWT, OC::NTT>(acc);
    rproj.template run<OC::NTT>(sGd + 16 * tt0 * PA, l15, l4, acc);
#pragma unroll
    for (int i = 0; i < OC::NWT; ++i)
#pragma unroll
      for (int j = 0; j < OC::NTT; ++j)
        stx4(io_go + (row0 + 16 * (tt0 + j) + l15) * C + 16 * (nt0 + i) + 4 * l4, acc[i][j]);
  }
}

bool shape_ok(const GptArgs& d) {   // the row-block kernels see B * T rows: any token count whose rows tile by 32
  return mmfn_gpt_block_rows_supported(d.C, d.T) == 0 && d.B > 0 && ((size_t)d.B * d.T) % R == 0;
}

}  // namespace

extern "C" int mmfn_gpt_debug_read(int64_t* out64) {   // dev only: copies the 64 stamps to host memory
#ifdef MMFN_GPT_STAMPS
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gpt_dbg)) != hipSuccess) return MMFN_EINVAL;
  return (int)hipMemcpy(out64, p, 64 * sizeof(long long), hipMemcpyDeviceToHost);
#else
  (void)out64;
  return MMFN_EINVAL;
#endif
}

extern "C" int mmfn_sizeof_gpt_block_desc(void) { return (int)sizeof(mmfn_gpt_block_desc); }

extern "C" int mmfn_gpt_block_supported(int C, int NH, int T) {
  return ((C == 64 || C == 128) && NH == 4 && T == 192) ? 0 : MMFN_EINVAL;
}
extern "C" int mmfn_gpt_block_rows_supported(int C, int T) {
  return ((C == 64 || C == 128) && T > 0 && T % 32 == 0) ? 0 : MMFN_EINVAL;
}

namespace {
template <bool BF>
int mlp_fwd_launch(const mmfn_gpt_block_desc* d, void* stream) {
  if (!d || !shape_ok(*d)) return MMFN_EINVAL;
  if (d->resid_pdrop < 0.f || d->resid_pdrop >= 1.f || (d->resid_pdrop > 0.f && !d->rng_state)) return MMFN_EINVAL;
  const dim3 grid((unsigned)((size_t)d->B * d->T / R));
  if (d->C == 64) hipLaunchKernelGGL((gpt_mlp_fwd_kernel<64, BF>), grid, dim3(NTHR), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL((gpt_mlp_fwd_kernel<128, BF>), grid, dim3(NTHR), 0, (hipStream_t)stream, *d);
  MMFN_LAUNCH_CHECK();
  return 0;
}

template <bool BF>
int bwd_rows_launch(const mmfn_gpt_block_desc* upper, const mmfn_gpt_block_desc* lower, void* stream) {
  const mmfn_gpt_block_desc* any = upper ? upper : lower;
  if (!any || !shape_ok(*any)) return MMFN_EINVAL;
  if (upper && lower && (upper->C != lower->C || upper->B != lower->B || upper->T != lower->T)) return MMFN_EINVAL;
  if (any->resid_pdrop < 0.f || any->resid_pdrop >= 1.f || (any->resid_pdrop > 0.f && !any->rng_state)) return MMFN_EINVAL;
  // bf16 mode: the operand copies are where a gradient changes type - they cannot be "the same tensor as g"
  if (BF && ((lower && (!lower->gd2 || (!upper && !lower->gd))) || (upper && lower && !upper->gd_below))) return MMFN_EINVAL;
  const dim3 grid((unsigned)((size_t)any->B * any->T / R));
  const mmfn_gpt_block_desc& u = upper ? *upper : *lower;
  const mmfn_gpt_block_desc& l = lower ? *lower : *upper;
  if (any->C == 64)
    hipLaunchKernelGGL((gpt_bwd_rows_kernel<64, BF>), grid, dim3(NTHR), 0, (hipStream_t)stream, u, l, upper ? 1 : 0, lower ? 1 : 0);
  else
    hipLaunchKernelGGL((gpt_bwd_rows_kernel<128, BF>), grid, dim3(NTHR), 0, (hipStream_t)stream, u, l, upper ? 1 : 0, lower ? 1 : 0);
  MMFN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmfn_gpt_block_mlp_fwd_f32(const mmfn_gpt_block_desc* d, void* stream) { return mlp_fwd_launch<false>(d, stream); }
extern "C" int mmfn_gpt_block_mlp_fwd_bf16(const mmfn_gpt_block_desc* d, void* stream) { return mlp_fwd_launch<true>(d, stream); }
extern "C" int mmfn_gpt_block_bwd_rows_f32(const mmfn_gpt_block_desc* upper, const mmfn_gpt_block_desc* lower, void* stream) {
  return bwd_rows_launch<false>(upper, lower, stream);
}
extern "C" int mmfn_gpt_block_bwd_rows_bf16(const mmfn_gpt_block_desc* upper, const mmfn_gpt_block_desc* lower, void* stream) {
  return bwd_rows_launch<true>(upper, lower, stream);
}
