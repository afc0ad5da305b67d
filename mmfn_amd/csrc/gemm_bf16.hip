// GEMM / implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 with bf16 operands IN HBM (the bf16 training mode,
// BASELINE configs[2]): activations, saved tensors and the per-step weight shadows are bf16, accumulation is fp32, weight
// gradients leave in fp32.  Replaces the aten addmm / cuDNN convolution dispatches of the reference's training step
// (model_vec.py:82-89,121-123 Linear; :509-593 the torchvision BasicBlock convolutions) under torch.autocast-style arithmetic.
//
// Two kernel families:
//   NT  (mmfn_gemm16 forms 0-2)  C[M,N] = A[M,K] . B[N,K]^T, both operands k-contiguous: Linear forward, Linear data gradient
//       (over the transposed weight shadow), convolution forward (A = implicit im2col of the NHWC input, k = (tap, ci)) and
//       convolution data gradient of ANY stride (A = transposed-convolution gather of dY, k = (tap, co), B = the
//       [Ci][tap][Co] weight shadow).  Operands go global -> LDS directly (global_load_lds, 16 B per lane: a wave's piece
//       is 8 rows x 128 B), the LDS image is lane-linear and the XOR slot swizzle is applied to the SOURCE address, so the
//       fragment ds_read_b128s are conflict-free without padding; k-tile 64, double-buffered, one barrier per k-tile.
//       The epilogue stages the fp32 accumulators through the (now free) operand LDS so that bias / activation / mask /
//       dropout / residual run on 8 consecutive columns per lane and C leaves as 16-byte rows of bf16.
//   TN  (forms 3-4)  C[M,N] = sum_k A[k,M] . B[k,N]: the weight gradients (contraction over tokens / pixels, both operands
//       contraction-major in memory).  The tiles are staged as they lie in memory (rows = contraction index) and the MFMA
//       fragments are read with ds_read_b64_tr_b16, the gfx950 LDS transpose read: no register transposes, no second copy
//       of the activations.  The contraction is split over blockIdx.y; fp32 slabs are combined by a second small kernel.
#include <algorithm>

#include "common.h"

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int BK = 64;   // k-tile (bf16 elements): 128 B per operand row

__device__ __attribute__((aligned(256))) unsigned short g_zero16[128] = {0};   // padding / invalid taps read this

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0,
                                   0);
}

__device__ __forceinline__ int xcd_remap(int b, int nb) {
  // workgroup b runs on XCD b % 8 (observed dispatch policy; speed only): give each XCD a contiguous run of tiles
  const int xcd = b & 7, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

struct Row8 { float v[8]; };
__device__ __forceinline__ Row8 load8_bf16(const void* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  Row8 r;
  r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xFFFF0000u);
  r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xFFFF0000u);
  r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xFFFF0000u);
  r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ void store8_bf16(void* p, const float* v) {
  uint4 u;
  u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  u.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
  u.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

// ------------------------------------------------------------------------------------------ NT
// LDS stage: A rows [BM][64] then B rows [BN][64] bf16, 128 B per row, no padding.  Physical 16-byte slot s of row r holds
// logical k-chunk s ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 service group read 16 rows that are distinct mod 16 at
// one logical chunk, i.e. 16 different slot positions of the 256-byte bank row.
__device__ __forceinline__ int nt_swz(int r) { return (r >> 1) & 7; }

// vmcnt(n) only (expcnt / lgkmcnt untouched): gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define MMFN_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x70 | 0xF00)

// NS LDS stages, tiles kt+1 .. kt+NS-2 in flight while tile kt is multiplied: the step's GEMMs are a few hundred tiles each
// (1-2 blocks per CU), so it is the depth of the per-block operand stream, not occupancy, that hides the L2 / HBM latency.
// One raw s_barrier per k-tile; the global_load_lds pieces are retired with COUNTED vmcnt waits (a __syncthreads() would drain
// them all); the stage overwritten in iteration kt was last read in iteration kt-2, two barriers back.
// NS = 2 is the plain double buffer (tile kt+1 in flight, drained before the barrier that ends iteration kt): half the LDS, so
// twice the resident blocks - the better trade when the grid is many blocks per CU.  The host picks (desc.stages, tuning table).
template <int BM, int BN, int NS>
struct NtCfg {
  static constexpr int STAGE = (BM + BN) * 128;
  static constexpr int EPI = 4 * (BM / 64) * 32 * (BN / 64) * 32 * 4;
  static constexpr int SMEM = NS * STAGE > EPI ? NS * STAGE : EPI;
};

// Stride-2 data gradient by output-pixel parity.  An input pixel (ih, iw) receives only the taps with kh == (ih + pad) mod 2,
// kw == (iw + pad) mod 2: of the nine taps of a 3x3 filter one, two, two or four, of a 1x1 filter one or none - gathered
// naively, 3/4 of the k-tiles multiply the zero page.  So the rows of such a launch are taken in parity-pure tiles: row tile
// t holds BM pixels of class t & 3 (classes interleaved, so that the contiguous run of tiles an XCD gets and the dispatch order
// both mix heavy and light classes), and its k-loop visits the live taps only.
__device__ __forceinline__ int parity_pixel_row(const mmfn_gemm16_desc& d, int cls, int local) {
  const int w2 = d.W >> 1, hw2 = (d.H >> 1) * w2;
  const int b = local / hw2, rem = local - b * hw2;
  const int i = rem / w2, j = rem - i * w2;
  return (b * d.H + 2 * i + (cls >> 1)) * d.W + 2 * j + (cls & 1);
}

template <int FORM, int BM, int BN, int NS>
__global__ __launch_bounds__(NT) void gemm16_nt_kernel(const mmfn_gemm16_desc d, const int tiles_n, const int tap_arg) {
  const int tap_shift = tap_arg & 255;
  const bool par = FORM == 2 && (tap_arg >> 8) != 0;
  constexpr int TM = BM / 64, TN = BN / 64;      // 32x32 accumulator tiles per wave (2 x 2 waves)
  constexpr int PA = BM / 32, PB = BN / 32;      // 1 KB pieces (8 rows) per wave per stage
  constexpr int STAGE = NtCfg<BM, BN, NS>::STAGE;
  constexpr int D = NS == 2 ? 1 : NS - 2;
  constexpr int EPI_LD = TN * 32;                // fp32 staging row of one wave tile; 16-byte chunks XOR-swizzled by (row >> 1) & 1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n;
  const int n0 = (bid % tiles_n) * BN;
  // par: m0 counts the rows of the tile's parity class (local rows), Mlim = pixels per class
  const int pcls = par ? (tile_m & 3) : 0;
  const int m0 = (par ? (tile_m >> 2) : tile_m) * BM;
  const int Mlim = par ? (d.M >> 2) : d.M;
  int nkt = d.K / BK;
  unsigned long long live_taps = 0;   // par: 4 bits per live tap, in tap order
  if (par) {
    const int py = pcls >> 1, px = pcls & 1;
    int nlive = 0;
    for (int kh = (py + d.pad) & 1; kh < d.KH; kh += 2)
      for (int kw = (px + d.pad) & 1; kw < d.KW; kw += 2) live_taps |= (unsigned long long)(kh * d.KW + kw) << (4 * nlive++);
    nkt = nlive << tap_shift;
  }
  const bf16_t* A = reinterpret_cast<const bf16_t*>(d.A);
  const bf16_t* Bp = reinterpret_cast<const bf16_t*>(d.B);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  // ---- loader state: this lane's row of each piece, with the swizzled chunk offset folded into the pointer
  const bf16_t* pa[PA];
  int ay[PA], ax[PA];
  const bf16_t* pb[PB];
  const int lr = lane >> 3, ls = lane & 7;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = (wave + 4 * i) * 8 + lr;
    const int c = ls ^ nt_swz(r);
    int m = min(m0 + r, Mlim - 1);
    if (par) m = parity_pixel_row(d, pcls, m);
    ay[i] = ax[i] = 0;
    if (FORM == 0) {
      pa[i] = A + (size_t)m * d.lda + c * 8;
    } else if (FORM == 1) {   // conv forward: m = (b, oh, ow) over the OUTPUT pixels
      const int ohw = d.OH * d.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oh = rem / d.OW, ow = rem - oh * d.OW;
      ay[i] = oh * d.stride - d.pad;
      ax[i] = ow * d.stride - d.pad;
      pa[i] = A + (size_t)b * d.H * d.W * d.Cin + c * 8;
    } else {                  // data gradient: m = (b, ih, iw) over the INPUT pixels, source = dY [B, OH, OW, Cout]
      const int hw = d.H * d.W;
      const int b = m / hw, rem = m - b * hw;
      const int ih = rem / d.W, iw = rem - ih * d.W;
      ay[i] = ih + d.pad;
      ax[i] = iw + d.pad;
      pa[i] = A + (size_t)b * d.OH * d.OW * d.Cout + c * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int r = (wave + 4 * i) * 8 + lr;
    const int c = ls ^ nt_swz(r);
    pb[i] = Bp + (size_t)min(n0 + r, d.N - 1) * d.ldb + c * 8;
  }
  // ---- lean gather (convolution forms whose source pixel is lane base + a per-tap constant: forward of any stride, data gradient
  // of stride 1 or in parity tiles).  Measured on the first version (tools/experiments/conv16_pmc.sh): 61 VALU + 64 SALU
  // instructions per wave and k-tile around 4 MFMAs, almost all of it address arithmetic redone per piece - tap decode with an
  // integer division, two coordinates, four compares, a 64-bit multiply-add.  Here a lane keeps ONE pointer (its pixel at tap
  // (0, 0)) and ONE bit mask (which taps fall inside the image) per piece; a k-tile adds a wave-uniform offset that changes only
  // when the tap does, and tests one bit.
  const int ntaps = d.KH * d.KW;
  const bool lean = FORM != 0 && ntaps <= 32 && (FORM == 1 || d.stride == 1 || par);
  const bf16_t* pl[PA];
  unsigned vmask[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) { pl[i] = pa[i]; vmask[i] = 0; }
  if (lean) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      // source pixel of tap (kh, kw): FORM 1: (ay + kh, ax + kw) of x;  FORM 2, stride 1: (ay - kh, ax - kw) of dY;
      // FORM 2, parity tiles: lane pixel (2y + py, 2x + px) -> (y + (py + pad - kh) / 2, x + (px + pad - kw) / 2) of dY
      int by = ay[i], bx = ax[i];
      if (par) { by = (ay[i] - d.pad) >> 1; bx = (ax[i] - d.pad) >> 1; }
      const int sh = FORM == 1 ? d.H : d.OH, sw = FORM == 1 ? d.W : d.OW, sc = FORM == 1 ? d.Cin : d.Cout;
      pl[i] = pa[i] + ((ptrdiff_t)by * sw + bx) * sc;
      // a tap (kh, kw) is inside iff its row is and its column is: KH + KW tests instead of KH * KW
      unsigned mh = 0, mw = 0;
      for (int kh = 0; kh < d.KH; ++kh) {
        int y;
        if (FORM == 1) y = by + kh;
        else if (!par) y = by - kh;
        else {
          const int dy = (pcls >> 1) + d.pad - kh;
          if (dy & 1) continue;                     // not a tap row of this parity class
          y = by + dy / 2;
        }
        if ((unsigned)y < (unsigned)sh) mh |= 1u << kh;
      }
      for (int kw = 0; kw < d.KW; ++kw) {
        int x;
        if (FORM == 1) x = bx + kw;
        else if (!par) x = bx - kw;
        else {
          const int dx = (pcls & 1) + d.pad - kw;
          if (dx & 1) continue;
          x = bx + dx / 2;
        }
        if ((unsigned)x < (unsigned)sw) mw |= 1u << kw;
      }
      unsigned mk = 0;
      for (int kh = 0; kh < d.KH; ++kh)
        if ((mh >> kh) & 1u) mk |= mw << (kh * d.KW);
      vmask[i] = mk;
    }
  }
  // wave-uniform state of the NEXT k-tile to stage (stage() is called for k-tiles 0, 1, 2, ... in order)
  int s_seq = 0, s_chunk = 0, s_tap = 0, s_toff = 0, s_ktb0 = 0;
  auto set_tap = [&](int seq) {   // seq: index in the sequence of taps this block visits
    s_tap = par ? (int)((live_taps >> (4 * seq)) & 15) : seq;
    const int kh = s_tap / d.KW, kw = s_tap - kh * d.KW;
    if (FORM == 1) s_toff = (kh * d.W + kw) * d.Cin;
    else if (!par) s_toff = -(kh * d.OW + kw) * d.Cout;
    else s_toff = ((((pcls >> 1) + d.pad - kh) / 2) * d.OW + ((pcls & 1) + d.pad - kw) / 2) * d.Cout;
    s_ktb0 = s_tap << tap_shift;
  };
  if (lean) set_tap(0);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // (uniform by construction: the LDS-DMA destinations become scalars)

  auto stage_lean = [&](int buf) {
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + BM * 128;
    const ptrdiff_t aoff = (ptrdiff_t)(s_toff + s_chunk * BK);
    const int boff = (s_ktb0 + s_chunk) * BK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bf16_t* src = ((vmask[i] >> s_tap) & 1u) ? pl[i] + aoff : zero;
      glds16(src, As + (wave_u + 4 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(pb[i] + boff, Bs + (wave_u + 4 * i) * 1024);
    if (++s_chunk == (1 << tap_shift)) { s_chunk = 0; set_tap(++s_seq); }
  };

  auto stage = [&](int kt, int buf) {
    if (lean) { stage_lean(buf); return; }
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + BM * 128;
    constexpr bool live = true;   // (stage() is only called for existing k-tiles: the tail's counted waits shrink instead)
    int kh = 0, kw = 0, c0 = 0, ktb = kt;   // ktb: this k-tile's position in the filter operand
    if (FORM != 0) {   // wave-uniform tap of this k-tile (channels % 64 == 0: a k-tile never straddles a tap)
      int tap = kt >> tap_shift;
      const int chunk = kt - (tap << tap_shift);
      if (par) {
        tap = (int)((live_taps >> (4 * tap)) & 15);
        ktb = (tap << tap_shift) + chunk;
      }
      c0 = chunk * BK;
      kh = tap / d.KW;
      kw = tap - kh * d.KW;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bf16_t* src;
      if (FORM == 0) {
        src = live ? pa[i] + (size_t)kt * BK : zero;
      } else if (FORM == 1) {
        const int ih = ay[i] + kh, iw = ax[i] + kw;
        const bool ok = live && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
        src = ok ? pa[i] + ((size_t)ih * d.W + iw) * d.Cin + c0 : zero;
      } else {
        const int th = ay[i] - kh, tw = ax[i] - kw;
        bool ok = live && th >= 0 && tw >= 0;
        int oh = th, ow = tw;
        if (d.stride == 2) {
          ok = ok && !((th | tw) & 1);
          oh = th >> 1; ow = tw >> 1;
        } else if (d.stride != 1) {
          oh = th / d.stride; ow = tw / d.stride;
          ok = ok && oh * d.stride == th && ow * d.stride == tw;
        }
        ok = ok && oh < d.OH && ow < d.OW;
        src = ok ? pa[i] + ((size_t)oh * d.OW + ow) * d.Cout + c0 : zero;
      }
      glds16(src, As + (wave_u + 4 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(live ? pb[i] + (size_t)ktb * BK : zero, Bs + (wave_u + 4 * i) * 1024);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nkt) stage(s, s);
  if (NS == 2) {
    MMFN_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
  }
  // The k-loop is unrolled over the NS stages, so the stage a k-tile is read from / staged into is a compile-time constant: the
  // fragment reads take their stage as the instruction's immediate offset and the LDS-DMA destinations are literals (no per-tile
  // buffer bookkeeping or address adds; -10 instructions per wave and k-tile).
  for (int kt0 = 0; kt0 < nkt; kt0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kt = kt0 + u;
      if (kt >= nkt) break;
      // tiles kt+1 .. kt+D stay in flight; near the end fewer exist and the (immediate) wait count shrinks with them
      const int ahead = nkt - 1 - kt;
      if (ahead >= D) stage(kt + D, (u + D) % NS);
      if (NS > 2) {
        if (ahead >= D) MMFN_WAIT_VMCNT((PA + PB) * D);     // this wave's pieces of tile kt have landed ...
        else if (D > 1 && ahead == 1) MMFN_WAIT_VMCNT(PA + PB);
        else MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();                       // ... and everybody else's
      }
      const unsigned char* As = smem + u * STAGE;
      const unsigned char* Bs = As + BM * 128;
      // all fragment reads of the k-tile first (the compiler then retires them with counted lgkmcnt waits under the MFMAs)
      bf16x8 a[BK / 16][TM], b[BK / 16][TN];
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const int c = ks * 2 + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wm * TM * 32 + i * 32 + l31;
          a[ks][i] = *reinterpret_cast<const bf16x8*>(As + r * 128 + ((c ^ nt_swz(r)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int r = wn * TN * 32 + j * 32 + l31;
          b[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + ((c ^ nt_swz(r)) << 4));
        }
      }
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      if (NS == 2) {   // tile kt+1 landed, and nobody still reads the stage the next iteration overwrites
        MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  __syncthreads();      // nobody still reads operands: the stages become the epilogue's staging area

  // ---- epilogue: accumulators -> LDS (fp32, one region per wave) -> 8 consecutive columns per lane
  float* ep = reinterpret_cast<float*>(smem) + wave * (TM * 32 * EPI_LD);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, cc = j * 32 + l31;
        ep[rl * EPI_LD + ((((cc >> 2) ^ ((rl >> 1) & 1)) << 2) | (cc & 3))] = acc[i][j][r];
      }
  __builtin_amdgcn_wave_barrier();   // the wave reads back its own region only (LDS is in order per wave): no block barrier
  constexpr int LPR = TN * 4;            // lanes per row (8 columns each)
  constexpr int RPP = 64 / LPR;          // rows per pass
  const int f = d.flags;
  uint64_t key = 0;
  float inv_keep = 1.f;
  if (f & MMFN_EPI_DROPOUT) { key = mmfn_rng_key(d.rng_state, d.rng_stream); inv_keep = 1.0f / (1.0f - d.drop_p); }
  const int cl = (lane % LPR) * 8;
  const int col = n0 + wn * TN * 32 + cl;
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool col_ok = col < d.N;         // N % 8 == 0 is required: a lane's 8 columns are valid together
  if ((f & MMFN_EPI_BIAS) && col_ok) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + col), b1 = *reinterpret_cast<const f32x4*>(d.bias + col + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias[e] = b0[e]; bias[4 + e] = b1[e]; }
  }
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bnm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bnr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (d.stats && d.stats_mode == 2 && col_ok) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { bnm[e] = d.bn_mean[col + e]; bnr[e] = d.bn_rstd[col + e]; }
  }
#pragma unroll
  for (int p = 0; p < TM * 32 / RPP; ++p) {
    const int rl = p * RPP + lane / LPR;
    const int lrow = m0 + wm * TM * 32 + rl;
    if (lrow >= Mlim || !col_ok) continue;
    const int row = par ? parity_pixel_row(d, pcls, lrow) : lrow;
    float v[8];
    {
      const int sw = (rl >> 1) & 1, q0 = cl >> 2;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(ep + rl * EPI_LD + ((q0 ^ sw) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(ep + rl * EPI_LD + (((q0 + 1) ^ sw) << 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
    }
    if (d.stats && d.stats_mode == 0) {   // BatchNorm batch statistics of the raw convolution output
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
    }
    if (f & MMFN_EPI_BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[e];
    }
    if (f & MMFN_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (f & MMFN_EPI_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mmfn_gelu(v[e]);
    }
    if (f & MMFN_EPI_MASK_AUX) {
      const Row8 a = load8_bf16(reinterpret_cast<const bf16_t*>(d.aux) + (size_t)row * d.ldaux + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = a.v[e] > 0.f ? v[e] : 0.f;
    }
    if (f & MMFN_EPI_DROPOUT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= mmfn_dropout_scale(key, (uint64_t)row * (uint64_t)d.N + (uint64_t)(col + e), d.drop_p, inv_keep);
    }
    if (f & MMFN_EPI_RESIDUAL) {
      if (f & MMFN_EPI16_RES_F32) {   // the transformers' fp32 residual stream (with MMFN_EPI16_OUT_F32)
        const float* rp = reinterpret_cast<const float*>(d.res) + (size_t)row * d.ldr + col;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
      } else {
        const Row8 a = load8_bf16(reinterpret_cast<const bf16_t*>(d.res) + (size_t)row * d.ldr + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a.v[e];
      }
    }
    if (d.stats && d.stats_mode != 0 && !(f & (MMFN_EPI_ACCUM | MMFN_EPI_RELU_LAST))) {
      if (d.stats_mode == 1) {          // column sums of the final value: the bias gradient of the Linear whose dX this is
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      } else {                          // the reductions of the BatchNorm backward this gradient enters (bn_bwd's col_partial pass)
        const Row8 xq = load8_bf16(reinterpret_cast<const bf16_t*>(d.bn_x) + (size_t)row * d.ldc + col);
        float ge[8];
        if (d.bn_y) {
          const Row8 yq = load8_bf16(reinterpret_cast<const bf16_t*>(d.bn_y) + (size_t)row * d.ldc + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) ge[e] = yq.v[e] > 0.f ? bf2f(f2bf(v[e])) : 0.f;   // the STORED (bf16) gradient, as bn_bwd reads it
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) ge[e] = bf2f(f2bf(v[e]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (xq.v[e] - bnm[e]) * bnr[e];
          s1[e] += ge[e];
          s2[e] += ge[e] * xh;
        }
      }
    }
    if (f & MMFN_EPI16_OUT_F32) {
      float* c = reinterpret_cast<float*>(d.C) + (size_t)row * d.ldc + col;
      if (f & MMFN_EPI_ACCUM) {
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(c), o1 = *reinterpret_cast<const f32x4*>(c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
      }
      if (f & MMFN_EPI_RELU_LAST) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
      *reinterpret_cast<f32x4*>(c) = o0;
      *reinterpret_cast<f32x4*>(c + 4) = o1;
    } else {
      bf16_t* c = reinterpret_cast<bf16_t*>(d.C) + (size_t)row * d.ldc + col;
      if (f & MMFN_EPI_ACCUM) {
        const Row8 a = load8_bf16(c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a.v[e];
      }
      if (f & MMFN_EPI_RELU_LAST) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      store8_bf16(c, v);
    }
  }
  if (d.stats) {
    // per-(row tile, wave row) partial column sums: lanes that share a column group combine through the wave, the two wave
    // rows of the block write separate partial rows -> [tiles_m * 2][2][N] doubles, finished by mmfn_bn_finalize_stats_f32
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    if (lane < LPR && col_ok) {
      double* p = d.stats + ((size_t)tile_m * 2 + wm) * 2 * d.N;
#pragma unroll
      for (int e = 0; e < 8; ++e) { p[col + e] = (double)s1[e]; p[d.N + col + e] = (double)s2[e]; }
    }
  }
}

// ------------------------------------------------------------------------------------------ TN (weight gradients)
// LDS stage: A tile [64 contraction rows][BM] then B tile [64][BN] bf16, rows as they lie in memory.  Physical 16-byte slot s
// of contraction row k holds logical column chunk s ^ swz(k); swz spreads the four rows a transpose read touches over the
// bank row (256-byte rows: (k & 3) << 2; 128-byte rows, two per bank row: ((k >> 1) & 1) << 2).
template <int BW>
__device__ __forceinline__ int tn_swz(int k) { return BW == 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2); }

// MFMA operand fragment (32 output indices x 16 contraction indices, lane l: index l & 31, contraction (l >> 5) * 8 .. + 7) from
// a contraction-major LDS tile through two ds_read_b64_tr_b16: lane w of a 16-lane group supplies the address of the 64-bit
// word (row w / 4, 4 columns starting at 4 * (w % 4)) and receives column w of the group's 4 x 16 block.
template <int BW>
__device__ __forceinline__ bf16x8 tn_fragment(const unsigned char* tile, int col0, int k0, int lane) {
  const int w = lane & 15, g2 = (lane >> 4) & 1, hh = lane >> 5;
  const int col = col0 + g2 * 16 + (w & 3) * 4;
  bf16x8 out;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = k0 + hh * 8 + q * 4 + (w >> 2);
    const unsigned char* p = tile + k * (BW * 2) + ((((col >> 3) ^ tn_swz<BW>(k)) << 4) | ((col & 7) << 1));
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
#pragma unroll
    for (int e = 0; e < 4; ++e) out[q * 4 + e] = v[e];
  }
  return out;
}

template <int FORM, int BM, int BN, int NS>
__global__ __launch_bounds__(NT) void gemm16_tn_kernel(const mmfn_gemm16_desc d, const int tiles_n, const int kt_per_split,
                                                       const int log2_ow, const int log2_ohw) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int RA = 1024 / (BM * 2), RB = 1024 / (BN * 2);   // contraction rows per 1 KB piece
  constexpr int PA = 64 / RA / 4, PB = 64 / RB / 4;           // pieces per wave per stage
  constexpr int STAGE = 64 * (BM + BN) * 2;
  constexpr int D = NS == 2 ? 1 : NS - 2;   // stages / tiles in flight, as in the NT kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int nkt = (d.K + 63) / 64;
  const int kt_begin = blockIdx.y * kt_per_split, kt_end = min(nkt, kt_begin + kt_per_split);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(d.A);
  const bf16_t* Bp = reinterpret_cast<const bf16_t*>(d.B);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);

  // lane -> (row in piece, slot): the piece is RA rows x (BM * 2 / 16) slots, lane-linear
  constexpr int SA = BM * 2 / 16, SB = BN * 2 / 16;
  const int ar = lane / SA, as = lane % SA, br = lane / SB, bs = lane % SB;
  int bkh = 0, bkw = 0, bci = 0;
  if (FORM == 1) {   // conv weight gradient: the column tile [n0, n0 + BN) lies inside one tap (Cin % BN == 0)
    const int tap = n0 / d.Cin;
    bci = n0 - tap * d.Cin;
    bkh = tap / d.KW;
    bkw = tap - bkh * d.KW;
  }

  // ---- lean staging (as in the NT kernel): a lane keeps one pointer per piece - its contraction row of k-tile 0 - and a k-tile
  // adds a wave-uniform offset.  The convolution form's row is an output pixel: with power-of-two OW and OH * OW >= 64 the 64
  // pixels of a k-tile are 64 / OW whole rows (or a 64-pixel stretch of one row) of ONE image, so the pixel splits into a uniform
  // part (image, first row / column of the group) and a lane part that never changes.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool whole = (d.K & 63) == 0;                                  // no ragged last k-tile
  const bool lean = whole && (FORM == 0 || log2_ohw >= 6);
  const bf16_t* paL[PA];
  const bf16_t* pbL[PB];
  int lys[PB], lxs[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int kl = (wave + 4 * i) * RA + ar;
    paL[i] = A + (size_t)kl * d.lda + min(m0 + (as ^ tn_swz<BM>(kl)) * 8, d.M - 8);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int kl = (wave + 4 * i) * RB + br;
    const int c = bs ^ tn_swz<BN>(kl);
    lys[i] = lxs[i] = 0;
    if (FORM == 0) {
      pbL[i] = Bp + (size_t)kl * d.ldb + min(n0 + c * 8, d.N - 8);
    } else {
      const int ly = log2_ow < 6 ? (kl >> log2_ow) : 0, lx = log2_ow < 6 ? (kl & ((1 << log2_ow) - 1)) : kl;
      lys[i] = ly * d.stride;
      lxs[i] = lx * d.stride;
      pbL[i] = Bp + ((ptrdiff_t)lys[i] * d.W + lxs[i]) * d.Cin + bci + c * 8;
    }
  }
  auto stage_lean = [&](int kt, int buf) {
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + 64 * BM * 2;
    const ptrdiff_t aoff = (ptrdiff_t)kt * 64 * d.lda;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(paL[i] + aoff, As + (wave_u + 4 * i) * 1024);
    if (FORM == 0) {
      const ptrdiff_t boff = (ptrdiff_t)kt * 64 * d.ldb;
#pragma unroll
      for (int i = 0; i < PB; ++i) glds16(pbL[i] + boff, Bs + (wave_u + 4 * i) * 1024);
    } else {
      const int gshift = log2_ohw - 6;                                  // 64-pixel groups per image = 1 << gshift
      const int b = kt >> gshift, p0 = (kt & ((1 << gshift) - 1)) << 6;    // image, first pixel of the group inside it
      const int uy = (p0 >> log2_ow) * d.stride - d.pad + bkh, ux = (p0 & ((1 << log2_ow) - 1)) * d.stride - d.pad + bkw;
      const ptrdiff_t boff = (((ptrdiff_t)b * d.H + uy) * d.W + ux) * d.Cin;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const bool ok = (unsigned)(uy + lys[i]) < (unsigned)d.H && (unsigned)(ux + lxs[i]) < (unsigned)d.W;
        glds16(ok ? pbL[i] + boff : zero, Bs + (wave_u + 4 * i) * 1024);
      }
    }
  };

  auto stage = [&](int kt, int buf) {
    if (lean) { stage_lean(kt, buf); return; }
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + 64 * BM * 2;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int piece = wave + 4 * i;
      const int kl = piece * RA + ar;               // contraction row inside the tile
      const int k = kt * 64 + kl;
      const int c = as ^ tn_swz<BM>(kl);            // logical column chunk this physical slot holds
      const int mcol = min(m0 + c * 8, d.M - 8);    // M % 8 == 0; columns beyond M are clamped (never stored)
      const bf16_t* src = (k < d.K && kt < kt_end) ? A + (size_t)k * d.lda + mcol : zero;
      glds16(src, As + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int piece = wave + 4 * i;
      const int kl = piece * RB + br;
      const int k = kt * 64 + kl;
      const int c = bs ^ tn_swz<BN>(kl);
      const bf16_t* src;
      if (FORM == 0) {
        const int ncol = min(n0 + c * 8, d.N - 8);
        src = (k < d.K && kt < kt_end) ? Bp + (size_t)k * d.ldb + ncol : zero;
      } else {   // k = output pixel (b, oh, ow); the row is x[b, oh*s - p + kh, ow*s - p + kw, bci + c*8 ...]
        const int b = k >> log2_ohw, rem = k & ((1 << log2_ohw) - 1);
        const int oh = rem >> log2_ow, ow = rem & ((1 << log2_ow) - 1);
        const int ih = oh * d.stride - d.pad + bkh, iw = ow * d.stride - d.pad + bkw;
        const bool ok = k < d.K && kt < kt_end && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
        src = ok ? Bp + ((size_t)(b * d.H + ih) * d.W + iw) * d.Cin + bci + c * 8 : zero;
      }
      glds16(src, Bs + piece * 1024);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (kt_begin + s < kt_end) stage(kt_begin + s, s);
  if (NS == 2) {
    MMFN_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
  }
  // unrolled over the NS stages (compile-time stage offsets), as the NT kernel
  for (int kt0 = kt_begin; kt0 < kt_end; kt0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kt = kt0 + u;
      if (kt >= kt_end) break;
      const int ahead = kt_end - 1 - kt;
      if (ahead >= D) stage(kt + D, (u + D) % NS);
      if (NS > 2) {
        if (ahead >= D) MMFN_WAIT_VMCNT((PA + PB) * D);
        else if (D > 1 && ahead == 1) MMFN_WAIT_VMCNT(PA + PB);
        else MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
      }
      const unsigned char* As = smem + u * STAGE;
      const unsigned char* Bs = As + 64 * BM * 2;
      bf16x8 a[4][TM], b[4][TN];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[ks][i] = tn_fragment<BM>(As, wm * TM * 32 + i * 32, ks * 16, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[ks][j] = tn_fragment<BN>(Bs, wn * TN * 32 + j * 32, ks * 16, lane);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      if (NS == 2) {
        MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  // fp32 output (a gradient) or a split slab; 32 consecutive columns per store instruction
  const bool to_slab = gridDim.y > 1;
  float* out = to_slab ? d.workspace + (size_t)blockIdx.y * d.M * d.N : reinterpret_cast<float*>(d.C);
  const int ldo = to_slab ? d.N : d.ldc;
  if (m0 + BM <= d.M && n0 + BN <= d.N) {   // interior tile: one pointer per lane, stores at compile-time row multiples of ldo
    float* p0 = out + (size_t)(m0 + wm * TM * 32 + 4 * h) * ldo + n0 + wn * TN * 32 + l31;
    const bool accum = !to_slab && (d.flags & MMFN_EPI_ACCUM);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = p0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32;
          *p = accum ? acc[i][j][r] + *p : acc[i][j][r];
        }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.M) continue;
        float v = acc[i][j][r];
        if (!to_slab && (d.flags & MMFN_EPI_ACCUM)) v += out[(size_t)row * ldo + col];
        out[(size_t)row * ldo + col] = v;
      }
    }
}

// slabs [splits][M][N] -> C (fp32, ldc), fixed order
__global__ __launch_bounds__(256) void gemm16_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C,
                                                            int ldc, int accum) {
  const size_t total4 = (size_t)M * N / 4;
  const size_t slab = (size_t)M * N;
  for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    const float* p = ws + i4 * 4;
    int z = 0;
    for (; z + 1 < splits; z += 2) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + (size_t)z * slab), b = *reinterpret_cast<const f32x4*>(p + (size_t)(z + 1) * slab);
      s0 += a;
      s1 += b;
    }
    if (z < splits) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)z * slab);
    s0 += s1;
    const size_t idx = i4 * 4;
    const int row = (int)(idx / N), col = (int)(idx - (size_t)row * N);
    float* c = C + (size_t)row * ldc + col;
    if (accum) s0 += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = s0;
  }
}

int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

// (tile id, splits) of a TN launch: enough blocks to fill 256 CUs about twice, at least 4 k-tiles per split
void tn_config(const mmfn_gemm16_desc& d, int bm, int bn, int* splits, int* kt_per) {
  const int nkt = (d.K + 63) / 64;
  const int tiles = ceil_div(d.M, bm) * ceil_div(d.N, bn);
  int sk = d.splitk > 0 ? d.splitk : std::max(1, std::min(nkt / 4, (512 + tiles - 1) / tiles));
  sk = std::max(1, std::min(sk, nkt));
  int per = ceil_div(nkt, sk);
  sk = ceil_div(nkt, per);
  *splits = sk;
  *kt_per = per;
}

void pick_tile(const mmfn_gemm16_desc& d, int* bm, int* bn) {
  int t = d.tile;
  if (t == 0) {
    // 128x128 when that still gives every CU a block (or the problem is too small to care), else 64x64
    const int big = ceil_div(d.M, 128) * ceil_div(d.N, 128);
    t = (d.N % 128 == 0 || d.N > 256) && big >= 200 ? 1 : 2;
    if (d.form >= 3) t = (d.M >= 128 && d.N >= 128 && (d.form == 3 || d.Cin % 128 == 0)) ? 1 : 2;
  }
  *bm = (t == 1 || t == 3) ? 128 : 64;
  *bn = (t == 1 || t == 4) ? 128 : 64;
}

}  // namespace

extern "C" int mmfn_sizeof_gemm16_desc(void) { return (int)sizeof(mmfn_gemm16_desc); }

extern "C" int64_t mmfn_gemm_bf16_workspace_bytes(const mmfn_gemm16_desc* d) {
  if (!d || d->form < 3) return 0;
  int bm, bn, sk, per;
  pick_tile(*d, &bm, &bn);
  tn_config(*d, bm, bn, &sk, &per);
  return sk > 1 ? (int64_t)sk * d->M * d->N * (int64_t)sizeof(float) : 0;
}

extern "C" int mmfn_gemm_bf16_stats_rows(const mmfn_gemm16_desc* d) {
  if (!d || d->form > 2) return 0;
  int bm, bn;
  pick_tile(*d, &bm, &bn);
  return 2 * ceil_div(d->M, bm);
}

// stride-2 data gradients run by pixel parity (parity_pixel_row) when every row tile can be parity-pure, else as the plain gather
bool parity_dgrad_ok(const mmfn_gemm16_desc& d, int bm) {
  if (d.form != 2 || d.stride != 2 || (d.H & 1) || (d.W & 1) || d.KH * d.KW > 9 || d.H <= 0 || d.W <= 0) return false;
  if (d.M % (d.H * d.W) || (d.M / 4) % bm) return false;
  return true;
}

template <int F, int BM_, int BN_, int NS_>
int launch_nt_ns(const mmfn_gemm16_desc& d, int tap_shift, hipStream_t s) {
  constexpr int smem = NtCfg<BM_, BN_, NS_>::SMEM;
  static bool ready = false;   // more than 64 KB of dynamic LDS needs the attribute, once per kernel
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_nt_kernel<F, BM_, BN_, NS_>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess)
      return MMFN_EINVAL;
    ready = true;
  }
  int tap_arg = tap_shift;
  if (F == 2 && parity_dgrad_ok(d, BM_)) tap_arg |= 1 << 8;
  hipLaunchKernelGGL((gemm16_nt_kernel<F, BM_, BN_, NS_>), dim3(ceil_div(d.M, BM_) * ceil_div(d.N, BN_)), dim3(NT), smem, s, d,
                     ceil_div(d.N, BN_), tap_arg);
  return 0;
}
template <int F, int BM_, int BN_>
int launch_nt(const mmfn_gemm16_desc& d, int tap_shift, int stages, hipStream_t s) {
  if (stages == 2) return launch_nt_ns<F, BM_, BN_, 2>(d, tap_shift, s);
  if (stages == 3) return launch_nt_ns<F, BM_, BN_, 3>(d, tap_shift, s);
  return launch_nt_ns<F, BM_, BN_, 4>(d, tap_shift, s);
}
#define LAUNCH_NT(F, BM_, BN_) rc_nt = launch_nt<F, BM_, BN_>(d, tap_shift, stages, s)
template <int F, int BM_, int BN_, int NS_>
int launch_tn_ns(const mmfn_gemm16_desc& d, int sk, int per, int l2ow, int l2ohw, hipStream_t s) {
  constexpr int smem = NS_ * 64 * (BM_ + BN_) * 2;
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_tn_kernel<F, BM_, BN_, NS_>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem) != hipSuccess)
      return MMFN_EINVAL;
    ready = true;
  }
  hipLaunchKernelGGL((gemm16_tn_kernel<F, BM_, BN_, NS_>), dim3(ceil_div(d.M, BM_) * ceil_div(d.N, BN_), sk), dim3(NT), smem, s, d,
                     ceil_div(d.N, BN_), per, l2ow, l2ohw);
  return 0;
}
template <int F, int BM_, int BN_>
int launch_tn(const mmfn_gemm16_desc& d, int sk, int per, int l2ow, int l2ohw, int stages, hipStream_t s) {
  if (stages == 2) return launch_tn_ns<F, BM_, BN_, 2>(d, sk, per, l2ow, l2ohw, s);
  if (stages == 3) return launch_tn_ns<F, BM_, BN_, 3>(d, sk, per, l2ow, l2ohw, s);
  return launch_tn_ns<F, BM_, BN_, 4>(d, sk, per, l2ow, l2ohw, s);
}
#define LAUNCH_TN(F, BM_, BN_) rc_tn = launch_tn<F, BM_, BN_>(d, sk, per, l2ow, l2ohw, stages, s)

extern "C" int mmfn_gemm_bf16(const mmfn_gemm16_desc* dp, void* stream) {
  if (!dp) return MMFN_EINVAL;
  const mmfn_gemm16_desc& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MMFN_EINVAL;
  int bm, bn;
  pick_tile(d, &bm, &bn);
  // LDS stages (desc.stages: 0 = auto): deep operand streams for grids of a few blocks per CU, the double buffer (half the
  // LDS, twice the resident blocks) for grids of many; the weight-gradient forms split the contraction into many blocks anyway
  int stages = d.stages;
  if (stages < 2 || stages > 4) {
    const int blocks = ceil_div(d.M, bm) * ceil_div(d.N, bn);
    stages = d.form <= 2 ? (blocks >= 1536 ? 2 : 4) : 2;
  }
  if (d.form <= 2) {
    if (d.K % BK || d.N % 8 || d.ldb % 8 || d.ldc % 8) return MMFN_EINVAL;
    int tap_shift = 0;
    if (d.form == 0) {
      if (d.lda % 8) return MMFN_EINVAL;
    } else {
      const int ch = d.form == 1 ? d.Cin : d.Cout;   // channels of the gathered tensor
      if (ch % BK) return MMFN_EINVAL;
      tap_shift = ilog2_exact(ch / BK);
      if (tap_shift < 0 || d.K != d.KH * d.KW * ch) return MMFN_EINVAL;
    }
    if ((d.flags & MMFN_EPI_RESIDUAL) && (!d.res || d.ldr % 8)) return MMFN_EINVAL;
    if ((d.flags & MMFN_EPI_MASK_AUX) && (!d.aux || d.ldaux % 8)) return MMFN_EINVAL;
    if (d.stats && d.stats_mode == 2 && (!d.bn_x || !d.bn_mean || !d.bn_rstd || (d.flags & MMFN_EPI16_OUT_F32))) return MMFN_EINVAL;
    if (d.stats && (d.stats_mode < 0 || d.stats_mode > 2)) return MMFN_EINVAL;
    if (d.stats && d.stats_mode != 0 && (d.flags & (MMFN_EPI_ACCUM | MMFN_EPI_RELU_LAST))) return MMFN_EINVAL;
    int rc_nt = 0;
#define NT_FORMS(BM_, BN_)                          \
  if (d.form == 0) LAUNCH_NT(0, BM_, BN_);          \
  else if (d.form == 1) LAUNCH_NT(1, BM_, BN_);     \
  else LAUNCH_NT(2, BM_, BN_)
    if (bm == 128 && bn == 128) { NT_FORMS(128, 128); }
    else if (bm == 128) { NT_FORMS(128, 64); }
    else if (bn == 128) { NT_FORMS(64, 128); }
    else { NT_FORMS(64, 64); }
    if (rc_nt) return rc_nt;
    MMFN_LAUNCH_CHECK();
    return 0;
  }
  if (d.form > 4) return MMFN_EINVAL;
  // TN: fp32 output only
  if (!(d.flags & MMFN_EPI16_OUT_F32) || d.M % 8 || d.lda % 8 || d.M < 8) return MMFN_EINVAL;
  int l2ow = 0, l2ohw = 0;
  if (d.form == 3) {
    if (d.N % 8 || d.ldb % 8 || d.N < 8) return MMFN_EINVAL;
  } else {
    l2ow = ilog2_exact(d.OW);
    l2ohw = ilog2_exact(d.OH * d.OW);
    if (l2ow < 0 || l2ohw < 0 || d.Cin % bn || d.N != d.KH * d.KW * d.Cin) return MMFN_EINVAL;
  }
  int sk, per;
  tn_config(d, bm, bn, &sk, &per);
  if (sk > 1 && !d.workspace) return MMFN_EINVAL;
  int rc_tn = 0;
#define TN_FORMS(BM_, BN_)                      \
  if (d.form == 3) LAUNCH_TN(0, BM_, BN_);      \
  else LAUNCH_TN(1, BM_, BN_)
  if (bm == 128 && bn == 128) { TN_FORMS(128, 128); }
  else if (bm == 128) { TN_FORMS(128, 64); }
  else if (bn == 128) { TN_FORMS(64, 128); }
  else { TN_FORMS(64, 64); }
  if (rc_tn) return rc_tn;
  MMFN_LAUNCH_CHECK();
  if (sk > 1) {
    if ((size_t)d.M * d.N % 4 || d.ldc % 4) return MMFN_EINVAL;
    const int blocks = (int)std::min<size_t>(((size_t)d.M * d.N / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(gemm16_reduce_kernel, dim3(blocks), dim3(256), 0, s, d.workspace, sk, d.M, d.N, reinterpret_cast<float*>(d.C),
                       d.ldc, (d.flags & MMFN_EPI_ACCUM) ? 1 : 0);
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}
