// 3x3 stride-1 convolution of the bf16 training mode with the input patch of a pixel tile resident in LDS (bf16 operands,
// v_mfma_f32_32x32x16_bf16, fp32 accumulation).  Replaces, for the torchvision BasicBlock convolutions of the three ResNet trunks
// (model_vec.py:509-521,539-593) under torch.autocast-style arithmetic, the cuDNN convolution forward / backward-data dispatch
// TOGETHER with the elementwise pass that produces its input: native_batch_norm's apply (+ skip + ReLU) in the forward, the
// BatchNorm backward's apply in the data gradient.
//
// Why not the implicit GEMM of gemm_bf16.hip: that kernel gathers every input pixel nine times (once per tap) from L2 into LDS -
// the bf16 trunk convolutions were L2-bound (9-12 TB/s of L2 -> LDS traffic), and because the operands travel global -> LDS
// directly nothing can be applied to them on the way, so each ConvBN needed the apply launch and a round trip of the activation.
// Here a block owns BM = 64 / 128 output pixels (TI images x TH rows x TW columns) and BN output channels:
//   1. prologue: the (TH + 2) x (TW + 2) halo patch, ALL contraction channels, goes global -> registers -> LDS once.  On the way
//      the producer's elementwise function is applied (PRO 1: y = [relu](bn(co) [+ res]); PRO 2: dco = BatchNorm backward of
//      g), pixels outside the image become zeros of the APPLIED tensor, and the block of column tile 0 writes the applied
//      values of the pixels it owns to HBM (the activation the skip connection / the weight gradient / the backward mask read);
//   2. main loop over (tap, 64-channel chunk) in the k order of the implicit GEMM (so the accumulation sequence, and with it
//      every output bit, equals the gather kernel's on equal inputs): the A fragments are ds_read_b128s of the resident patch
//      at the tap's row offset, only the filter tiles stream through an LDS ring (global_load_lds, counted vmcnt waits);
//   3. the epilogue of gemm_bf16.hip: accumulators staged through LDS, 8 consecutive channels per lane, BatchNorm statistics
//      partial sums (mode 0) or the BatchNorm-backward reductions of the layer below (mode 2), optional residual, bf16 rows out.
// LDS patch layout: row = patch pixel, pitch = K * 2 bytes; the 16-byte unit q of row r lives at unit q ^ sw(r) with
// sw(r) = (r >> 1) & 7 for K = 64 (two rows per 256-byte bank line) and r & 15 otherwise, so the 16 consecutive patch rows a
// ds_read_b128 service group touches at one tap are 16 different bank positions.
#include <algorithm>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NT = 256;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0,
                                   0);
}
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int xcd = b & 7, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
__device__ __forceinline__ int w_swz(int r) { return (r >> 1) & 7; }
#define MMFN_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x70 | 0xF00)

struct Row8 { float v[8]; };
__device__ __forceinline__ Row8 unpack8(const uint4 u) {
  Row8 r;
  r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xFFFF0000u);
  r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xFFFF0000u);
  r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xFFFF0000u);
  r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 u;
  u.x = (unsigned)mmfn_f2bf(v[0]) | ((unsigned)mmfn_f2bf(v[1]) << 16);
  u.y = (unsigned)mmfn_f2bf(v[2]) | ((unsigned)mmfn_f2bf(v[3]) << 16);
  u.z = (unsigned)mmfn_f2bf(v[4]) | ((unsigned)mmfn_f2bf(v[5]) << 16);
  u.w = (unsigned)mmfn_f2bf(v[6]) | ((unsigned)mmfn_f2bf(v[7]) << 16);
  return u;
}
__device__ __forceinline__ float bfround(float f) { return __uint_as_float((unsigned)mmfn_f2bf(f) << 16); }

// pixel tile geometry (host side: halo_geom)
struct HaloGeom {
  int lTW, lTH;         // log2 of the tile's columns / rows
  int TI;               // images per tile
  int TW2, TH2;         // patch columns / rows per image (TW + 2, TH + 2)
  int PR;               // patch rows = TI * TH2 * TW2
  int tiles_x, tiles_y; // tiles per image row / column
  int w_off;            // byte offset of the filter ring in LDS (patch bytes rounded up to 1 KB)
  int NC;               // 64-channel chunks of the contraction = K / 64
};

template <int TM, int TN, int NSW, int PRO>
__global__ __launch_bounds__(NT) void conv16_halo_kernel(const mmfn_conv16_halo_desc d, const HaloGeom g) {
  constexpr int BN = TN * 64;
  constexpr int WSTAGE = BN * 128;
  constexpr int D = NSW == 2 ? 1 : NSW - 2;
  constexpr int PB = BN / 32;                    // 1 KB filter pieces per wave and stage
  constexpr int EPI_LD = TN * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem + g.w_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int K = d.K, pitch = K * 2;
  const bool narrow = K == 64;
  const int tiles_n = d.N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n;
  const int n0 = (bid - tile_m * tiles_n) * BN;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  int t = tile_m;
  const int txi = t % g.tiles_x; t /= g.tiles_x;
  const int tyi = t % g.tiles_y;
  const int b0 = (t / g.tiles_y) * g.TI, y0 = tyi * TH, x0 = txi * TW;
  const unsigned short* X = reinterpret_cast<const unsigned short*>(d.x);
  const unsigned short* Wp = reinterpret_cast<const unsigned short*>(d.w);

  // ---- filter ring: lane's row of each piece, swizzled chunk folded into the pointer (as gemm16_nt_kernel)
  const unsigned short* pb[PB];
  {
    const int lr = lane >> 3, ls = lane & 7;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int r = (wave + 4 * i) * 8 + lr;
      pb[i] = Wp + (size_t)(n0 + r) * 9 * K + ((ls ^ w_swz(r)) * 8);
    }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int NC = g.NC, NSTEP = 9 * NC;
  int s_wt = 0, s_chunk = 0;   // filter tap / chunk of the NEXT step to stage (steps are staged in order)
  auto stage = [&](int buf) {
    const int off = s_wt * K + s_chunk * 64;
    unsigned char* Bs = Ws + buf * WSTAGE;
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(pb[i] + off, Bs + (wave_u + 4 * i) * 1024);
    if (++s_chunk == NC) { s_chunk = 0; ++s_wt; }
  };
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < NSTEP) stage(s);

  // ---- prologue: the halo patch, all K channels, through registers into LDS
  {
    const int QPR = K >> 3;                 // 16-byte units per patch row (8, 16, 32, 64: divides 256)
    const int RPI = NT / QPR;               // patch rows per pass of the block
    const int q = tid & (QPR - 1);
    const int c0 = q * 8;
    int row = tid / QPR;
    // (image, patch row, patch column) of `row`, then advanced incrementally by RPI rows per pass
    const int per_img = g.TH2 * g.TW2;
    int pi = row / per_img, rem = row - pi * per_img;
    int py = rem / g.TW2, px = rem - py * g.TW2;
    const int dy = RPI / g.TW2, dx = RPI - dy * g.TW2;
    float ca[8], cb[8], cc[8], cd[8], ce[8];
    if (PRO == 1) {         // y = fma(x, alpha, beta) [+ res] [relu]   (bn_apply_kernel's spelling)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ca[e] = mmfn_bn_alpha(d.p_weight[c0 + e], d.p_rstd[c0 + e]);
        cb[e] = mmfn_bn_beta(d.p_bias[c0 + e], d.p_mean[c0 + e], ca[e]);
      }
    } else if (PRO == 2) {  // dco = (ge - m1 - xhat * m2) * (w * rstd)   (bn_bwd_apply_kernel)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ca[e] = d.p_rstd[c0 + e];
        cb[e] = d.p_mean[c0 + e];
        cc[e] = d.p_means[c0 + e];
        cd[e] = d.p_means[K + c0 + e];
        ce[e] = d.p_weight[c0 + e] * ca[e];
      }
    }
    const bool writer = n0 == 0;
    constexpr int U = 4;
    for (int r0 = 0; r0 < g.PR; r0 += U * RPI) {
      uint4 v0[U], v1[U], v2[U];
      ptrdiff_t src[U];
      int lrow[U];
      bool own[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool ok = row < g.PR && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
        lrow[u] = row < g.PR ? row : -1;
        own[u] = ok && py >= 1 && py <= TH && px >= 1 && px <= TW;
        src[u] = ok ? (((ptrdiff_t)(b0 + pi) * d.H + iy) * d.W + ix) * K + c0 : -1;
        // advance to the row of the next pass
        row += RPI;
        px += dx;
        if (px >= g.TW2) { px -= g.TW2; ++py; }
        py += dy;
        if (py >= g.TH2) { py -= g.TH2; ++pi; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v0[u] = make_uint4(0, 0, 0, 0);
        v1[u] = v0[u];
        v2[u] = v0[u];
        if (src[u] >= 0) {
          v0[u] = *reinterpret_cast<const uint4*>(X + src[u]);
          if (PRO == 1 && d.p_res) v1[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.p_res) + src[u]);
          if (PRO == 2) {
            v1[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.p_x) + src[u]);
            if (d.p_y) v2[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.p_y) + src[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lrow[u] < 0) continue;
        uint4 o = v0[u];
        if (PRO == 1 && src[u] >= 0) {
          const Row8 xv = unpack8(v0[u]);
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = mmfn_bn_affine(xv.v[e], ca[e], cb[e]);
          if (d.p_res) {
            const Row8 rv = unpack8(v1[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += rv.v[e];
          }
          if (d.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.0f);
          }
          o = pack8(f);
          if (writer && own[u] && d.a_out) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(d.a_out) + src[u]) = o;
        } else if (PRO == 2 && src[u] >= 0) {
          const Row8 gv = unpack8(v0[u]), xv = unpack8(v1[u]);
          float ge[8], f[8];
          if (d.p_y) {
            const Row8 yv = unpack8(v2[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ge[e] = yv.v[e] > 0.0f ? gv.v[e] : 0.0f;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ge[e] = gv.v[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (xv.v[e] - cb[e]) * ca[e];
            f[e] = (ge[e] - cc[e] - xh * cd[e]) * ce[e];
          }
          o = pack8(f);
          if (writer && own[u]) {
            if (d.a_out) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(d.a_out) + src[u]) = o;
            if (d.ge_out) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(d.ge_out) + src[u]) = pack8(ge);
          }
        }
        const int sw = narrow ? ((lrow[u] >> 1) & 7) : (lrow[u] & 15);
        *reinterpret_cast<uint4*>(smem + lrow[u] * pitch + ((q ^ sw) << 4)) = o;
      }
    }
  }
  __syncthreads();   // patch complete (and the first D filter tiles have landed: the barrier drains vmcnt)

  // ---- main loop
  int rbase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int p = wm * TM * 32 + i * 32 + l31;
    const int tx = p & (TW - 1), ty = (p >> g.lTW) & (TH - 1), ti = p >> (g.lTW + g.lTH);
    rbase[i] = (ti * g.TH2 + ty) * g.TW2 + tx;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int r_wt = 0, r_chunk = 0;   // filter tap / chunk of the step being multiplied
  for (int s0 = 0; s0 < NSTEP; s0 += NSW) {
#pragma unroll
    for (int u = 0; u < NSW; ++u) {
      const int s = s0 + u;
      if (s >= NSTEP) break;
      const int ahead = NSTEP - 1 - s;
      if (ahead >= D) stage((u + D) % NSW);
      if (NSW > 2) {
        if (ahead >= D) MMFN_WAIT_VMCNT(PB * D);
        else if (D > 1 && ahead == 1) MMFN_WAIT_VMCNT(PB);
        else MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
      }
      const unsigned char* Bs = Ws + u * WSTAGE;
      // the patch tap this filter tap multiplies: the forward reads pixel + (kh - 1, kw - 1), the data gradient (flip) pixel + (1 - kh, 1 - kw)
      const int at = d.flip ? 8 - r_wt : r_wt;
      const int akh = at / 3, akw = at - akh * 3;
      const int toff = akh * g.TW2 + akw;
      bf16x8 a[4][TM], b[4][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = rbase[i] + toff;
        const unsigned char* rp = smem + r * pitch;
        const int g4 = (narrow ? ((r >> 1) & 7) : (r & 15)) << 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int q = r_chunk * 8 + ks * 2 + h;
          a[ks][i] = *reinterpret_cast<const bf16x8*>(rp + ((q << 4) ^ g4));
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int r = wn * TN * 32 + j * 32 + l31;
          b[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + ((c ^ w_swz(r)) << 4));
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      if (++r_chunk == NC) { r_chunk = 0; ++r_wt; }
      if (NSW == 2) {
        MMFN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  __syncthreads();   // nobody still reads the patch / the ring: they become the epilogue's staging area

  // ---- epilogue (gemm16_nt_kernel's): accumulators -> LDS (one region per wave) -> 8 consecutive channels per lane
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
  __builtin_amdgcn_wave_barrier();
  constexpr int LPR = TN * 4;
  constexpr int RPP = 64 / LPR;
  const int cl = (lane % LPR) * 8;
  const int col = n0 + wn * TN * 32 + cl;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bnm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bnr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int smode = d.stats ? d.stats_mode : -1;
  if (smode == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { bnm[e] = d.bn2_mean[col + e]; bnr[e] = d.bn2_rstd[col + e]; }
  }
#pragma unroll
  for (int p = 0; p < TM * 32 / RPP; ++p) {
    const int rl = p * RPP + lane / LPR;
    const int pix = wm * TM * 32 + rl;
    const int tx = pix & (TW - 1), ty = (pix >> g.lTW) & (TH - 1), ti = pix >> (g.lTW + g.lTH);
    const size_t row = ((size_t)(b0 + ti) * d.H + y0 + ty) * d.W + x0 + tx;
    float v[8];
    {
      const int sw = (rl >> 1) & 1, q0 = cl >> 2;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(ep + rl * EPI_LD + ((q0 ^ sw) << 2));
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(ep + rl * EPI_LD + (((q0 + 1) ^ sw) << 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
    }
    if (smode == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
    }
    if (d.out_res) {
      const Row8 a = unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.out_res) + row * d.N + col));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a.v[e];
    }
    if (smode == 2) {   // the two reductions of the BatchNorm backward this data gradient enters (gemm16_nt_kernel, stats_mode 2)
      const Row8 xq = unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.bn2_x) + row * d.N + col));
      float ge[8];
      if (d.bn2_y) {
        const Row8 yq = unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(d.bn2_y) + row * d.N + col));
#pragma unroll
        for (int e = 0; e < 8; ++e) ge[e] = yq.v[e] > 0.f ? bfround(v[e]) : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) ge[e] = bfround(v[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xq.v[e] - bnm[e]) * bnr[e];
        s1[e] += ge[e];
        s2[e] += ge[e] * xh;
      }
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(d.out) + row * d.N + col) = pack8(v);
  }
  if (smode >= 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    if (lane < LPR) {
      double* p = d.stats + ((size_t)tile_m * 2 + wm) * 2 * d.N;
#pragma unroll
      for (int e = 0; e < 8; ++e) { p[col + e] = (double)s1[e]; p[d.N + col + e] = (double)s2[e]; }
    }
  }
}

int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

constexpr int LDS_LIMIT = 160 * 1024;

// tile 1 = 128 pixels x 64 channels, 2 = 128 x 128, 3 = 64 x 64, 4 = 64 x 128
void tile_dims(int tile, int* tm, int* tn) {
  *tm = (tile == 1 || tile == 2) ? 2 : 1;
  *tn = (tile == 2 || tile == 4) ? 2 : 1;
}

bool halo_geom(const mmfn_conv16_halo_desc& d, int tm, HaloGeom* g) {
  const int BM = tm * 64;
  const int lW = ilog2_exact(d.W), lH = ilog2_exact(d.H);
  if (lW < 0 || lH < 0 || d.W < 8 || d.H < 4) return false;
  const int TW = std::min(d.W, 16);
  const int TH = std::min(d.H, BM / TW);
  const int TI = BM / (TW * TH);
  if (TI < 1 || d.B % TI) return false;
  g->lTW = ilog2_exact(TW);
  g->lTH = ilog2_exact(TH);
  g->TI = TI;
  g->TW2 = TW + 2;
  g->TH2 = TH + 2;
  g->PR = TI * g->TH2 * g->TW2;
  g->tiles_x = d.W / TW;
  g->tiles_y = d.H / TH;
  g->w_off = (g->PR * d.K * 2 + 1023) & ~1023;
  g->NC = d.K / 64;
  return true;
}

int smem_bytes(const HaloGeom& g, int tm, int tn, int nsw) {
  const int ring = g.w_off + nsw * tn * 64 * 128;
  const int epi = tm * tn * 16384;
  return std::max(ring, epi);
}

// (tile, stages) of a launch: the descriptor's, else by problem size - enough blocks for the 256 CUs first, then the larger tile
bool pick_config(const mmfn_conv16_halo_desc& d, int* tile, int* nsw, HaloGeom* g) {
  if (d.K % 64 || d.N % 64 || d.K > 512 || d.K < 64 || d.B <= 0) return false;
  int cand[4], nc = 0;
  if (d.tile >= 1 && d.tile <= 4) cand[nc++] = d.tile;
  else {
    // measured (tools/halo_bench.py, B = 32): with 128 output channels the 128-channel tiles win (one patch prologue feeds twice
    // the filter columns): 128 x 128 while that still gives every CU two blocks, else 64 x 128; with 64 channels 128 pixels x 64
    const long pixels = (long)d.B * d.H * d.W;
    const bool wide = d.N % 128 == 0;
    if (wide && pixels / 128 * (d.N / 128) >= 512) cand[nc++] = 2;
    if (wide && pixels / 64 * (d.N / 128) >= 256) cand[nc++] = 4;
    if (pixels / 128 * (d.N / 64) >= 512) cand[nc++] = 1;
    cand[nc++] = 3;
  }
  for (int c = 0; c < nc; ++c) {
    int tm, tn;
    tile_dims(cand[c], &tm, &tn);
    if (d.N % (tn * 64)) continue;
    HaloGeom gg;
    if (!halo_geom(d, tm, &gg)) continue;
    // filter ring: the double buffer by default - the patch prologue, not the filter stream, is what a block waits for, and the
    // smaller footprint keeps more blocks (and the other lanes' kernels) resident: 32.4 vs 39.0 us on layer1's shape
    int st = d.stages;
    if (st < 2 || st > 4) st = 2;
    while (st > 2 && smem_bytes(gg, tm, tn, st) > LDS_LIMIT) --st;
    if (smem_bytes(gg, tm, tn, st) > LDS_LIMIT) continue;
    *tile = cand[c];
    *nsw = st;
    *g = gg;
    return true;
  }
  return false;
}

template <int TM, int TN, int NSW, int PRO>
int launch_one(const mmfn_conv16_halo_desc& d, const HaloGeom& g, hipStream_t s) {
  const int smem = smem_bytes(g, TM, TN, NSW);
  static int ready = 0;   // the largest dynamic LDS size this instantiation has been enabled for
  if (smem > ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16_halo_kernel<TM, TN, NSW, PRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_LIMIT) != hipSuccess)
      return MMFN_EINVAL;
    ready = LDS_LIMIT;
  }
  const int tiles_m = (d.B / g.TI) * g.tiles_y * g.tiles_x;
  hipLaunchKernelGGL((conv16_halo_kernel<TM, TN, NSW, PRO>), dim3(tiles_m * (d.N / (TN * 64))), dim3(NT), smem, s, d, g);
  return 0;
}
template <int TM, int TN, int PRO>
int launch_ns(const mmfn_conv16_halo_desc& d, const HaloGeom& g, int nsw, hipStream_t s) {
  if (nsw == 2) return launch_one<TM, TN, 2, PRO>(d, g, s);
  if (nsw == 3) return launch_one<TM, TN, 3, PRO>(d, g, s);
  return launch_one<TM, TN, 4, PRO>(d, g, s);
}
template <int PRO>
int launch_tile(const mmfn_conv16_halo_desc& d, const HaloGeom& g, int tile, int nsw, hipStream_t s) {
  if (tile == 1) return launch_ns<2, 1, PRO>(d, g, nsw, s);
  if (tile == 2) return launch_ns<2, 2, PRO>(d, g, nsw, s);
  if (tile == 3) return launch_ns<1, 1, PRO>(d, g, nsw, s);
  return launch_ns<1, 2, PRO>(d, g, nsw, s);
}

}  // namespace

extern "C" int mmfn_sizeof_conv16_halo_desc(void) { return (int)sizeof(mmfn_conv16_halo_desc); }

extern "C" int mmfn_conv3x3_halo_bf16_ok(const mmfn_conv16_halo_desc* d) {
  if (!d) return 0;
  int tile, nsw;
  HaloGeom g;
  return pick_config(*d, &tile, &nsw, &g) ? tile : 0;
}

extern "C" int mmfn_conv3x3_halo_bf16_stats_rows(const mmfn_conv16_halo_desc* d) {
  if (!d) return 0;
  int tile, nsw;
  HaloGeom g;
  if (!pick_config(*d, &tile, &nsw, &g)) return 0;
  return 2 * (d->B / g.TI) * g.tiles_y * g.tiles_x;
}

extern "C" int mmfn_conv3x3_halo_bf16(const mmfn_conv16_halo_desc* dp, void* stream) {
  if (!dp) return MMFN_EINVAL;
  const mmfn_conv16_halo_desc& d = *dp;
  if (!d.x || !d.w || !d.out || d.pro < 0 || d.pro > 2) return MMFN_EINVAL;
  if (d.pro == 1 && (!d.p_mean || !d.p_rstd || !d.p_weight || !d.p_bias)) return MMFN_EINVAL;
  if (d.pro == 2 && (!d.p_mean || !d.p_rstd || !d.p_weight || !d.p_means || !d.p_x)) return MMFN_EINVAL;
  if (d.stats && d.stats_mode != 0 && d.stats_mode != 2) return MMFN_EINVAL;
  if (d.stats && d.stats_mode == 2 && (!d.bn2_x || !d.bn2_mean || !d.bn2_rstd)) return MMFN_EINVAL;
  int tile, nsw;
  HaloGeom g;
  if (!pick_config(d, &tile, &nsw, &g)) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (d.pro == 0) rc = launch_tile<0>(d, g, tile, nsw, s);
  else if (d.pro == 1) rc = launch_tile<1>(d, g, tile, nsw, s);
  else rc = launch_tile<2>(d, g, tile, nsw, s);
  if (rc) return rc;
  MMFN_LAUNCH_CHECK();
  return 0;
}
