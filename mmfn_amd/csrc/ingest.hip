// Sensor ingest on the GPU (the reference does this on CPU workers, dataloader.py:271-308):
//   camera   u8 HWC 300x400x3 -> centre crop 256x256 -> f32 NHWC, ImageNet mean/std on 0..255
//            values exactly as model_vec.py:33-44 writes it ((x - mean) * (1/std), no /255)
//   LiDAR    XYZ(I) points -> 2x256x256 BEV histogram (z <= -2 / z > -2), np.histogramdd bin
//            rules (half-open, last bin right-closed, out-of-range dropped), min(count,5)/5
//   NCHW f32 module inputs -> NHWC (with optional per-channel normalisation)
// Camera: a cropped row is 768 contiguous bytes that start 8-byte aligned (300x400 frames, 256 crop): a thread takes 16
// consecutive bytes with two 8-byte loads (a wave reads 1 KB contiguous) and writes the 16 floats they become as four
// 16-byte stores; flat byte i of a row is channel i mod 3, so no pixel-wise indexing is needed.
// LiDAR: the splat privatises a 16-row band of both histograms in LDS (32 KB), so there are no global atomics, no
// zero-fill pass and no finalise pass: 16 x B blocks of 1024 threads (256 blocks at batch 16: one per CU) each stream the
// sample's points as 16-byte loads (HBM sees them once, the other bands hit L2 / Infinity Cache), keep the ones that
// fall into their band and write their slice of the final f32 map once.  Integer counts: bit-exact, order-independent.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int BINS = 256;
constexpr int BAND = 16;
constexpr int SPLAT_NT = 1024;

__global__ __launch_bounds__(NT) void ingest_rgb_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int H,
                                                           int W, int crop, float m0, float m1, float m2, float i0, float i1,
                                                           float i2) {
  const int r0 = H / 2 - crop / 2, c0 = W / 2 - crop / 2;
  const int64_t total = (int64_t)B * crop * crop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % crop);
    const int y = (int)((i / crop) % crop);
    const int b = (int)(i / ((int64_t)crop * crop));
    const uint8_t* p = in + (((size_t)b * H + r0 + y) * W + c0 + x) * 3;
    float* o = out + i * 3;
    o[0] = ((float)p[0] - m0) * i0;
    o[1] = ((float)p[1] - m1) * i1;
    o[2] = ((float)p[2] - m2) * i2;
  }
}

// in [B, C, P] -> out [B, P, C], optional (x - mean[c]) * inv_std[c]
__global__ __launch_bounds__(NT) void nchw_to_nhwc_small_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                                                int P, const float* __restrict__ mean,
                                                                const float* __restrict__ inv_std) {
  const int64_t total = (int64_t)B * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P), b = (int)(i / P);
    for (int c = 0; c < C; ++c) {
      float v = in[((size_t)b * C + c) * P + p];
      if (mean) v = (v - mean[c]) * inv_std[c];
      out[i * C + c] = v;
    }
  }
}

// 16 bytes -> 16 floats per thread; rows must start 8-byte aligned and hold a multiple of 16 bytes (host checks)
__global__ __launch_bounds__(NT) void ingest_rgb_u8_vec_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int H,
                                                               int W, int crop, float m0, float m1, float m2, float i0, float i1,
                                                               float i2) {
  const int r0 = H / 2 - crop / 2, c0 = W / 2 - crop / 2;
  const int per_row = crop * 3 / 16;
  const int64_t total = (int64_t)B * crop * per_row;
  const float mean[3] = {m0, m1, m2}, istd[3] = {i0, i1, i2};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % per_row);
    const int64_t row = i / per_row;             // b * crop + y
    const int y = (int)(row % crop), b = (int)(row / crop);
    const uint8_t* p = in + (((size_t)b * H + r0 + y) * W + c0) * 3 + 16 * t;
    const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
    const uint32_t wds[4] = {lo.x, lo.y, hi.x, hi.y};
    float* o = out + row * (int64_t)crop * 3 + 16 * t;
    int ch = t % 3;   // byte 16 t of the row is channel (16 t) mod 3 = t mod 3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = (float)((wds[k] >> (8 * e)) & 0xffu);
        v[e] = (x - (ch == 0 ? mean[0] : ch == 1 ? mean[1] : mean[2])) * (ch == 0 ? istd[0] : ch == 1 ? istd[1] : istd[2]);
        ch = ch == 2 ? 0 : ch + 1;
      }
      *reinterpret_cast<f32x4*>(o + 4 * k) = v;
    }
  }
}

// VEC: points are 16-byte records (stride 4 floats, 16-byte aligned): one float4 load per point
template <bool VEC>
__global__ __launch_bounds__(SPLAT_NT) void lidar_splat_kernel(const float* __restrict__ pts, int nbatch, int N, int stride_f,
                                                               float* __restrict__ out /* [B,256,256,2] */, int flip_y) {
  __shared__ int hist[2][BAND][BINS];
  // XCD-aware block order: workgroup L runs on XCD L % 8 (each XCD has its own L2), so the 16 band blocks of one sample
  // are given ids that agree mod 8 - the sample's points are then fetched from HBM by ONE L2 and the other 15 blocks hit it
  // (with bands spread over all XCDs every L2 fetched every sample: 8x the traffic, rocprofv3 FETCH_SIZE)
  const int L = blockIdx.x;
  const int b = (L & 7) + 8 * (L / (8 * (BINS / BAND)));
  const int band = (L >> 3) % (BINS / BAND);
  if (b >= nbatch) return;
  for (int i = threadIdx.x; i < 2 * BAND * BINS; i += SPLAT_NT) (&hist[0][0][0])[i] = 0;
  __syncthreads();
  const float* P = pts + (size_t)b * N * stride_f;
  const int x_lo = band * BAND;
  // x bins of this band cover [-16 + x_lo/8, -16 + (x_lo + BAND)/8): a cheap fp32 pre-test with a one-bin margin drops the
  // 15/16 of the points that are nowhere near before the exact fp64 binning
  const float lo_f = -16.0f + (float)(x_lo - 1) * 0.125f, hi_f = -16.0f + (float)(x_lo + BAND + 1) * 0.125f;
  for (int i = threadIdx.x; i < N; i += SPLAT_NT) {
    float xf, yf0, zf;
    if (VEC) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(P + (size_t)i * 4);
      xf = p[0]; yf0 = p[1]; zf = p[2];
    } else {
      xf = P[(size_t)i * stride_f]; yf0 = P[(size_t)i * stride_f + 1]; zf = P[(size_t)i * stride_f + 2];
    }
    if (!(xf >= lo_f && xf <= hi_f)) continue;
    const float yf = flip_y ? -yf0 : yf0;
    // exact in fp64: edges are -16 + i/8 and -24 + i/8
    const double xs = ((double)xf + 16.0) * 8.0, ys = ((double)yf + 24.0) * 8.0;
    if (!(xs >= 0.0 && xs <= 256.0 && ys >= 0.0 && ys <= 256.0)) continue;
    int ix = (int)xs, iy = (int)ys;
    if (ix == BINS) ix = BINS - 1;  // right-closed last bin
    if (iy == BINS) iy = BINS - 1;
    if (ix < x_lo || ix >= x_lo + BAND) continue;
    atomicAdd(&hist[zf <= -2.0f ? 0 : 1][ix - x_lo][iy], 1);
  }
  __syncthreads();
  float* O = out + ((size_t)b * BINS + x_lo) * BINS * 2;
  // two bins (4 floats, 16 bytes) per thread and iteration
  for (int i = threadIdx.x; i < BAND * BINS / 2; i += SPLAT_NT) {
    const int r = (2 * i) / BINS, c = (2 * i) % BINS;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c0 = hist[0][r][c + e], c1 = hist[1][r][c + e];
      v[2 * e] = (float)(c0 > 5 ? 5 : c0) / 5.0f;
      v[2 * e + 1] = (float)(c1 > 5 ? 5 : c1) / 5.0f;
    }
    *reinterpret_cast<f32x4*>(O + (size_t)i * 4) = v;
  }
}

// VectorNet input: lane nodes [R, n, 5] -> vectors [R*(n-1), 7] = [x0,y0,x1,y1,f2,f3,f4]
// (model_vec.py:368-381 _lane_to_vector)
__global__ __launch_bounds__(NT) void lane_to_vector_kernel(const float* __restrict__ lane, float* __restrict__ vec, int64_t R,
                                                            int n) {
  const int64_t total = R * (n - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (n - 1);
    const int v = (int)(i % (n - 1));
    const float* a = lane + (r * n + v) * 5;
    const float* c = a + 5;
    float* o = vec + i * 7;
    o[0] = a[0]; o[1] = a[1]; o[2] = c[0]; o[3] = c[1]; o[4] = c[2]; o[5] = c[3]; o[6] = c[4];
  }
}
int grid_for(int64_t total) { return (int)(ceil_div64(total, NT) < 8192 ? ceil_div64(total, NT) : 8192); }
}  // namespace

extern "C" int mmfn_ingest_rgb_u8(const uint8_t* in, float* out, int B, int H, int W, int crop, void* stream) {
  if (H < crop || W < crop) return MMFN_EINVAL;
  const size_t first = ((size_t)(H / 2 - crop / 2) * W + (W / 2 - crop / 2)) * 3;
  if (((uintptr_t)in + first) % 8 == 0 && ((size_t)W * 3) % 8 == 0 && ((size_t)H * W * 3) % 8 == 0 && (crop * 3) % 16 == 0 &&
      (uintptr_t)out % 16 == 0) {
    const int64_t total = (int64_t)B * crop * (crop * 3 / 16);
    hipLaunchKernelGGL(ingest_rgb_u8_vec_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, in, out, B, H, W, crop,
                       0.485f, 0.456f, 0.406f, (float)(1.0 / 0.229), (float)(1.0 / 0.224), (float)(1.0 / 0.225));
    MMFN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(ingest_rgb_u8_kernel, dim3(grid_for((int64_t)B * crop * crop)), dim3(NT), 0, (hipStream_t)stream, in, out, B, H,
                     W, crop, 0.485f, 0.456f, 0.406f, (float)(1.0 / 0.229), (float)(1.0 / 0.224), (float)(1.0 / 0.225));
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int P, const float* mean, const float* inv_std,
                                     void* stream) {
  hipLaunchKernelGGL(nchw_to_nhwc_small_kernel, dim3(grid_for((int64_t)B * P)), dim3(NT), 0, (hipStream_t)stream, in, out, B, C, P,
                     mean, inv_std);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_lidar_splat_f32(const float* pts, int B, int N, int stride_floats, float* out, int flip_y, void* stream) {
  if (stride_floats < 3 || N < 0) return MMFN_EINVAL;
  if (B <= 0) return 0;
  const dim3 grid(8 * (BINS / BAND) * ceil_div(B, 8));
  if (stride_floats == 4 && (uintptr_t)pts % 16 == 0)
    hipLaunchKernelGGL(lidar_splat_kernel<true>, grid, dim3(SPLAT_NT), 0, (hipStream_t)stream, pts, B, N, stride_floats, out, flip_y);
  else
    hipLaunchKernelGGL(lidar_splat_kernel<false>, grid, dim3(SPLAT_NT), 0, (hipStream_t)stream, pts, B, N, stride_floats, out,
                       flip_y);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_lane_to_vector_f32(const float* lane, float* vec, int64_t R, int n, void* stream) {
  if (n < 2) return MMFN_EINVAL;
  hipLaunchKernelGGL(lane_to_vector_kernel, dim3(grid_for(R * (n - 1))), dim3(NT), 0, (hipStream_t)stream, lane, vec, R, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}
