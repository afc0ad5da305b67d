// Sensor ingest on the GPU (the reference does this on CPU workers, dataloader.py:271-308):
//   camera   u8 HWC 300x400x3 -> centre crop 256x256 -> f32 NHWC, ImageNet mean/std on 0..255
//            values exactly as model_vec.py:33-44 writes it ((x - mean) * (1/std), no /255)
//   LiDAR    XYZ(I) points -> 2x256x256 BEV histogram (z <= -2 / z > -2), np.histogramdd bin
//            rules (half-open, last bin right-closed, out-of-range dropped), min(count,5)/5
//   NCHW f32 module inputs -> NHWC (with optional per-channel normalisation)
// The splat privatises a 32-row band of both histograms in LDS (64 KB), so there are no global
// atomics, no zero-fill pass and no finalise pass: every band block scans the sample's points
// (256 KB, L2-resident) and writes its slice of the final f32 map once.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int BINS = 256;
constexpr int BAND = 32;

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

__global__ __launch_bounds__(NT) void lidar_splat_kernel(const float* __restrict__ pts, int N, int stride_f,
                                                         float* __restrict__ out /* [B,256,256,2] */, int flip_y) {
  __shared__ int hist[2][BAND][BINS];
  const int band = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * BAND * BINS; i += NT) (&hist[0][0][0])[i] = 0;
  __syncthreads();
  const float* P = pts + (size_t)b * N * stride_f;
  const int x_lo = band * BAND;
  for (int i = threadIdx.x; i < N; i += NT) {
    const float xf = P[(size_t)i * stride_f], yf0 = P[(size_t)i * stride_f + 1], zf = P[(size_t)i * stride_f + 2];
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
  for (int i = threadIdx.x; i < BAND * BINS; i += NT) {
    const int c0 = hist[0][i / BINS][i % BINS], c1 = hist[1][i / BINS][i % BINS];
    float2 v;
    v.x = (float)(c0 > 5 ? 5 : c0) / 5.0f;
    v.y = (float)(c1 > 5 ? 5 : c1) / 5.0f;
    *reinterpret_cast<float2*>(O + (size_t)i * 2) = v;
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
  hipLaunchKernelGGL(lidar_splat_kernel, dim3(BINS / BAND, B), dim3(NT), 0, (hipStream_t)stream, pts, N, stride_floats, out, flip_y);
  MMFN_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmfn_lane_to_vector_f32(const float* lane, float* vec, int64_t R, int n, void* stream) {
  if (n < 2) return MMFN_EINVAL;
  hipLaunchKernelGGL(lane_to_vector_kernel, dim3(grid_for(R * (n - 1))), dim3(NT), 0, (hipStream_t)stream, lane, vec, R, n);
  MMFN_LAUNCH_CHECK();
  return 0;
}
