// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with NACC independent accumulators,
// 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int ITERS>
__global__ void k16(float* out, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (ITERS * NACC);
}
template <int NACC, int ITERS>
__global__ void k32(float* out, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  float a = a0 + threadIdx.x, b = b0;
  long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (ITERS * NACC);
}
template <typename F>
void run(const char* name, F kern, int threads, int blocks) {
  float* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, threads>>>(d, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, threads>>>(d, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  printf("%-40s threads/block %4d blocks %5d: %.1f clock64 ticks per MFMA per wave, %.3f ms\n", name, threads, blocks, cyc, ms);
  hipFree(d);
}
int main() {
  // clock64 = s_memtime ticks at a constant 100 MHz on gfx9?  report both ticks and wall time; FLOPs from wall time
  const int IT = 4000;
  run("16x16x4  9 acc, 1 wave/SIMD", k16<9, IT>, 256, 256);
  run("16x16x4  9 acc, 2 waves/SIMD", k16<9, IT>, 512, 256);
  run("16x16x4 24 acc, 2 waves/SIMD", k16<24, IT>, 512, 256);
  run("16x16x4  2 acc, 1 wave/SIMD", k16<2, IT>, 256, 256);
  run("16x16x4  1 acc, 1 wave/SIMD", k16<1, IT>, 256, 256);
  run("32x32x2  4 acc, 1 wave/SIMD", k32<4, IT>, 256, 256);
  run("32x32x2  4 acc, 2 waves/SIMD", k32<4, IT>, 512, 256);
  run("32x32x2  1 acc, 1 wave/SIMD", k32<1, IT>, 256, 256);
  // wall-time derived rates
  return 0;
}
