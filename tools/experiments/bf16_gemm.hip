// EXPERIMENT (not part of libmmfn_hip.so): C[M,N] = A[M,K] . B[N,K]^T with fp32 operands in HBM converted to bf16 while
// they are staged into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate / output.  Measures what the bf16-operand mode planned
// in DESIGN.md section 8 can reach on the transformer GEMM shapes with the current 128x128 / 4-wave tiling.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;

__device__ __forceinline__ int slot_of(int row, int slot) { return slot ^ ((row >> 2) & 3); }

__global__ __launch_bounds__(NT) void gemm_bf16_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[2][(BM + BN) * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int nkt = K / BK;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 ra[4], rb[4];
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + i * NT, row = u >> 3, q = u & 7;
      ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)min(m0 + row, M - 1) * K + kt * BK + q * 4);
      rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)min(n0 + row, N - 1) * K + kt * BK + q * 4);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + i * NT, row = u >> 3, q = u & 7;
      const int off = row * BK + slot_of(row, q >> 1) * 8 + (q & 1) * 4;
      bf16x4 a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = (__bf16)ra[i][e]; b[e] = (__bf16)rb[i][e]; }
      *reinterpret_cast<bf16x4*>(&sm[buf][off]) = a;
      *reinterpret_cast<bf16x4*>(&sm[buf][BM * BK + off]) = b;
    }
  };
  load(0);
  store(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) load(kt + 1);
    const __bf16* As = sm[cur];
    const __bf16* Bs = sm[cur] + BM * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 64 + i * 32 + l31, rb_ = wn * 64 + i * 32 + l31;
        a[i] = *reinterpret_cast<const bf16x8*>(&As[ra_ * BK + slot_of(ra_, 2 * s + h) * 8]);
        b[i] = *reinterpret_cast<const bf16x8*>(&Bs[rb_ * BK + slot_of(rb_, 2 * s + h) * 8]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

extern "C" int exp_gemm_bf16_nt(const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
  if (K % BK) return 1;
  const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
  hipLaunchKernelGGL(gemm_bf16_nt, dim3(tm * tn), dim3(NT), 0, (hipStream_t)stream, A, B, C, M, N, K, tn);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
