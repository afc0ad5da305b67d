// Probe of ds_read_b64_tr_b16 on gfx950: which element does lane l / register element j receive, as a function of the
// addresses the lanes supply?  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe ; run on the GPU box.
// Hypothesis (guide: "lane l, elem j reads lds[(l&15) + j*16 + (l>>4)*64]" for lane-linear addresses base + 8*l):
//   within each 16-lane group, result element j of lane i = 16-bit element (i % 4) of the 64-bit word whose address was supplied by
//   lane (4*j + i/4) of the same group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* addr_bytes, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const uint32_t a = (uint32_t)(uintptr_t)(&lds[0]) + addr_bytes[lane];
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

int main() {
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
  for (int test = 0; test < 3; ++test) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, w = l & 15;
      if (test == 0) addr[l] = 8 * l;                                        // lane-linear
      else if (test == 1) addr[l] = ((g * 4 + (w >> 2)) * 64 + (w & 3) * 4) * 2;   // rows of 64 elements: row g*4 + w/4, cols 4*(w%4)
      else addr[l] = ((g * 4 + (w >> 2)) * 128 + 32 + (w & 3) * 4 + 16 * (g & 1)) * 2;   // rows of 128, column offset
    }
    hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    std::vector<uint16_t> out(256);
    hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15;
        const int src_lane = g * 16 + 4 * j + i / 4;
        const int expect = addr[src_lane] / 2 + (i % 4);
        if (out[l * 4 + j] != expect) ++bad;
      }
    printf("test %d: %d mismatches vs hypothesis\n", test, bad);
    if (bad) {
      for (int l = 0; l < 64; l += 1) printf("lane %2d: %4d %4d %4d %4d\n", l, out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]);
    }
  }
  return 0;
}
