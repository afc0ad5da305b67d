// Argument block shared by the attention kernels (attention.hip: one 32-query tile per block, any T <= 256;
// attention_wg.hip: one (sample, head, half) per workgroup for T a multiple of 64).
#pragma once
#include "common.h"

struct AttnArgs {
  const float* q; const float* k; const float* v;  // row stride ld, head h at column h*HS
  float* o;                                         // [B*T, ldo]
  float* lse;                                       // [B, NH, T]
  const float* dO;                                  // backward: grad of o (ldo)
  float* delta;                                     // [B, NH, T]  sum_j P_j dP_j
  float* dq; float* dk; float* dv;                  // row stride ldg
  const int* kv_len;                                // optional [B]: keys >= kv_len[b] are masked
  const uint64_t* rng_state;
  int B, T, NH, ld, ldo, ldg;
  float scale, drop_p;
  uint32_t rng_stream;
  long long* dbg;                                   // MMFN_ATTN_DEBUG=1 (dev only): phase time stamps of workgroup 0
  int io_bf16;                                      // bf16 mode: q k v o dO dq dk dv point at bf16 (row strides in elements); lse / delta stay fp32
};

// attention_wg.hip.  which: 0 forward, 1 backward dQ (+delta), 2 backward dK/dV.  Returns -1 when the shape is not
// covered by the workgroup-per-half kernels (the caller then uses the tile kernels of attention.hip).
int mmfn_attn_wg_launch(int which, int hs, const AttnArgs& a, hipStream_t s);
// attention16.hip: the bf16 mode's kernels (bf16 I/O AND bf16 MFMA, fp32 softmax / accumulation); T = 64 / 128 / 192.  -1 = not covered.
int mmfn_attn16_launch(int which, int hs, const AttnArgs& a, hipStream_t s);
