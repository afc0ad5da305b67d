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

// Dropout of the attention probabilities (attn_pdrop) in attention_wg.hip and attention16.hip: the keep mask of element (b, h, query,
// key) is a 32-bit hash (lowbias32) of query * T + key, salted per (rng state, stream, b, h) - the same function in the three
// kernels of a family, with the salt, the threshold compare on the raw 32 bits and 1 / (1 - p) hoisted into a struct (the same
// mixer as common.h's mmfn_rng_u32, which every other dropout site and the tile kernels of attention.hip use per flat index).
#ifdef __HIPCC__
struct Drop {
  uint32_t salt, thresh;
  float inv_keep;
  bool on;
  __device__ __forceinline__ void init(const AttnArgs& a, int b, int hd) { init(a.drop_p, a.rng_state, a.rng_stream, a.NH, b, hd); }
  __device__ __forceinline__ void init(float drop_p, const uint64_t* rng_state, uint32_t rng_stream, int NH, int b, int hd) {
    on = drop_p > 0.f;
    salt = 0; thresh = 0; inv_keep = 1.f;
    if (on) {
      const uint64_t k = mmfn_rng_key(rng_state, rng_stream) + (uint64_t)(b * NH + hd) * 0x9E3779B97F4A7C15ull;
      salt = mmfn_rng_salt(k);
      thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967040.0f);
      inv_keep = 1.0f / (1.0f - drop_p);
    }
  }
  // keep-scale of element (query, key): 0 or 1 / (1 - p)
  __device__ __forceinline__ float scale(int query, int key, int T) const {
    return mmfn_hash32((uint32_t)(query * T + key) ^ salt) >= thresh ? inv_keep : 0.f;
  }
};
#endif

// attention_wg.hip.  which: 0 forward, 1 backward dQ (+delta), 2 backward dK/dV.  Returns -1 when the shape is not
// covered by the workgroup-per-half kernels (the caller then uses the tile kernels of attention.hip).
int mmfn_attn_wg_launch(int which, int hs, const AttnArgs& a, hipStream_t s);
// attention16.hip: the bf16 mode's kernels (bf16 I/O AND bf16 MFMA, fp32 softmax / accumulation); T = 64 / 128 / 192.  -1 = not covered.
int mmfn_attn16_launch(int which, int hs, const AttnArgs& a, hipStream_t s);
