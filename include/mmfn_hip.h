/*
 * mmfn_hip.h — C ABI of libmmfn_hip.so: the MI355X (gfx950) kernels behind the MMFN
 * training hot path.
 *
 * Boundary: the reference (Kin-Zhang/mmfn) has no native layer; the
 * arithmetic it dispatches lives in aten/cuDNN behind nn.Module.forward and autograd.  These
 * entry points replace those dispatches one-for-one.  Every launcher:
 *   - takes raw device pointers, sizes and the HIP stream (void*, a hipStream_t),
 *   - allocates nothing, never synchronises, is safe to capture into a hipGraph,
 *   - returns 0 on success, a hipError_t (>0) or a negative MMFN_E* code on failure.
 * All activations are fp32, channels-last (NHWC feature maps, [rows, C] token matrices).
 *
 * Reference interface replaced is cited per entry (paths relative to /root/reference).
 */
#ifndef MMFN_HIP_H
#define MMFN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMFN_EINVAL (-1) /* bad argument / unsupported shape */

/* ---- library info -------------------------------------------------------------------- */
int mmfn_abi_version(void);
int mmfn_sizeof_gemm_desc(void);
/* launches an empty kernel: smoke test that the code object loads on this GPU */
int mmfn_device_selftest(void* stream);

/* ---- utility ------------------------------------------------------------------------- */
int mmfn_fill_f32(float* p, float v, int64_t n, void* stream);
/* y = a*x + b*y (b == 0 ignores the old y) */
int mmfn_axpby_f32(float* y, const float* x, float a, float b, int64_t n, void* stream);
/* dropout RNG state {seed, step}: step += 1 (launched once per training step, graph-safe) */
int mmfn_rng_advance(uint64_t* state, void* stream);

/* ---- GEMM / implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 ----------------------- */
/* operand addressing modes */
enum {
  MMFN_A_ROWMAJOR = 0, /* A[m*lda + k]                           (Linear fwd / dX)        */
  MMFN_A_COLMAJOR = 1, /* A[k*lda + m]                           (dW: A = dY^T)           */
  MMFN_A_IM2COL = 2,   /* A = im2col(x NHWC), m=(b,oh,ow) k=(kh,kw,ci)   (conv fwd)       */
  MMFN_A_DGRAD = 3     /* A = transposed-conv gather of dY, m=(b,ih,iw) k=(kh,kw,co)      */
};
enum {
  MMFN_B_NK = 0,      /* B[n*ldb + k]  (weights [out,in])                                 */
  MMFN_B_KN = 1,      /* B[k*ldb + n]                                                     */
  MMFN_B_IM2COL = 2,  /* B = im2col(x) with k=(b,oh,ow), n=(kh,kw,ci)    (conv wgrad)     */
  MMFN_B_DGRADW = 3   /* B = W[co][kh][kw][ci] read as k=(kh,kw,co), n=ci (conv dgrad)    */
};
/* epilogue flags */
enum {
  MMFN_EPI_BIAS = 1,       /* + bias[n]                                                   */
  MMFN_EPI_RELU = 2,       /* max(v, 0)                                                   */
  MMFN_EPI_GELU = 4,       /* exact erf GELU                                              */
  MMFN_EPI_MASK_AUX = 8,   /* v = aux[m,n] > 0 ? v : 0   (ReLU backward)                  */
  MMFN_EPI_DROPOUT = 16,   /* v = keep ? v/(1-p) : 0, counter-based RNG                   */
  MMFN_EPI_RESIDUAL = 32,  /* + res[m*ldr + n]                                            */
  MMFN_EPI_ACCUM = 64      /* + C[m,n] (beta = 1)                                         */
};

typedef struct mmfn_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  const float* aux;
  const uint64_t* rng_state; /* device: [0]=seed, [1]=step counter (MMFN_EPI_DROPOUT)     */
  float* workspace;          /* split-K slabs: splitk*M*N floats (may be NULL if splitk<=1) */
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldr, ldaux;
  int32_t a_mode, b_mode;
  /* conv geometry (IM2COL / DGRAD modes): input H,W,Cin; output OH,OW,Cout; kernel */
  int32_t H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int32_t flags;
  int32_t splitk;       /* 0 auto, 1 none, >1 forced number of k slices                   */
  int32_t tile;         /* 0 auto, 1 = 128x128, 2 = 64x64                                 */
  uint32_t rng_stream;  /* distinguishes dropout sites                                    */
  float drop_p;
} mmfn_gemm_desc;

/* C = epilogue(A*B).  Replaces aten addmm / cudnn convolution fwd, dgrad, wgrad dispatched by
 * nn.Linear (model_vec.py:82-89,121-123,...) and torchvision ResNet convs (model_vec.py:509-575). */
int mmfn_gemm_f32(const mmfn_gemm_desc* d, void* stream);
/* bytes of split-K workspace mmfn_gemm_f32 needs for this descriptor (0 if none) */
int64_t mmfn_gemm_workspace_bytes(const mmfn_gemm_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* MMFN_HIP_H */
