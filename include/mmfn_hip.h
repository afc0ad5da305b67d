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
/* filter of the transposed convolution: w[Co][T][Ci] -> wt[Ci][T][Co], taps reversed.  With it the stride-1 data
 * gradient (aten convolution_backward input grad under every torchvision BasicBlock conv, model_vec.py:509-593) is
 * the forward implicit GEMM of dY with k-contiguous weights. */
int mmfn_conv_weight_flip_f32(const float* w, float* wt, int Co, int T, int Ci, void* stream);

/* ---- Winograd F(m x m, 3x3) transforms, m = 2 or 4, for the 3x3 stride-1 convolutions of ResNet layer3/4 (cuDNN's choice
 * under the same aten::convolution calls, model_vec.py:509-593).  n = (m+2)^2 element-wise products:
 *   conv = output_tf( batched GEMM over t < n of V[t] . U[t]^T ),
 *   U[n][Co][Ci] = G w G^T,  V[n][tiles][C] = B^T x B  (tiles = B*(H/m)*(W/m), zero padding 1),
 *   y = A^T Mt A (+ res, NHWC like y).  H and W multiples of m, C % 4 == 0. */
int mmfn_wino_weight_f32(const float* w, float* U, int Co, int Ci, int m, void* stream);
/* F(4x4,3x3) filter transforms of many layers in one launch.  table: DEVICE array of n_layers records
 * { const float* w; float* U; int32 Co; int32 Ci; int64 start } (32 bytes each, start = running sum of Co*Ci, ascending);
 * total = sum of Co*Ci.  U[l] receives [36][Co][Ci] exactly as mmfn_wino_weight_f32(m = 4) writes it. */
int mmfn_wino_weight_group_f32(const void* table, int n_layers, int64_t total, void* stream);
int mmfn_wino_input_f32(const float* x, float* V, int B, int H, int W, int C, int m, void* stream);
/* F(4x4) input transform of d = [relu]( (x - mean) * rstd * weight + bias [+ res] ): the PRODUCER's BatchNorm apply
 * (aten batch_norm's elementwise pass + add_ + relu_ between two convolutions of a torchvision BasicBlock chain,
 * model_vec.py:509-593) runs inside the consumer's transform; x = the producer's convolution output, the zero padding pads d.
 * y != NULL also receives d as an NHWC tensor (a block output that the next block's skip connection reads); y == NULL: d is
 * never written, and the backward kernels recompute its ReLU sign from x (relu_bias arguments below). */
int mmfn_wino_input_bn_f32(const float* x, const float* res, const float* mean, const float* rstd, const float* weight,
                           const float* bias, int relu, float* y, float* V, int B, int H, int W, int C, void* stream);
int mmfn_wino_output_f32(const float* Mt, const float* res, float* y, int B, int H, int W, int C, int m, void* stream);
/* F(4x4) output transform that also writes the BatchNorm batch-statistics partial rows of y ([*nblk_out][2][C] doubles, at most
 * 512 rows; *nblk_out is a host int), to be finished by mmfn_bn_finalize_stats_f32 */
int mmfn_wino_output_stats_f32(const float* Mt, float* y, double* partials, int* nblk_out, int B, int H, int W, int C,
                               void* stream);
/* weight gradient in the F(4x4,3x3) domain: dw = G^T [ sum_tiles (A dY A^T) . (B^T x B) ] G
 *   dMt[36][tiles][Co] = A dy A^T per 4x4 patch;  dU[t] = dMt[t]^T . V[t] (batched GEMM);  dw[Co][3][3][Ci] = G^T dU G */
int mmfn_wino_outgrad_f32(const float* dy, float* dMt, int B, int H, int W, int C, void* stream);
/* The same transform of dy = BatchNorm-backward(g, y, x) formed on the fly (mmfn_bn_bwd_f32's apply pass fused in): g = dL/d(BN
 * output), y != NULL applies the ReLU mask (y > 0); y == NULL with relu_bias != NULL (the BatchNorm's bias) applies the mask
 * recomputed from x (the output was consumed by mmfn_wino_input_bn_f32 and never written); x = the convolution output,
 * means[2][C] from mmfn_bn_bwd_reduce_f32; ge_out (optional) = masked g.  dy itself is never written: these layers only consume
 * it in the Winograd domain. */
int mmfn_wino_outgrad_bn_f32(const float* g, const float* y, const float* x, const float* mean, const float* rstd,
                             const float* weight, const float* relu_bias, const float* means, float* ge_out, float* dMt, int B,
                             int H, int W, int C, void* stream);
/* Data gradient of the same convolution as the ADJOINT of its forward Winograd pipeline: dV [36][tiles][Ci] (= dM . U, one
 * 36-batch GEMM over the forward's own transformed filter) -> dx = overlap-add of B dV B^T over the tiles' 6x6 input patches
 * (+ res).  H, W multiples of 4; replaces cuDNN's backward-data for the BasicBlock 3x3 convolutions (model_vec.py:539-593). */
int mmfn_wino_input_adjoint_f32(const float* dV, const float* res, float* dx, int B, int H, int W, int C, void* stream);
int mmfn_wino_wgrad_out_f32(const float* dU, float* dw, int Co, int Ci, void* stream);
/* The same transform over the un-combined partial products of a split-K batched GEMM (MMFN_EPI_KEEP_SLABS): slabs
 * [splits][36][Co][Ci], summed in slice order while they are read - the split-K combine launch of every Winograd weight gradient
 * (65 per training step) disappears into this one.  Ci a multiple of 64. */
int mmfn_wino_wgrad_out_slabs_f32(const float* slabs, int splits, float* dw, int Co, int Ci, void* stream);
/* The 7x7 stride-2 stems (torchvision conv1, model_vec.py:509,515: 3 camera / 2 BEV channels) as explicit im2col + plain GEMM:
 * col[B*OH*OW][KP] (fp32, or bf16 with out_bf16) = the zero-padded patch matrix of x [B,H,W,Cin] (fp32, Cin <= 4), k = (kh, kw, ci),
 * columns K = KH*KW*Cin .. KP zero.  Forward = col . w_padded^T, weight gradient = dY^T . col (the same matrix, kept). */
int mmfn_im2col_small(const float* x, void* col, int out_bf16, int B, int H, int W, int Cin, int KH, int KW, int stride, int pad,
                      int KP, void* stream);
/* rows of K fp32 values from pitch ps to pitch pd >= K (columns K..pd zero-filled; dst fp32, or bf16 with dst_bf16): the stem
 * filter [Cout][K] -> [Cout][KP] for the GEMM above, and its padded gradient back to [Cout][K] */
int mmfn_repitch_rows(const float* src, void* dst, int dst_bf16, int R, int K, int ps, int pd, void* stream);
/* y = a*x + b*y (b == 0 ignores the old y) */
int mmfn_axpby_f32(float* y, const float* x, float a, float b, int64_t n, void* stream);
/* out = y > 0 ? g : 0  (ReLU backward) */
int mmfn_relu_mask_f32(const float* g, const float* y, float* out, int64_t n, void* stream);
/* dropout RNG state {seed, step}: step += 1 (launched once per training step, graph-safe) */
int mmfn_rng_advance(uint64_t* state, void* stream);
/* out[i] = in[i] * (keep_i ? 1/(1-p) : 0) with the mask of a contiguous [M,N] epilogue dropout
 * (index i = row*N + col): backward of nn.Dropout fused into a GEMM epilogue */
int mmfn_dropout_apply_f32(const float* in, float* out, int64_t n, float p, const uint64_t* rng_state,
                           uint32_t rng_stream, void* stream);

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
  MMFN_EPI_ACCUM = 64,     /* + C[m,n] (beta = 1)                                         */
  MMFN_EPI_BF16_OPERANDS = 128, /* opt-in mixed precision for plain GEMM forms: A and B rounded to bf16 on the way into LDS,
                                  bf16 MFMA, fp32 accumulate / epilogue / output (autocast-style; BASELINE configs[2]) */
  MMFN_EPI_BF16X3 = 256    /* fp32 arithmetic on the bf16 MFMA pipe (plain GEMM forms): each operand element split exactly
                              into three bf16 terms, six cross products accumulated in fp32: product error < 2^-22 relative */,
  MMFN_EPI_LN_FOLD = 4096, /* see mmfn_gemm_desc.ln_c1 */
  MMFN_EPI_COLSUM_A = 8192, /* see mmfn_gemm_desc.colsum */
  MMFN_EPI_KEEP_SLABS = 16384, /* a launch that splits K (mmfn_gemm_f32_splits(d) > 1) leaves its partial products in the workspace,
                              [split][batch][M][N], and runs NO combine kernel: the consumer sums them in slice order as it reads
                              (mmfn_wino_wgrad_out_slabs_f32).  C is then not written.  No other epilogue flag may be set. */
  MMFN_EPI_RELU_LAST = 512 /* max(v, 0) as the LAST step, after residual / accumulate: conv + folded BatchNorm + skip + ReLU in one
                              launch (eval mode, mmfn_bn_fold_f32) */
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
  int32_t tile;         /* 0 auto, 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128 (2x2 waves of 32x32 MFMA tiles); on request only:
                         * 5 = 192x64, 6 = 64x192 (wave tile 96x32 / 32x96), 7 = 64x64 as two waves of 32x64 */
  uint32_t rng_stream;  /* distinguishes dropout sites                                    */
  float drop_p;
  /* batched GEMM (radar GAT, model_rad.py:816-824): problem z uses A + z*strideA, ... (floats);
   * batch <= 1 means a single problem.  Batched launches split K when the epilogue has no per-batch operand
   * (no residual / mask / accumulate; dropout only with packed outputs, strideC == M*N). */
  int32_t batch;
  int32_t dg_parity; /* internal: set by the launcher for stride-2 dgrad (output-parity decomposition) */
  int64_t strideA, strideB, strideC;
  /* MMFN_EPI_LN_FOLD (NT form, K = the LayerNorm width): C = LN(A) . B^T + b computed as
   *   rstd_m * (A . B'^T - mean_m * c1_n) + c2_n,   B' = B . diag(gamma), c1_n = sum_k B'_nk, c2_n = sum_k beta_k B_nk + b_n
   * (mmfn_ln_fold_weights_f32 derives B', c1, c2 once per step): the GEMM reads the raw rows through the unchanged
   * global -> LDS path, accumulates each row's sum and sum of squares from the A fragments it feeds the MFMAs anyway, and applies
   * the normalisation in the epilogue - native_layer_norm + addmm of model_vec.py:117-118,82-98 (ln1 -> key/query/value) and
   * :119,121 (ln2 -> mlp.0) in one launch.  B = B', bias = c2, ln_c1 = c1; ln_mean / ln_rstd (optional, [M]) receive the row
   * statistics the LayerNorm backward needs.  No split-K, M and N multiples of the tile. */
  const float* ln_c1;
  float* ln_mean;
  float* ln_rstd;
  float ln_eps;
  int32_t reserved0;
  /* MMFN_EPI_COLSUM_A (TN form A_COLMAJOR x B_KN, single problem, 16-byte aligned operands, K a multiple of 16): colsum[m] =
   * sum_k A[k, m], the column sums of the A operand, from the fragments the kernel feeds its MFMAs anyway.  For a Linear's weight
   * gradient dW = dY^T X that is the BIAS gradient (aten sum(dY, 0) beside addmm's backward, model_vec.py:82-89,121-123) without
   * the two column-sum launches.  With split-K the per-slice sums go to the tail of the workspace (mmfn_gemm_workspace_bytes
   * accounts for it) and the combine launch adds them in slice order. */
  float* colsum;
} mmfn_gemm_desc;

/* ---- bf16 training mode (BASELINE configs[2]): GEMM / implicit-GEMM convolution with bf16 operands in HBM ------------- */
/* forms of mmfn_gemm_bf16 */
enum {
  MMFN_G16_NT = 0,        /* C[M,N] = A[M,K] . B[N,K]^T, both k-contiguous (Linear forward; Linear dX over the W^T shadow) */
  MMFN_G16_CONV_FWD = 1,  /* A = implicit im2col of x [B,H,W,Cin] (m = output pixel, k = (kh,kw,ci)), B = w [Cout][KH][KW][Cin] */
  MMFN_G16_CONV_DGRAD = 2,/* A = transposed-conv gather of dY [B,OH,OW,Cout] (m = input pixel, k = (kh,kw,co)), any stride,
                             B = the [Cin][KH][KW][Cout] weight shadow (mmfn_shadow_transpose_bf16); C = dx [B,H,W,Cin] */
  MMFN_G16_TN = 3,        /* C[M,N] = sum_k A[k,M] . B[k,N] (Linear dW = dY^T X), fp32 output */
  MMFN_G16_CONV_WGRAD = 4 /* A = dY [pixels][Cout], B = implicit im2col of x with k = output pixel, n = (kh,kw,ci): dw [Cout][KH][KW][Cin] fp32 */
};
#define MMFN_EPI16_OUT_F32 1024 /* C (and the MMFN_EPI_ACCUM read of it) is fp32 instead of bf16; res / aux stay bf16 */
#define MMFN_EPI16_RES_F32 2048 /* the residual operand is fp32 (the transformers' residual stream stays fp32 in the bf16 mode, as under
                                 * torch.autocast: x + Linear(.) with x fp32) */

typedef struct mmfn_gemm16_desc {
  const void* A;   /* bf16 */
  const void* B;   /* bf16 */
  void* C;         /* bf16, or fp32 with MMFN_EPI16_OUT_F32 (always for the TN forms) */
  const float* bias;
  const void* res; /* bf16 [M, ldr] */
  const void* aux; /* bf16 [M, ldaux] (MMFN_EPI_MASK_AUX) */
  const uint64_t* rng_state;
  float* workspace; /* TN forms: split slabs, mmfn_gemm_bf16_workspace_bytes() */
  double* stats;    /* NT forms, optional: per-column partial sums of the output tile, [2 * ceil(M / tile rows)][2][N] doubles, by
                       stats_mode: 0 = (sum, sum of squares) of the RAW accumulators - BatchNorm batch statistics of a convolution
                       output, finished by mmfn_bn_finalize_stats_f32;  1 = (sum, sum of squares) of the FINAL value after the whole
                       epilogue - column sums = the bias gradient when the output is a Linear's input gradient, finished by
                       mmfn_colsum_partials_f64;  2 = (sum ge, sum ge * xhat) with ge = final value masked by bn_y > 0 (bn_y may be
                       NULL) and xhat = (bn_x - bn_mean) * bn_rstd: the two reductions of the BatchNorm backward whose output
                       gradient this launch produces (a data gradient feeding the BatchNorm of the layer below), finished by
                       mmfn_bn_bwd_bf16(..., partials) */
  const void* bn_y;       /* stats_mode 2: bf16 [M, N] (ldc), the BatchNorm output whose sign gates the ReLU, or NULL */
  const void* bn_x;       /* stats_mode 2: bf16 [M, N] (ldc), the BatchNorm input (the convolution output) */
  const float* bn_mean;   /* stats_mode 2: [N] */
  const float* bn_rstd;   /* stats_mode 2: [N] */
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldr, ldaux; /* in elements */
  int32_t form;
  int32_t H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int32_t flags;       /* MMFN_EPI_* (BIAS RELU GELU MASK_AUX DROPOUT RESIDUAL ACCUM RELU_LAST) | MMFN_EPI16_OUT_F32 | MMFN_EPI16_RES_F32 */
  int32_t splitk;      /* TN forms: 0 auto, >= 1 forced number of contraction slices */
  int32_t tile;        /* 0 auto, 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128 */
  uint32_t rng_stream;
  float drop_p;
  int32_t stages;      /* LDS stages of the operand pipeline: 0 auto, 2 = double buffer, 3 / 4 = 1 / 2 k-tiles in flight beyond it */
  int32_t stats_mode;  /* see `stats` */
  int32_t reserved;
} mmfn_gemm16_desc;

int mmfn_sizeof_gemm16_desc(void);
int64_t mmfn_gemm_bf16_workspace_bytes(const mmfn_gemm16_desc* d);
/* rows of d->stats a launch writes: 2 * ceil(M / tile rows) */
int mmfn_gemm_bf16_stats_rows(const mmfn_gemm16_desc* d);
/* Replaces, in the bf16 mode, the same aten addmm / cuDNN convolution forward / backward-data / backward-filter dispatches as
 * mmfn_gemm_f32 (model_vec.py:82-89,121-123 Linear; :509-593 ResNet convolutions) with torch.autocast(bfloat16)-style arithmetic:
 * bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16), bf16 activations out, fp32 weight gradients out.
 * Requirements: K % 64 == 0 and N % 8 == 0 (NT forms; conv: channel counts multiples of 64), leading dimensions multiples
 * of 8, 16-byte aligned pointers; conv wgrad needs OW and OH*OW powers of two. */
int mmfn_gemm_bf16(const mmfn_gemm16_desc* d, void* stream);

/* ---- bf16 training mode: 3x3 stride-1 pad-1 convolution with the halo patch of a pixel tile resident in LDS and the
 * PRODUCER's elementwise pass applied while the patch is staged (csrc/conv16_halo.hip).
 * Replaces, per torchvision BasicBlock convolution (model_vec.py:509-521,539-593 - resnet.py BasicBlock.forward: conv -> bn ->
 * relu -> conv -> bn -> += identity -> relu), TWO aten dispatches of the reference's autocast step with one launch:
 *   pro 1, flip 0: native_batch_norm's apply (+ add + relu) of the layer below  +  cudnn convolution forward;
 *   pro 2, flip 1: native_batch_norm_backward's elementwise pass              +  cudnn convolution backward-data;
 *   pro 0:         the convolution alone (input already an activation).
 * Operands bf16 [B, H, W, K] (contraction channels K = 64 .. 512, multiple of 64) and w bf16 [N][3][3][K] (forward: the filter
 * [Cout][3][3][Cin]; data gradient: the [Cin][3][3][Cout] shadow of mmfn_shadow_transpose_bf16, taps reversed by `flip`);
 * out bf16 [B, H, W, N], N a multiple of 64; H, W powers of two >= 8 (W) / 4 (H).  The k order is the implicit GEMM's, so on
 * equal inputs `out` equals mmfn_gemm_bf16's MMFN_G16_CONV_FWD / _DGRAD bit for bit. */
typedef struct mmfn_conv16_halo_desc {
  const void* x;        /* pro 0: the activation; pro 1: the producer's convolution output co; pro 2: g = dL/dy of the BatchNorm */
  const void* w;
  void* out;
  double* stats;        /* optional partial sums [mmfn_conv3x3_halo_bf16_stats_rows()][2][N], by stats_mode (0 / 2 of mmfn_gemm16_desc) */
  const void* out_res;  /* optional bf16 [B, H, W, N] added to the output (a skip connection's gradient joining the data gradient) */
  const void* bn2_y;    /* stats_mode 2: as mmfn_gemm16_desc.bn_y / bn_x / bn_mean / bn_rstd */
  const void* bn2_x;
  const float* bn2_mean;
  const float* bn2_rstd;
  const float* p_mean;  /* prologue: [K] batch mean / rstd / BatchNorm weight / bias of the producer */
  const float* p_rstd;
  const float* p_weight;
  const float* p_bias;  /* pro 1 */
  const float* p_means; /* pro 2: [2][K] = mean(ge), mean(ge * xhat) (mmfn_bn_bwd_finalize_f64) */
  const void* p_res;    /* pro 1: optional residual bf16 [B, H, W, K] added before the ReLU */
  const void* p_y;      /* pro 2: optional BatchNorm output whose sign gates g (ReLU) */
  const void* p_x;      /* pro 2: the BatchNorm input (the producer's convolution output) */
  void* a_out;          /* optional bf16 [B, H, W, K]: the applied tensor (pro 1: y; pro 2: dco), written once by the tile that owns a pixel */
  void* ge_out;         /* pro 2, optional: the ReLU-masked g */
  int32_t B, H, W, K, N;
  int32_t pro;          /* 0 / 1 / 2 */
  int32_t relu;         /* pro 1 */
  int32_t flip;         /* 1: tap (kh, kw) multiplies the pixel at (1 - kh, 1 - kw) (data gradient) instead of (kh - 1, kw - 1) */
  int32_t tile;         /* 0 auto, 1 = 128 pixels x 64 channels, 2 = 128 x 128, 3 = 64 x 64, 4 = 64 x 128 */
  int32_t stages;       /* filter ring depth: 0 auto, 2 .. 4 */
  int32_t stats_mode;
} mmfn_conv16_halo_desc;
int mmfn_sizeof_conv16_halo_desc(void);
/* the tile a launch of this descriptor would use, 0 if the shape is not served (the caller then runs apply + mmfn_gemm_bf16) */
int mmfn_conv3x3_halo_bf16_ok(const mmfn_conv16_halo_desc* d);
int mmfn_conv3x3_halo_bf16_stats_rows(const mmfn_conv16_halo_desc* d);
int mmfn_conv3x3_halo_bf16(const mmfn_conv16_halo_desc* d, void* stream);

/* C = epilogue(A*B).  Replaces aten addmm / cudnn convolution fwd, dgrad, wgrad dispatched by
 * nn.Linear (model_vec.py:82-89,121-123,...) and torchvision ResNet convs (model_vec.py:509-575). */
int mmfn_gemm_f32(const mmfn_gemm_desc* d, void* stream);
/* bytes of split-K workspace mmfn_gemm_f32 needs for this descriptor (0 if none) */
int64_t mmfn_gemm_workspace_bytes(const mmfn_gemm_desc* d);
/* contraction slices the launch of this descriptor will use (1: the result goes to C); d->workspace must be what the launch gets */
int mmfn_gemm_f32_splits(const mmfn_gemm_desc* d);

/* ---- normalisation ------------------------------------------------------------------- */
/* scratch for the norm kernels below (bytes); one buffer of this size for the largest C suffices */
int64_t mmfn_norm_workspace_bytes(int C);
/* BatchNorm2d training statistics over x[M,C] (NHWC rows): mean/rstd (biased var, eps) for the
 * normalisation, running stats updated with momentum and the UNBIASED variance, and
 * num_batches_tracked += 1.  Replaces aten native_batch_norm(training=True) under every
 * torchvision BasicBlock / stem bn1 (model_vec.py:510,516,520-521,539-541,...). */
int mmfn_bn_train_stats_f32(const float* x, int64_t M, int C, float eps, float momentum, float* mean, float* rstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, void* workspace,
                            void* stream);
/* eval mode: mean = running_mean, rstd = 1/sqrt(running_var + eps) */
int mmfn_bn_finalize_stats_f32(const double* partials, int nblk, int64_t M, int C, float eps, float momentum, float* mean,
                               float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream);
int mmfn_bn_eval_prepare_f32(const float* running_mean, const float* running_var, float eps, int C, float* mean,
                             float* rstd, void* stream);
/* y = [relu]( (x - mean) * rstd * weight + bias [+ res] ) */
/* Eval-mode BatchNorm folded into the preceding convolution: w_out[co][k] = w[co][k] * s[co], b_out[co] = beta[co] - mean[co] * s[co],
 * s = gamma * rsqrt(var + eps) over the RUNNING statistics (model_vec.py:509-593 in eval mode); the convolution then runs with
 * bias = b_out (+ residual, MMFN_EPI_RELU / MMFN_EPI_RELU_LAST).  K = KH*KW*Cin elements per output channel. */
int mmfn_bn_fold_f32(const float* w, int Cout, int K, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* w_out, float* b_out, void* stream);
int mmfn_bn_apply_f32(const float* x, const float* res, float* y, int64_t M, int C, const float* mean, const float* rstd,
                      const float* weight, const float* bias, int relu, void* stream);
/* backward of y = relu?(bn(x) [+res]): g = dL/dy; if y != NULL the ReLU mask (y > 0) is applied first; y == NULL with
 * relu_bias != NULL (the BatchNorm's bias; no residual): the mask (bn(x) > 0) is recomputed from x - the forward never wrote y
 * (mmfn_wino_input_bn_f32).  Writes dx, dweight, dbias and (optionally) ge_out = masked g, the gradient of the residual branch. */
int mmfn_bn_bwd_f32(const float* g, const float* y, const float* x, int64_t M, int C, const float* mean, const float* rstd,
                    const float* weight, const float* relu_bias, float* dx, float* ge_out, float* dweight, float* dbias,
                    void* workspace, void* stream);
/* The reductions of mmfn_bn_bwd_f32 without its apply pass: dweight, dbias, means[2][C] = (mean(ge), mean(ge * xhat));
 * relu_weight / relu_bias (both or neither): the recomputed mask as above */
int mmfn_bn_bwd_reduce_f32(const float* g, const float* y, const float* x, int64_t M, int C, const float* mean,
                           const float* rstd, const float* relu_weight, const float* relu_bias, float* dweight, float* dbias,
                           float* means, void* workspace, void* stream);
/* Operands of MMFN_EPI_LN_FOLD (mmfn_gemm_desc.ln_c1) for every (LayerNorm -> Linear) pair of the step in one launch.  table: DEVICE
 * array of n_entries records { const float* W [N][K]; const float* gamma, *beta [K]; const float* bias [N] or NULL; float* Wf
 * [N][K]; float* c1, *c2 [N]; int32 N, K; int64 row0 } (72 bytes; row0 = running sum of N, ascending), total_rows = sum of N:
 *   Wf[n][k] = W[n][k] * gamma[k],  c1[n] = sum_k Wf[n][k],  c2[n] = sum_k beta[k] * W[n][k] + bias[n]   (sums in fp64) */
int mmfn_ln_fold_weights_f32(const void* table, int n_entries, int64_t total_rows, void* stream);
/* LayerNorm over rows of x[M,C] (C % 64 == 0, C <= 512), optional fused activation on the output
 * (act: 0 none, 1 ReLU, 2 exact GELU).  Replaces aten native_layer_norm (+relu/gelu) of
 * model_vec.py:117-118,162 (GPT) and :252,335-336,345-346,352-353 (VectorNet). */
int mmfn_layernorm_fwd_f32(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                           int M, int C, float eps, int act, void* stream);
/* dx = LN'(g * act'(.)) [+ dres]; dweight, dbias reduced over rows */
int mmfn_layernorm_bwd_f32(const float* g, const float* x, const float* weight, const float* bias, const float* mean,
                           const float* rstd, const float* dres, float* dx, float* dweight, float* dbias, int M, int C,
                           int act, void* workspace, void* stream);
/* Same, plus an optional second output dx_dropped = dx * keep_scale(row * C + col) for the dropout that the forward applied
 * in the epilogue of the following residual branch's last GEMM (counter RNG: same state / stream / index), so the backward
 * needs no separate dropout pass over dx (model_vec.py:107-108,130-131: resid_drop).  dx_dropped NULL = plain.
 * dx_colsum (optional, [C]): column sums of dx_dropped (of dx when dx_dropped is NULL) = the bias gradient of the Linear
 * whose output gradient this tensor is (attn.proj / mlp.2), produced here instead of by a separate mmfn_colsum_f32. */
int mmfn_layernorm_bwd_drop_f32(const float* g, const float* x, const float* weight, const float* bias, const float* mean,
                                const float* rstd, const float* dres, float* dx, float* dweight, float* dbias, int M, int C,
                                int act, float* dx_dropped, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                                float* dx_colsum, void* workspace, void* stream);
/* The same backward in two halves, for callers that take the row reductions off the critical path (dx is what the next
 * kernel of the chain needs; dweight / dbias / dx_colsum only feed the optimizer): _partial_ writes dx (and dx_dropped) plus
 * per-block partial rows to `partials` (mmfn_layernorm_bwd_rows(M) * (want_colsum ? 3 : 2) * C floats, caller-owned so that
 * it survives until) _finalize_, which may run later or on another stream, reduces them into dweight / dbias / dx_colsum. */
int mmfn_layernorm_bwd_rows(int M);
int mmfn_layernorm_bwd_partial_f32(const float* g, const float* x, const float* weight, const float* bias, const float* mean,
                                   const float* rstd, const float* dres, float* dx, int M, int C, int act, float* dx_dropped,
                                   float drop_p, const uint64_t* rng_state, uint32_t rng_stream, int want_colsum,
                                   float* partials, void* stream);
int mmfn_layernorm_bwd_finalize_f32(const float* partials, int rows, int C, float* dweight, float* dbias, float* dx_colsum,
                                    void* stream);
/* mmfn_layernorm_bwd_finalize_f32 for n LayerNorms of one (rows, C) in one launch: table (device memory) holds per entry the four
 * pointers partials, dweight, dbias, dx_colsum (NULL: the entry's partial rows are [rows][2][C]).  The 17 LayerNorms of a fusion
 * transformer (model_vec.py:112-133 Block.ln1 / ln2, :172 ln_f) finish their parameter gradients with one launch. */
int mmfn_layernorm_bwd_finalize_batched_f32(const void* const* table, int n, int rows, int C, void* stream);
/* out[c] = sum_r in[r*ld + c]   (bias gradients) */
int64_t mmfn_colsum_workspace_bytes(int64_t M, int C);
int mmfn_colsum_f32(const float* in, int64_t M, int C, int ld, float* out, void* workspace, void* stream);
/* `batch` such sums in one launch pair: entry z reads in + z*stride_in, writes out + z*stride_out (floats); workspace =
 * batch * mmfn_colsum_workspace_bytes(M, C).  (The same bias gradient of the eight blocks of a fusion transformer.) */
int mmfn_colsum_batched_f32(const float* in, int batch, int64_t stride_in, int64_t M, int C, int ld, float* out,
                            int64_t stride_out, void* workspace, void* stream);

/* ---- pooling / token assembly / upsampling on NHWC maps ----------------------------------- */
/* MaxPool2d(3,2,1) with first-max argmax (uint8 tap index) — torchvision stem (model_vec.py:512,518) */
int mmfn_maxpool3x3s2_fwd_f32(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C, void* stream);
int mmfn_maxpool3x3s2_bwd_f32(const float* gy, const uint8_t* idx, float* gx, int B, int H, int W, int C, void* stream);
/* tok[b, g*64+a, :] = dropout(pos_emb + AdaptiveAvgPool2d(8,8)(frame g of sample b)[a,:] + vel_emb(velocity[b]))
 * for n_modal feature maps F_m[B*frames[m],S,S,C]: model_vec.py:527-529 + GPT.forward :223-235 in one pass.  A token
 * group g is one frame: the frames[0] frames of modality 0 (n_views*seq_len camera frames per sample), then the
 * frames[1] of modality 1 (seq_len LiDAR frames), ... in the order GPT.forward concatenates them (:226-231).
 * frames == NULL: one frame per modality (seq_len = n_views = 1), group of modality m = m. */
int mmfn_tokens_fwd_f32(const float* const* feats, int n_modal, const int32_t* frames, int B, int S, int C, const float* pos, const float* vel_w,
                        const float* vel_b, const float* velocity, float* tok, float drop_p, const uint64_t* rng_state,
                        uint32_t rng_stream, void* stream);
int64_t mmfn_tokens_bwd_workspace_bytes(int T, int C);
/* in place: gtok *= dropout mask; dpos[T,C], dvel_w[C], dvel_b[C] */
int mmfn_tokens_bwd_f32(float* gtok, int B, int T, int C, const float* velocity, float* dpos, float* dvel_w, float* dvel_b,
                        float drop_p, const uint64_t* rng_state, uint32_t rng_stream, void* workspace, void* stream);
/* The next three work on one modality: B = its frames over the whole batch (samples * frames), frame i belongs to sample
 * i / frames and owns token group m + i % frames (m = the modality's first group, `frames` = its frames per sample;
 * seq_len = n_views = 1: frames = 1 and m = the modality index).
 * out = feat + bilinear_upsample(align_corners=True)(the frame's 64 tokens as 8x8xC)   (model_vec.py:531-536) */
int mmfn_upsample_add_fwd_f32(const float* feat, const float* tok, float* out, int B, int S, int C, int T, int m, int frames,
                              void* stream);
/* adjoint of the upsample: gtok[sample, group*64+a, :] = sum_pixels w(a,pixel) G[i,pixel,:] */
int mmfn_upsample_adj_f32(const float* G, float* gtok, int B, int S, int C, int T, int m, int frames, void* stream);
/* dF = G + avgpool-adjoint(the frame's 64 token gradients) */
int mmfn_pool_bcast_add_f32(const float* G, const float* gtok, float* dF, int B, int S, int C, int T, int m, int frames, void* stream);
/* out[b,c] = sum_m sum_frames mean_p feats[m][b*frames[m]+j,p,c]   (model_vec.py:585-596) and its backward; B = samples,
 * frames as in mmfn_tokens_fwd_f32 */
int mmfn_gap_sum_fwd_f32(const float* const* feats, int n, const int32_t* frames, int B, int P, int C, float* out, void* stream);
int mmfn_gap_sum_bwd_f32(const float* g, float* const* outs, int n, const int32_t* frames, int B, int P, int C, void* stream);
/* in[B,R,Cc] -> out[B,Cc,R] */
int mmfn_transpose_f32(const float* in, float* out, int B, int R, int Cc, void* stream);

/* ---- fused attention --------------------------------------------------------------------- */
/* o = softmax(q k^T * scale [keys >= kv_len[b] masked]) (dropout) v per (batch, head); q/k/v rows
 * have stride ld, head h at columns [h*HS, (h+1)*HS); T <= 384, HS in {16,32,64,128}.
 * Replaces the bmm/softmax/dropout/bmm + transposes of SelfAttention.forward (model_vec.py:96-105)
 * and MaskSelfAttention.forward (model_vec.py:308-322).  lse[B,NH,T] is saved for the backward. */
int mmfn_attention_fwd_f32(const float* q, const float* k, const float* v, int ld, float* o, int ldo, float* lse, int B,
                           int T, int NH, int HS, float scale, const int32_t* kv_len, float drop_p,
                           const uint64_t* rng_state, uint32_t rng_stream, void* stream);
int mmfn_attention_bwd_f32(const float* q, const float* k, const float* v, int ld, const float* o, const float* dO, int ldo,
                           const float* lse, float* delta, float* dq, float* dk, float* dv, int ldg, int B, int T, int NH,
                           int HS, float scale, const int32_t* kv_len, float drop_p, const uint64_t* rng_state,
                           uint32_t rng_stream, void* stream);
/* Development aid (MMFN_ATTN_DEBUG=1): copies 32 s_memtime stamps (waves 0 and 7 of workgroup 0, phase boundaries of the
 * last workgroup-form forward launch) to HOST memory; MMFN_EINVAL when the instrumentation is off. */
int mmfn_attn_debug_read(int64_t* out32);

/* ---- fused GPT block (narrow fusion transformers: n_embd 64 / 128, 4 heads, T = 192) ------------------------------
 * One transformer block of model_vec.py:112-133 in TWO forward launches and THREE backward launches (the separate kernels: 8 and
 * 9 on the dependent chain, each at the 9-18 us launch floor where the whole block holds < 2 GFLOP):
 *   mmfn_gpt_block_attn_fwd_f32   ln1 -> key / query / value of ONE head -> softmax(q k^T) (dropout) v, one workgroup per (sample,
 *                                 head, query half): replaces native_layer_norm + 3 addmm + the bmm / softmax / dropout / bmm of
 *                                 :96-105, :126 (the existing fused attention with the projections as its prologue);
 *   mmfn_gpt_block_mlp_fwd_f32    x1 = x + drop(proj(o)); a2 = ln2(x1); h = relu(mlp.0(a2)); x2 = x1 + drop(mlp.2(h)) for a block of
 *                                 32 token rows per workgroup, the hidden activations never leaving LDS between the GEMMs:
 *                                 replaces 3 addmm + native_layer_norm + relu + 2 dropout + 2 add of :107-108,:126-131;
 *   mmfn_gpt_block_bwd_rows_f32   the row-local part of the backward between two attention backward passes, 32 token rows per
 *                                 workgroup: (upper block) dqkv . Wqkv -> ln1 backward (+ residual) -> dropout mask of the block
 *                                 below; (lower block) mlp.2 dgrad (ReLU mask from h) -> mlp.0 dgrad -> ln2 backward (+ residual)
 *                                 -> dropout mask -> proj dgrad.  Either half may be absent (NULL): first / last launch of a GPT.
 * Everything a weight gradient contracts with (a, a2, o, h; gd, gh, gd2, dqkv) is still written to HBM: the weight-gradient GEMMs
 * stay separate launches off the dependent chain.  Dropout masks: the counter RNG at the indices of the unfused kernels (attention:
 * stream rng_stream, proj: rng_stream + 1, mlp.2: rng_stream + 2; element (row, col) of an [M, C] tensor = row * C + col), so fused
 * and unfused forward / backward kernels can be mixed.  LayerNorm partial rows: [n_workgroups][2 or 3][C] as mmfn_layernorm_bwd_partial
 * writes them (dweight, dbias, column sums of the dropped gradient), n_workgroups = M / 32, for mmfn_layernorm_bwd_finalize_f32. */
typedef struct mmfn_gpt_block_desc {
  /* parameters, fp32, reference layouts: Linear weights [out][in]; wqkv = key | query | value stacked [3C][C] (model_vec.py:82-84) */
  const float* ln1_w; const float* ln1_b; const float* wqkv; const float* bqkv; const float* wproj; const float* bproj;
  const float* ln2_w; const float* ln2_b; const float* w1; const float* b1; const float* w2; const float* b2;
  /* forward activations, M = B * T rows */
  const float* x;                     /* [M][C] block input (residual stream) */
  float* a; float* mu1; float* rs1;   /* ln1(x) [M][C], its row mean / 1 / std [M] */
  float* qkv;                         /* [M][3C] key | query | value */
  float* o; float* lse;               /* attention output [M][C], log-sum-exp [B][NH][T] */
  float* x1; float* a2; float* mu2; float* rs2; float* h; float* x2;   /* [M][C], [M][C], [M], [M], [M][4C], [M][C] */
  /* backward tensors */
  const float* g;       /* [M][C] gradient arriving at x2 (lower block, read when no upper block produces it in the same launch) */
  const float* gd;      /* its copy under mlp.2's dropout mask; NULL = g (resid_pdrop == 0) */
  float* gh;            /* [M][4C] gradient of the hidden activations (after the ReLU mask) */
  float* g1;            /* [M][C] gradient of x1 */
  float* gd2;           /* its copy under proj's dropout mask; NULL when resid_pdrop == 0 */
  float* go;            /* [M][C] gradient of the attention output */
  const float* dqkv;    /* [M][3C] gradient of key | query | value (upper block) */
  float* g_below;       /* [M][C] gradient of x = the block input (upper block writes it) */
  float* gd_below;      /* its copy under the mask of the block below (stream rng_stream_below + 2); NULL = not wanted */
  float* part_ln1;      /* [M/32][2 or 3][C] partial rows of ln1's backward (3 with below_colsum) */
  float* part_ln2;      /* [M/32][3][C] partial rows of ln2's backward (third: column sums of gd2 = proj's bias gradient) */
  const uint64_t* rng_state;
  int32_t B, T, C, NH;
  float attn_pdrop, resid_pdrop, eps;
  uint32_t rng_stream, rng_stream_below;
  int32_t below_colsum;  /* part_ln1 carries a third row: the column sums of what leaves in gd_below (or g_below) */
  int32_t reserved;
} mmfn_gpt_block_desc;
/* MMFN_EINVAL unless C in {64, 128}, NH == 4, T == 192: all of the block's fused launches (the attention launch is shaped for 192 tokens) */
int mmfn_gpt_block_supported(int C, int NH, int T);
/* the row-block launches alone (mlp_fwd, bwd_rows; attention and its projections as separate launches): C in {64, 128}, T % 32 == 0 -
 * the rad variant's 256 tokens, the bf16 mode */
int mmfn_gpt_block_rows_supported(int C, int T);
int mmfn_sizeof_gpt_block_desc(void);
/* Development aid (builds with -DMMFN_GPT_STAMPS): 64 s_memtime stamps of workgroup 0 (waves 0 and 7) of the last row-block launch
 * to HOST memory; MMFN_EINVAL when the instrumentation is off. */
int mmfn_gpt_debug_read(int64_t* out64);
int mmfn_gpt_block_attn_fwd_f32(const mmfn_gpt_block_desc* d, void* stream);
int mmfn_gpt_block_mlp_fwd_f32(const mmfn_gpt_block_desc* d, void* stream);
int mmfn_gpt_block_bwd_rows_f32(const mmfn_gpt_block_desc* upper, const mmfn_gpt_block_desc* lower, void* stream);
/* The bf16 training mode's row-block kernels (v_mfma_f32_16x16x32_bf16): the same descriptor with every GEMM operand bf16 -
 * o, a2, h (forward), dqkv, gd, gd_below, gh, gd2, go (backward) point at bf16 tensors of the same shapes, and the weight fields at
 * the bf16 shadows: [out][in] for the forward (wproj, w1, w2), the TRANSPOSED [in][out] shadows for the backward (wqkv, w2, w1,
 * wproj).  The residual stream and its gradient (x, x1, x2, g, g1, g_below), LayerNorm parameters / statistics / partial rows and the
 * biases stay fp32.  gd / gd2 / gd_below are required (they are where a gradient becomes a bf16 operand, with or without dropout). */
int mmfn_gpt_block_mlp_fwd_bf16(const mmfn_gpt_block_desc* d, void* stream);
int mmfn_gpt_block_bwd_rows_bf16(const mmfn_gpt_block_desc* upper, const mmfn_gpt_block_desc* lower, void* stream);

/* ---- waypoint head: GRUCell x steps + Linear(64,2) + L1 loss (model_vec.py:666-680, phase2:104) ---- */
int64_t mmfn_gru_head_part_floats(void);
int mmfn_gru_head_fwd_f32(const float* z0, const float* target, const float* w_ih, const float* w_hh, const float* b_ih,
                          const float* b_hh, const float* w_out, const float* b_out, const float* gt, float* pred,
                          float* hs, float* gates, float* xin, float* loss_terms, float* loss, int B, int steps,
                          void* stream);
int mmfn_gru_head_bwd_f32(const float* pred, const float* gt, const float* dpred, float gscale, const float* w_ih,
                          const float* w_hh, const float* w_out, const float* hs, const float* gates, const float* xin,
                          float* dz0, float* part, int B, int steps, void* stream);

/* ---- optimizer (torch.optim.AdamW semantics, phase2_train_net.py:110,256) ---------------------- */
int mmfn_step_advance(int64_t* step, void* stream);
int mmfn_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const int64_t* step, float grad_scale, void* stream);
/* torch.optim.AdamW with param_groups (the reference's decay / no-decay split, model_vec.py:179-209) and hyper-parameters
 * in DEVICE memory, so a learning-rate schedule does not invalidate a captured hipGraph.  hyper: [n_groups][8] floats
 * {lr, beta1, beta2, eps, weight_decay, grad_scale, 0, 0}; group_of: one group id per 4 consecutive parameters (tensors
 * of the flat layout are 16-byte aligned), NULL = all group 0; n must be a multiple of 4; n_groups <= 16. */
int mmfn_adamw_groups_f32(float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of, const float* hyper,
                          int n_groups, const int64_t* step, void* stream);

/* ---- sensor ingest (dataloader.py:271-308, model_vec.py:33-44,368-381) ------------------------- */
int mmfn_ingest_rgb_u8(const uint8_t* in, float* out, int B, int H, int W, int crop, void* stream);
int mmfn_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int P, const float* mean, const float* inv_std,
                          void* stream);
int mmfn_lidar_splat_f32(const float* pts, int B, int N, int stride_floats, float* out, int flip_y, void* stream);
int mmfn_lane_to_vector_f32(const float* lane, float* vec, int64_t R, int n, void* stream);

/* ---- VectorNet polyline max-pool + concat (model_vec.py:269-282) -------------------------------- */
/* Lane attention of VectorNet for QUERY 0 ONLY (model_vec.py:301-324 MaskSelfAttention, of which VectornetEncoder.forward
 * :412 consumes lane 0's row alone): any number of lanes, keys >= kv_len[b] masked as the reference's -1e9 fill, kv_len 0 =
 * uniform attention.  qkv [B*L, 3*heads*64] = [q | k | v]; att0 [B, heads*64]; prob [B, heads, L] (saved for the backward);
 * the backward writes ALL of dqkv (dq of row 0, zeros for the dead query rows, dk / dv of every lane). */
int mmfn_lane0_attention_fwd_f32(const float* qkv, const int32_t* kv_len, int B, int L, int heads, int head_dim, float scale,
                                 float* att0, float* prob, void* stream);
int mmfn_lane0_attention_bwd_f32(const float* qkv, const float* prob, const float* g_att0, const int32_t* kv_len, int B, int L,
                                 int heads, int head_dim, float scale, float* dqkv, void* stream);
int mmfn_polyline_pool_fwd_f32(const float* y, float* out, uint8_t* arg, int R, int V, int H, int last, void* stream);
int mmfn_polyline_pool_bwd_f32(const float* gout, const uint8_t* arg, float* gy, int R, int V, int H, int last,
                               void* stream);

/* ---- radar GAT pieces (model_rad.py:800-884); the matrix products use the batched mmfn_gemm_f32 ---- */
int mmfn_elu_fwd_f32(const float* x, float* y, int64_t n, void* stream);
int mmfn_elu_bwd_f32(const float* g, const float* y, float* dx, int64_t n, void* stream);
/* p = softmax(adj > 0 ? LeakyReLU_alpha(e_pre) : -9e15) over rows of N; att = dropout(p) */
int mmfn_gat_softmax_fwd_f32(const float* e_pre, const float* adj, float alpha, float* p, float* att, int R, int N,
                             float drop_p, const uint64_t* rng_state, uint32_t rng_stream, void* stream);
int mmfn_gat_softmax_bwd_f32(const float* g_att, const float* p, const float* e_pre, const float* adj, float alpha,
                             float* g_epre, int R, int N, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                             void* stream);
/* y[orow] = log_softmax(x[row]) over C channels; swap: row (b, i*8+j) -> orow (b, j*8+i), i.e. the
 * view(B,8,8,512).transpose(1,3) of model_rad.py:883 expressed on NHWC rows */
int mmfn_log_softmax_fwd_f32(const float* x, float* y, int R, int C, int swap, void* stream);
int mmfn_log_softmax_bwd_f32(const float* g, const float* y, float* dx, int R, int C, int swap, void* stream);

/* ---- bf16 training mode (BASELINE configs[2]): the same kernels with bf16 activations in HBM ---------------------------
 * `void*` tensors are bf16 (raw 16-bit words, channels-last / [rows, C] like their fp32 twins); statistics, parameters, their
 * gradients, lse / delta and every `float*` stay fp32.  Each entry replaces the aten dispatch its _f32 twin cites, as
 * torch.autocast(bfloat16) would run it - except that the activations never exist in fp32.  Arithmetic is fp32 in registers. */
/* weight shadows, once per step: the flat fp32 parameter buffer rounded to bf16 at the same offsets ... */
int mmfn_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* ... and back (exact): the gradient buckets of the bf16 mode cross xGMI as bf16 (mmfn_allreduce_sum_bf16, include/mmfn_comm.h) and
 * return to the fp32 gradient buffer AdamW reads; n % 4 == 0 */
int mmfn_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
/* ... and transposed copies for the data gradients.  table: DEVICE array of n_entries records
 * { const float* src; bf16* dst; int32 R, T, C; int32 tiles_c; int64 tile0 } (40 bytes): dst[c][t][r] = src[r][t][c]
 * (Linear [out,in] -> [in,out] with T = 1; filters [Cout][taps][Cin] -> [Cin][taps][Cout]); tile0 = running sum of
 * T * ceil(R/32) * ceil(C/32), ascending; total_tiles = the grand total. */
int mmfn_shadow_transpose_bf16(const void* table, int n_entries, int64_t total_tiles, void* stream);
int mmfn_bn_train_stats_bf16(const void* x, int64_t M, int C, float eps, float momentum, float* mean, float* rstd,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, void* workspace, void* stream);
/* x_is_f32: the convolution output x (and, in the backward, its gradient dx) is fp32 - the 7x7 stems, whose 3- / 2-channel
 * convolutions stay on the fp32 kernel */
int mmfn_bn_apply_bf16(const void* x, int x_is_f32, const void* res, void* y, int64_t M, int C, const float* mean,
                       const float* rstd, const float* weight, const float* bias, int relu, void* stream);
int mmfn_bn_bwd_bf16(const void* g, const void* y, const void* x, int x_is_f32, int64_t M, int C, const float* mean,
                     const float* rstd, const float* weight, void* dx, void* ge_out, float* dweight, float* dbias,
                     void* workspace, void* stream);
/* The fusion transformers keep their residual stream (token matrix x and its gradient) in fp32 also in the bf16 mode - what
 * torch.autocast does to model_vec.py:124-132 (x + attn(ln1(x)): LayerNorm output and Linear operands bf16, the sum fp32); rounding
 * that stream to bf16 alone costs 0.03-0.04 of gradient cosine per backward stage (tools/experiments/bf16_where.py).  Hence:
 *   x_is_f32 / stream_is_f32: x (and dres, dx) are fp32; y, g and dx_dropped (the GEMM operands) stay bf16;
 *   dx_dropped with drop_p == 0 is the plain bf16 copy of dx (rng_state may be NULL then). */
int mmfn_layernorm_fwd_bf16(const void* x, int x_is_f32, const float* weight, const float* bias, void* y, float* mean, float* rstd,
                            int M, int C, float eps, int act, void* stream);
int mmfn_layernorm_bwd_partial_bf16(const void* g, const void* x, int stream_is_f32, const float* weight, const float* bias,
                                    const float* mean, const float* rstd, const void* dres, void* dx, int M, int C, int act,
                                    void* dx_dropped, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                                    int want_colsum, float* partials, void* stream);
int mmfn_colsum_bf16(const void* in, int64_t M, int C, int ld, float* out, void* workspace, void* stream);
/* out[c] = sum over rows of partials[row][0][c] (mmfn_gemm_bf16 stats_mode 1: the bias gradient from the producing GEMM's epilogue) */
int mmfn_colsum_partials_f64(const double* partials, int rows, int C, float* out, void* stream);
/* BatchNorm backward whose two reductions were emitted as partial rows by the GEMM that produced g (mmfn_gemm_bf16 stats_mode 2):
 * finalize (dweight, dbias, means) + the elementwise pass; arguments as mmfn_bn_bwd_bf16 (x / dx bf16) */
int mmfn_bn_bwd_partials_bf16(const double* partials, int rows, const void* g, const void* y, const void* x, int64_t M, int C,
                              const float* mean, const float* rstd, const float* weight, void* dx, void* ge_out, float* dweight,
                              float* dbias, void* workspace, void* stream);
/* The finalize half of mmfn_bn_bwd_partials_bf16 alone: dweight, dbias and means[2][C] = (mean ge, mean ge * xhat) from partial rows
 * [rows][2][C]; the elementwise half of native_batch_norm_backward then runs in the loader of mmfn_conv3x3_halo_bf16 (pro 2). */
int mmfn_bn_bwd_finalize_f64(const double* partials, int rows, int64_t M, int C, float* dweight, float* dbias, float* means, void* stream);
int mmfn_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream);
int mmfn_maxpool3x3s2_bwd_bf16(const void* gy, const uint8_t* idx, void* gx, int B, int H, int W, int C, void* stream);
/* tok_is_f32 / gtok_is_f32: the token matrix / the token gradient is the fp32 residual stream (features stay bf16); the fp32
 * token gradient goes through mmfn_tokens_bwd_f32 */
int mmfn_tokens_fwd_bf16(const void* const* feats, int n_modal, const int32_t* frames, int B, int S, int C, const float* pos, const float* vel_w,
                         const float* vel_b, const float* velocity, void* tok, int tok_is_f32, float drop_p,
                         const uint64_t* rng_state, uint32_t rng_stream, void* stream);
int mmfn_tokens_bwd_bf16(void* gtok, int B, int T, int C, const float* velocity, float* dpos, float* dvel_w, float* dvel_b,
                         float drop_p, const uint64_t* rng_state, uint32_t rng_stream, void* workspace, void* stream);
int mmfn_upsample_add_fwd_bf16(const void* feat, const void* tok, void* out, int B, int S, int C, int T, int m, int frames, void* stream);
int mmfn_upsample_adj_bf16(const void* G, void* gtok, int B, int S, int C, int T, int m, int frames, void* stream);
int mmfn_pool_bcast_add_bf16(const void* G, const void* gtok, int gtok_is_f32, void* dF, int B, int S, int C, int T, int m, int frames,
                             void* stream);
int mmfn_gap_sum_fwd_bf16(const void* const* feats, int n, const int32_t* frames, int B, int P, int C, float* out, void* stream);
int mmfn_gap_sum_bwd_bf16(const float* g, void* const* outs, int n, const int32_t* frames, int B, int P, int C, void* stream);
/* [B, R, Cc] -> [B, Cc, R] across the precision boundary: VectorNet (fp32 inside) -> the bf16 map feature, and its gradient back */
int mmfn_transpose_f32_to_bf16(const float* in, void* out, int B, int R, int Cc, void* stream);
int mmfn_transpose_bf16_to_f32(const void* in, float* out, int B, int R, int Cc, void* stream);
/* T = 64 / 128 / 192 tokens (the fusion transformers); operands widened to fp32 on their way into LDS */
int mmfn_attention_fwd_bf16(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int T,
                            int NH, int HS, float scale, const int32_t* kv_len, float drop_p, const uint64_t* rng_state,
                            uint32_t rng_stream, void* stream);
int mmfn_attention_bwd_bf16(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, int ldg, int B, int T, int NH, int HS,
                            float scale, const int32_t* kv_len, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMFN_HIP_H */
