/* pointnet_refine_hip.h - C ABI of the MI355X (gfx950) LineRefineNet hot-path library.
 *
 * The reference (1pathplanningzzj/pointnet_refine) has no FFI or plugin interface:
 * its hot path is the Python nn.Module surface of src/model.py.  This library is the
 * native layer the replacement nn.Modules (pointnet_refine_amd/model.py) call through
 * ctypes.  Every entry point below names the reference code it replaces.
 *
 * Conventions
 *  - all tensors are fp32, POINT-MAJOR: rows = B*N points, columns = channels,
 *    row-major with the stated leading dimension;
 *  - every pointer is DEVICE memory owned by the caller (PyTorch); the library
 *    allocates nothing, keeps no pointer after return and launches only on `stream`;
 *  - return 0 on success, a negative code otherwise; prh_last_error() gives the text
 *    (thread-local);
 *  - re-entrant; the autograd engine calls the backward entry points from its own
 *    thread, so each call sets the device from `device` before launching.
 */
#ifndef POINTNET_REFINE_HIP_H
#define POINTNET_REFINE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRH_OK 0
#define PRH_ERR_ARG (-1)
#define PRH_ERR_WORKSPACE (-2)
#define PRH_ERR_HIP (-3)

#define PRH_MAX_LAYERS 8

/* One shared-MLP layer: 1x1 Conv1d (= Linear over points) + BatchNorm1d (+ ReLU).
 * Reference: nn.Conv1d/nn.BatchNorm1d pairs of src/model.py:10-20 (encoder),
 * :23-27 (fusion), :150-159 (point_mlp). */
typedef struct {
  const float* w;        /* [cout, cin]  (Conv1d weight (cout,cin,1) viewed 2-D) */
  const float* b;        /* [cout] */
  const float* gamma;    /* [cout] BatchNorm weight */
  const float* beta;     /* [cout] BatchNorm bias */
  float* running_mean;   /* [cout] updated in train mode */
  float* running_var;    /* [cout] updated in train mode (unbiased variance) */
  int64_t* num_batches_tracked; /* scalar, +1 in train mode; may be NULL */
  int cin, cout;
} prh_bn_layer;

/* Gradients of one shared-MLP layer (any pointer may be NULL = not wanted). */
typedef struct {
  float* dw;      /* [cout, cin] */
  float* db;      /* [cout] */
  float* dgamma;  /* [cout] */
  float* dbeta;   /* [cout] */
} prh_bn_layer_grad;

/* MultiScalePointNetEncoder parameters, src/model.py:7-37. */
typedef struct {
  int in_channel;          /* C (4 in LineRefineNet; any C>=4 for the bare encoder) */
  int out_dim;             /* 1024 */
  prh_bn_layer conv[5];    /* conv1..5 + bn1..5 */
  prh_bn_layer fusion;     /* fusion.0 (cin = 64+128+256+512+out_dim) + fusion.1 */
  const float* gate_w1;    /* intensity_gate.0.weight [64] (Conv1d(1,64,1)) */
  const float* gate_b1;    /* [64] */
  const float* gate_w2;    /* intensity_gate.2.weight [out_dim, 64] */
  const float* gate_b2;    /* [out_dim] */
} prh_encoder_params;

typedef struct {
  prh_bn_layer_grad conv[5];
  prh_bn_layer_grad fusion;
  float* d_gate_w1; float* d_gate_b1; float* d_gate_w2; float* d_gate_b2;
} prh_encoder_grads;

/* Activations the forward keeps for the backward (caller-allocated).
 *   cat = 64+128+256+512+out_dim                                            */
typedef struct {
  float* z_cat;      /* [P, cat]      pre-BN outputs of conv1..5, concatenated by column */
  float* z_fus;      /* [P, out_dim]  pre-BN output of the fusion conv */
  float* gate;       /* [P, out_dim]  0.5+0.5*sigmoid(.) ; may be NULL when no backward */
  float* bn_scale;   /* [cat+out_dim] gamma*rstd            (BN as y = z*scale+shift) */
  float* bn_shift;   /* [cat+out_dim] beta - mean*scale */
  float* bn_mean;    /* [cat+out_dim] */
  float* bn_rstd;    /* [cat+out_dim] */
  int32_t* argmax;   /* [B, out_dim]  first arg-max point of the max-pool; NULL = skip */
  float* op_amax;    /* [8] or NULL.  Training with the split-fp16 cores: the forward writes the
                      * largest activation of conv1..5 ([0..4]), their maximum ([5]) and an upper
                      * bound of max(fused) ([6]) here, taken from the statistics epilogues, and
                      * the backward reads them as operand scales of its wgrads instead of
                      * re-measuring.  Pass the same buffer to both calls
                      * (and keep the GEMM mode unchanged in between), or NULL to both. */
} prh_encoder_saved;

/* Bytes of scratch the encoder entry points need for P = B*N points
 * (backward = 0: prh_encoder_forward only; 1: also prh_encoder_backward; 2: prh_encoder_backward with
 * d_fused_scratch = 1, which needs 4 * out_dim bytes per point less). */
size_t prh_encoder_workspace_bytes(int B, int N, int in_channel, int out_dim, int backward);

/* MultiScalePointNetEncoder.forward, src/model.py:39-62.
 *   ctx      [B,N,C] point-major (the reference takes the (B,C,N) transpose view)
 *   fused    [B,N,out_dim]          (reference returns its (B,out_dim,N) transpose)
 *   gfeat    [B,2*out_dim] = [max over N | mean over N]; NULL = skip pooling
 *   training 1: batch statistics + running-stat update, 0: running statistics */
int prh_encoder_forward(const prh_encoder_params* prm, const float* ctx, int B, int N,
                        int training, float momentum, float eps,
                        float* fused, float* gfeat, const prh_encoder_saved* saved,
                        void* workspace, size_t workspace_bytes, int device, void* stream);

/* Backward of the above (autograd of src/model.py:39-62).
 *   d_fused [B,N,out_dim] or NULL, d_gfeat [B,2*out_dim] or NULL (at least one)
 *   d_ctx   [B,N,C] or NULL
 *   training must equal the forward's flag.
 * d_gfeat is read only; saved.gate is consumed (overwritten in place).  d_fused is read only unless
 * d_fused_scratch = 1 (needs d_fused != NULL and d_gfeat == NULL - the path LineRefineNet takes): then the
 * library turns the caller's d_fused buffer into the fusion layer's gradient scratch (dy_f, then dz_f, in place)
 * instead of carving one from the workspace - 17 GB less at 4.19 M points; its content is undefined afterwards. */
int prh_encoder_backward(const prh_encoder_params* prm, const float* ctx, int B, int N,
                         int training, float* d_fused, const float* d_gfeat, int d_fused_scratch,
                         const prh_encoder_saved* saved, const prh_encoder_grads* grads,
                         float* d_ctx, void* workspace, size_t workspace_bytes, int device,
                         void* stream);

/* ---- bf16 mode (BASELINE config 3, "bf16 training"; prh_set_gemm_mode(4)) -----------------
 * Same encoder (src/model.py:39-62) with ONE bf16 MFMA product per MAC and the activations the
 * forward keeps - and `fused`, and the gradient buffers of the backward - STORED in bf16
 * (uint16_t = raw bf16 bits).  Accumulation, BatchNorm statistics (taken from the rounded
 * values), parameters and parameter gradients are fp32.  Channel widths must be multiples of 8
 * (the context rows are padded to 8 channels internally). */
typedef struct {
  uint16_t* z_cat;   /* [P, cat]      bf16 pre-BN outputs of conv1..5 */
  uint16_t* z_fus;   /* [P, out_dim]  bf16 pre-BN output of the fusion conv */
  uint16_t* gate;    /* [P, out_dim]  bf16 0.5+0.5*sigmoid(.) ; may be NULL when no backward */
  float* bn_scale;   /* [cat+out_dim] as in prh_encoder_saved */
  float* bn_shift;
  float* bn_mean;
  float* bn_rstd;
  int32_t* argmax;   /* [B, out_dim] or NULL */
} prh_encoder_saved_bf16;
size_t prh_encoder_bf16_workspace_bytes(int B, int N, int in_channel, int out_dim, int backward);
/* fused [B,N,out_dim] bf16; everything else as prh_encoder_forward */
int prh_encoder_forward_bf16(const prh_encoder_params* prm, const float* ctx, int B, int N, int training,
                             float momentum, float eps, uint16_t* fused, float* gfeat,
                             const prh_encoder_saved_bf16* saved, void* workspace, size_t workspace_bytes,
                             int device, void* stream);
/* d_fused [B,N,out_dim] bf16 or NULL; everything else as prh_encoder_backward */
int prh_encoder_backward_bf16(const prh_encoder_params* prm, const float* ctx, int B, int N, int training,
                              const uint16_t* d_fused, const float* d_gfeat, const prh_encoder_saved_bf16* saved,
                              const prh_encoder_grads* grads, float* d_ctx, void* workspace,
                              size_t workspace_bytes, int device, void* stream);
/* nn.Linear on a bf16 input (context_proj applied to the bf16 `fused`, src/model.py:147,194):
 * y fp32 [rows,n] = act(x W^T + b); backward: dy fp32 -> dx bf16 [rows,k], dw [n,k], db [n]
 * (any may be NULL).  k, n, ldx multiples of 8. */
size_t prh_linear_bf16_workspace_bytes(int rows, int k, int n, int backward);
int prh_linear_forward_bf16(const uint16_t* x, long ldx, const float* w, const float* b, float* y, int rows, int k,
                            int n, int relu, void* workspace, size_t workspace_bytes, int device, void* stream);
int prh_linear_backward_bf16(const uint16_t* x, long ldx, const float* w, const float* dy, uint16_t* dx, float* dw,
                             float* db, int rows, int k, int n, void* workspace, size_t workspace_bytes, int device,
                             void* stream);

/* Inference-only cross-attention with the key / value projections folded in (src/model.py:119-128
 * in eval mode): attention over the RAW rows X = memory + pos and Y = memory, which
 * are the same for all six layers, instead of over per-layer projected buffers -
 *   softmax(Q_h K_h^T) V_h = softmax((Q_h Wk_h) X^T) Y Wv_h^T + bv_h
 * (the key bias is constant along the keys and drops out of the softmax).  prh_cast_perm_bf16 makes
 * the bf16 row image the kernel reads from fp32 [rows, 256] (ld >= 256); prh_attn_fold_forward:
 * q [B*M, 256] projected (unscaled) queries, wk / wv [256, 256] and bv [256] = rows d..2d and 2d..3d
 * of the layer's packed in_proj parameters, o [B*M, 256] (before out_proj).  8 heads of 32 channels,
 * M <= 32, bf16 products with fp32 accumulation (BASELINE config 5). */
int prh_cast_perm_bf16(const float* src, long ld, uint16_t* dst, long rows, int device, void* stream);
/* ... or both images in one pass from their sources: x16 = bf16(memory + pos), y16 = bf16(memory) with
 * pos = PositionalEncoding(xyz) = relu(xyz W0^T + b0) W2^T + b2 (src/model.py:64-75; W0 [256,3], W2 [256,256]).
 * The hidden layer and memory + pos never exist in fp32: 2 KB per point instead of 7 (pos_hidden, Linear
 * with residual, two casts).  xyz rows read in place (ld >= 3), memory [rows, 256] (ld >= 256). */
int prh_posmem_images(const float* xyz, long ldx, const float* w0, const float* b0, const float* w2, const float* b2,
                      const float* memory, long ldm, long rows, uint16_t* x16, uint16_t* y16, int device, void* stream);
int prh_attn_fold_forward(const float* q, long ldq, const uint16_t* x16, const uint16_t* y16, const float* wk, long ldwk,
                          const float* wv, long ldwv, const float* bv, float* o, long ldo, int B, int M, int N, int H,
                          float scale, int device, void* stream);
/* bf16 mode, decoder side: the cross-attention key / value projections of all six layers
 * (src/model.py:123-126) write their [rows, 6*256] outputs in bf16 and receive bf16 gradients.
 *   prh_linear_forward_out16: y bf16 [rows,n] = x W^T + b from an fp32 x;
 *   prh_linear_backward_dy16: dy bf16 -> dx fp32 [rows,k], dw [n,k], db [n] (any may be NULL);
 *   workspace: prh_linear_bf16_workspace_bytes(rows, k, n, backward).
 *   prh_attn_forward_kv16 / prh_attn_backward_kv16: prh_attn_forward / prh_attn_backward with K, V
 *   (and dK, dV) in bf16 storage - leading dimensions in elements, multiples of 8. */
int prh_linear_forward_out16(const float* x, long ldx, const float* w, const float* b, uint16_t* y, int rows, int k,
                             int n, void* workspace, size_t workspace_bytes, int device, void* stream);
int prh_linear_backward_dy16(const float* x, long ldx, const float* w, const uint16_t* dy, float* dx, float* dw,
                             float* db, int rows, int k, int n, void* workspace, size_t workspace_bytes, int device,
                             void* stream);
int prh_attn_forward_kv16(const float* q, long ldq, const uint16_t* k, long ldk, const uint16_t* v, long ldv, float* o,
                          long ldo, float* lse, int B, int M, int N, int H, float scale, float dropout_p,
                          unsigned seed, int device, void* stream);
int prh_attn_backward_kv16(const float* q, long ldq, const uint16_t* k, long ldk, const uint16_t* v, long ldv,
                           const float* o, long ldo, const float* lse, const float* dout, long lddo, float* dq,
                           long lddq, uint16_t* dk, long lddk, uint16_t* dv, long lddv, int B, int M, int N, int H,
                           float scale, float dropout_p, unsigned seed, int device, void* stream);

/* ---- fused EVAL-mode encoder (src/model.py:39-62 with every BatchNorm in eval mode, + :147,194)
 * One kernel takes context [B,N,C] to memory [B,N,256] = context_proj(fused) - and, optionally,
 * fused [B,N,1024] and global_feat [B,2048] - with BatchNorm folded into the conv weights: a
 * tile of points stays resident on the CU, the weights stream from L2, no activation touches
 * HBM.  planes = 1: fp16 operands (BASELINE config 5, "batched fp16 forward"; parity gate 5e-2);
 * planes = 2: two fp16 planes, three MFMA products, fp32-level error (parity gate 1e-4).
 * Built for the reference's widths (64/128/256/512/1024, gate hidden 64, context_proj 256).
 *   prepare: fold + split the weights ONCE per set of weights into `image`
 *            (prh_encoder_fused_image_bytes bytes, 256-byte aligned); proj_w/proj_b [256,1024]/[256]
 *            or NULL (encoder API only);
 *   forward: any of memory / fused / gfeat may be NULL (at least one given); workspace
 *            (prh_encoder_fused_workspace_bytes) is needed for gfeat only.  Activations travel between
 *            the layers as fp16 planes: `saturated` (device word, may be NULL) is INCREMENTED for every
 *            group of four activations holding a value above 65504 (clamped) - a non-zero count means
 *            the result is not the module's and must be discarded (the caller zeroes the word). */
size_t prh_encoder_fused_image_bytes(int planes, int in_channel);
int prh_encoder_fused_prepare(const prh_encoder_params* prm, float eps, const float* proj_w, const float* proj_b,
                              int planes, void* image, size_t image_bytes, int device, void* stream);
size_t prh_encoder_fused_workspace_bytes(int B, int N, int planes);
int prh_encoder_fused_forward(const void* image, int planes, int in_channel, int has_proj, const float* ctx, int B,
                              int N, float* memory, float* fused, float* gfeat, unsigned* saturated, void* workspace,
                              size_t workspace_bytes, int device, void* stream);

/* nn.Linear forward y = act(x W^T + b): context_proj (src/model.py:147,194) and any
 * other Linear on the path.  x [rows,k] (ld ldx), w [n,k], y [rows,n]; relu: 0/1.
 * k and ldx must be multiples of 4.  workspace (may be NULL: exact fp32 MFMA core only) holds the
 * split weight image (and the operand-scale words) of the large-GEMM cores. */
size_t prh_linear_forward_workspace_bytes(int rows, int k, int n);
int prh_linear_forward(const float* x, long ldx, const float* w, const float* b, float* y,
                       int rows, int k, int n, int relu, void* workspace, size_t workspace_bytes,
                       int device, void* stream);

/* Same with a caller-supplied device scalar x_amax >= max|x| (NULL = measured by a read pass when
 * the split-fp16 core serves the GEMM): e.g. prh_encoder_saved.op_amax[6] for context_proj. */
int prh_linear_forward_ex(const float* x, long ldx, const float* w, const float* b, float* y,
                          int rows, int k, int n, int relu, const float* x_amax, void* workspace,
                          size_t workspace_bytes, int device, void* stream);

/* Same plus an optional residual input added in the GEMM epilogue: y = act(x W^T + b + resid),
 * resid [rows,n] (ld ldres; NULL = none) - e.g. k-input = memory + pos_emb(xyz)
 * (src/model.py:123-126) with the addition riding on the second Linear of the
 * positional-encoding MLP - and an optional device scalar w_amax = max|w|. */
int prh_linear_forward_full(const float* x, long ldx, const float* w, const float* b, const float* resid,
                            long ldres, float* y, int rows, int k, int n, int relu, const float* x_amax,
                            const float* w_amax, float dropout_p, unsigned dropout_seed, void* workspace,
                            size_t workspace_bytes, int device, void* stream);
/* dropout_p > 0: y = dropout(act(x W^T + b + resid)) with the keep decision a counter hash of
 * (dropout_seed, row, column) and survivors scaled by 1 / (1 - p) - the FFN hidden layer of the decoder
 * (src/model.py:131) leaves the GEMM epilogue already dropped out; no mask tensor exists, the backward
 * reads the decision off the output (prh_relu_mask_absmax with scale = 1 / (1 - p)). */

/* Operand maxima for the split-fp16 cores, measured once by the caller and handed to every GEMM
 * that reads the operand (x_amax / w_amax / dy_amax arguments; NULL = the launch measures it):
 * out[0] = max |x| over [rows, cols] (ld >= cols).  prh_linear_uses_operand_maxima: 1 when a
 * Linear of this shape runs on those cores in the current GEMM mode (otherwise the maxima are
 * not read and need not be measured). */
size_t prh_operand_absmax_workspace_bytes(void);
int prh_operand_absmax(const float* x, long ld, long rows, int cols, float* out, void* workspace,
                       size_t workspace_bytes, int device, void* stream);
/* ReLU backward of a Linear with the ReLU fused into its epilogue (src/model.py:131 linear1 +
 * activation; :162-166 reg_branches): out = y > 0 ? dy : 0 and amax_out[0] = max|out| (the operand
 * maximum of the split-fp16 backward GEMMs) in one pass.  n elements, n % 4 == 0, contiguous.
 * Workspace: prh_operand_absmax_workspace_bytes(). */
int prh_relu_mask_absmax(const float* dy, const float* y, float* out, long n, float scale, float* amax_out,
                         void* workspace, size_t workspace_bytes, int device, void* stream);
/* scale: 1 for a plain ReLU; 1 / (1 - p) when y came out of a ReLU + dropout epilogue (out = y > 0 ? dy * scale : 0). */
int prh_linear_uses_operand_maxima(int rows, int k, int n);

/* First layer of the positional-encoding MLP (src/model.py:64-75, nn.Linear(3, hidden) + ReLU)
 * as one elementwise pass: h[r,c] = relu(b0[c] + sum_j xyz[r*ld + j] w0[c*3 + j]), j < 3.
 * xyz rows are read in place with leading dimension ld >= 3 (ld = C for (B,N,C) context rows);
 * hidden: a power of two in [4, 1024].  backward: dw0 [hidden,3] and db0 [hidden] (either may be
 * NULL) from dh [rows,hidden] masked by h > 0; dxyz [rows,3] (contiguous; NULL = skip; needs w0
 * and hidden <= 256) for the decoder's query positions, which carry gradients. */
int prh_pos_hidden_forward(const float* xyz, long ld, const float* w0, const float* b0, float* h,
                           long rows, int hidden, int device, void* stream);
size_t prh_pos_hidden_backward_workspace_bytes(long rows, int hidden);
int prh_pos_hidden_backward(const float* xyz, long ld, const float* h, const float* dh, const float* w0,
                            float* dxyz, float* dw0, float* db0, long rows, int hidden, void* workspace,
                            size_t workspace_bytes, int device, void* stream);

/* nn.Linear with n <= 4 outputs as one HBM pass (the regression heads' Linear(128, 3),
 * src/model.py:162-166): y [rows,n] = x [rows,k] w[n,k]^T + b.  k/4 must be a power of two
 * <= 64 (k = 4 ... 256); x, y, dx contiguous.  backward: dx (NULL = skip), dw [n,k], db [n]
 * (either may be NULL). */
int prh_linear_small_forward(const float* x, const float* w, const float* b, float* y, long rows, int k,
                             int n, int device, void* stream);
size_t prh_linear_small_backward_workspace_bytes(long rows, int k, int n);
int prh_linear_small_backward(const float* x, const float* w, const float* dy, float* dx, float* dw,
                              float* db, long rows, int k, int n, void* workspace, size_t workspace_bytes,
                              int device, void* stream);

/* nn.Linear backward: dx = dy W (NULL = skip), dw = dy^T x, db = colsum(dy).
 * n and k multiples of 4. */
size_t prh_linear_backward_workspace_bytes(int rows, int k, int n);
int prh_linear_backward(const float* x, long ldx, const float* w, const float* dy, float* dx,
                        float* dw, float* db, int rows, int k, int n, void* workspace,
                        size_t workspace_bytes, int device, void* stream);

/* x_amax, dy_amax: optional device scalars bounding max|x| / max|dy| from above (NULL = measured) */
int prh_linear_backward_ex(const float* x, long ldx, const float* w, const float* dy, float* dx,
                           float* dw, float* db, int rows, int k, int n, const float* x_amax,
                           const float* dy_amax, void* workspace, size_t workspace_bytes, int device,
                           void* stream);
/* ... and w_amax = max|w| (the dgrad reads W^T, whose maximum is the same) */
int prh_linear_backward_full(const float* x, long ldx, const float* w, const float* dy, float* dx,
                             float* dw, float* db, int rows, int k, int n, const float* x_amax,
                             const float* dy_amax, const float* w_amax, void* workspace,
                             size_t workspace_bytes, int device, void* stream);

/* Stack of <= PRH_MAX_LAYERS shared-MLP layers applied to x [P,cin0]:
 * LineRefineNet.point_mlp, src/model.py:150-159,200-201 (relu_last = 0).
 *   z_cat [P, sum(cout)] pre-BN outputs (kept for backward), y [P, cout_last]. */
size_t prh_mlp_stack_workspace_bytes(int P, int n_layers, const prh_bn_layer* layers);
int prh_mlp_stack_forward(const prh_bn_layer* layers, int n_layers, int relu_last,
                          const float* x, int P, int training, float momentum, float eps,
                          float* z_cat, float* y, float* bn_scale, float* bn_shift,
                          float* bn_mean, float* bn_rstd, void* workspace,
                          size_t workspace_bytes, int device, void* stream);
int prh_mlp_stack_backward(const prh_bn_layer* layers, int n_layers, int relu_last,
                           const float* x, int P, int training, const float* dy,
                           const float* z_cat, const float* bn_scale, const float* bn_shift,
                           const float* bn_mean, const float* bn_rstd,
                           const prh_bn_layer_grad* grads, float* dx, void* workspace,
                           size_t workspace_bytes, int device, void* stream);

/* Fused cross-attention core of DetrTransformerDecoderLayer.cross_attn (src/model.py:84,123-126:
 * nn.MultiheadAttention(256, 8 heads, dropout on the attention weights), everything between
 * the in-projections and the out-projection): o = dropout(softmax(q k^T * scale)) v per head
 * of 32 channels, exact fp32 MFMA, online softmax.
 *   q [B*M, H*32] (ld ldq), k/v [B*N, H*32] (ld ldk/ldv: may be column blocks of a wider
 *   projection buffer), o [B*M, H*32], lse [B,H,M] (kept for the backward).
 * dropout_p = 0 in eval mode; the mask is a counter-based hash of (seed, b, h, query, key). */
int prh_attn_forward(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                     float* o, long ldo, float* lse, int B, int M, int N, int H, float scale,
                     float dropout_p, unsigned seed, int device, void* stream);
int prh_attn_backward(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                      const float* o, long ldo, const float* lse, const float* dout, long lddo,
                      float* dq, long lddq, float* dk, long lddk, float* dv, long lddv, int B, int M,
                      int N, int H, float scale, float dropout_p, unsigned seed, int device,
                      void* stream);

/* Per-line context builder: the crop / weight / sample / centre step of
 * LaneRefineDataset.__getitem__ (src/dataset.py:210-234) and process_single_line
 * (inference_whole_scene.py:98-121), weighted_sampling (src/dataset.py:78-130), batched over the
 * lines of one scene.
 *   cloud  [npts,4] xyz + intensity          dense [n_lines,n_dense,3] polyline resampled to
 *   line   [n_lines,m,3] resampled to m pts        n_dense points (200 in the reference)
 *   out    [n_lines,n_samples,4] xyz centred on the line's mean, raw intensity
 *   counts [n_lines] points inside the tube (before the max_candidates cap)
 *   dbg_weights [n_lines,max_candidates] or NULL: unnormalised sampling weights of the
 *               candidates in cloud order (lines with counts > n_samples only)
 * Crop masks and weights match the reference to fp32 rounding; the draws replace
 * numpy.random.choice by hashing (seed, line, point index): same distribution, not the same
 * sample.  Deterministic for a given seed.  Lines with more than max_candidates points in the
 * tube use the first max_candidates in cloud order (counts still reports the true number). */
size_t prh_context_workspace_bytes(int npts, int n_lines, int max_candidates);
int prh_context_build(const float* cloud, int npts, const float* dense, int n_dense, const float* line,
                      int m, int n_lines, float radius, float decay_scale, int n_samples,
                      int max_candidates, unsigned long long seed, float* out, int32_t* counts,
                      float* dbg_weights, void* workspace, size_t workspace_bytes, int device,
                      void* stream);

/* Row f1, query side of DetrTransformerDecoderLayer (src/model.py:117,128,133):
 *   y = LayerNorm(x + dropout(r)), nn.LayerNorm(256) semantics (eps, biased variance, affine),
 * rows x 256 fp32, one pass forward and one backward.  The dropout decision is a counter hash of
 * (seed, row, channel) - same distribution as nn.Dropout, not the same mask; dropout_p = 0 in
 * eval mode.  mean/rstd [rows] are saved by the forward for the backward (NULL = not kept).
 * The backward also produces dgamma / dbeta (sums over rows). */
int prh_add_dropout_layernorm_forward(const float* x, const float* r, const float* gamma, const float* beta,
                                      long rows, int channels, float eps, float dropout_p, unsigned seed,
                                      float* y, float* mean, float* rstd, int device, void* stream);
size_t prh_add_dropout_layernorm_workspace_bytes(void);
int prh_add_dropout_layernorm_backward(const float* dy, const float* x, const float* r, const float* gamma,
                                       const float* mean, const float* rstd, long rows, int channels,
                                       float dropout_p, unsigned seed, float* dx, float* dr, float* dgamma,
                                       float* dbeta, void* workspace, size_t workspace_bytes, int device,
                                       void* stream);

/* Row f3.  Deep-supervision L1 loss with its gradient in one pass (train.py:63-68,
 * train_dist.py:180-186: (1/L) sum_l nn.L1Loss(pred_l, target)):
 *   *loss (+)= sum_{l,e} |pred[l,e] - target[e]| / denom      d_pred[l,e] = sign(.) / denom
 * pred [n_layers, elems], target [elems]; denom = n_layers * elems of the full batch (a
 * micro-batched caller passes the full-batch denominator and accumulate = 1 from the second
 * chunk on); d_pred may be NULL.  loss is a device scalar.
 * geometry (device float[2] or NULL; elems must be xyz triples): the metrics the reference logs
 * every step (train_dist.py:190-203): [0] (+)= sum over points |target| / points (initial
 * point-to-point error), [1] (+)= sum |pred_last - target| / points (refined error). */
size_t prh_l1_loss_workspace_bytes(void);
int prh_l1_loss(const float* pred, const float* target, int n_layers, long elems, double denom,
                int accumulate, float* loss, float* d_pred, float* geometry, double points,
                void* workspace, size_t workspace_bytes, int device, void* stream);

/* torch.optim.Adam step (amsgrad off; train.py:40, train_dist.py:150) over flat, 16-byte
 * aligned fp32 buffers of n elements; step counts from 1. */
int prh_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, int device,
                  void* stream);

/* prh_attn_backward that also reports, per wave, the largest |dV| and |dK| it stored:
 * kv_amax_part [B*(H/4)*4][2] (or NULL) - reduced by the caller, it bounds the gradient operand
 * of the K/V projection's backward GEMMs (prh_linear_backward_ex dy_amax). */
int prh_attn_backward_ex(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                         const float* o, long ldo, const float* lse, const float* dout, long lddo,
                         float* dq, long lddq, float* dk, long lddk, float* dv, long lddv, int B, int M,
                         int N, int H, float scale, float dropout_p, unsigned seed, float* kv_amax_part,
                         int device, void* stream);

/* GEMM core selection (environment PRH_GEMM, or prh_set_gemm_mode at run time):
 *   split16 / 3 (default): large GEMMs on the split-fp16 cores - two fp16 planes per fp32 operand
 *            placed by a power-of-two scale from the operand's largest magnitude, three
 *            v_mfma_f32_16x16x32_f16 / 32x32x16 products, fp32 accumulation, fp32-level error;
 *   split / 1: split-bf16 cores - three bf16 planes, six v_mfma_f32_32x32x16_bf16 products,
 *            fp32-level error with no range assumption;
 *   fp32 / 0: the exact fp32 MFMA cores (v_mfma_f32_32x32x2_f32) everywhere.
 * All three are checked against the oracle at the same 1e-4 gate.
 *   bf16 / 2: opt-in REDUCED-PRECISION mode: plain bf16 operands, one MFMA product on the
 *            first-generation cores, fp32 accumulate and fp32 storage; parity gate 5e-2, not 1e-4.
 *   bf16s / 4: BASELINE config 3 ("bf16 training"): bf16 operands AND bf16 activation storage.
 *            The encoder and the Linear fed by it go through the *_bf16 entry points above; plain
 *            fp32-storage Linears of at least 512 x 64 x 64 run on the same bf16 core with their
 *            input converted in flight; everything else behaves as mode 2.  Parity gate 5e-2.
 * The mode is process-wide; set it before launching work, not concurrently with it, and keep
 * it unchanged between a forward call and its backward. */
int prh_set_gemm_mode(int mode);
int prh_get_gemm_mode(void);

/* Dropout under graph replay.  The attention and LayerNorm entry points take their dropout seed
 * as a host value, which a captured graph would freeze.  Register a device word here (process-
 * wide, like the GEMM mode; NULL = none): every dropout decision then hashes seed ^ f(*word),
 * read at kernel run time, so a caller that advances the word on the device before each replay
 * gets fresh masks while forward and backward of one step still agree. */
int prh_set_dropout_seed_source(const unsigned* device_word);

/* Raw GEMM cores, exported for the unit tests (tests/test_gemm_gpu.py).
 *   nt: c[m,n] = a[m,k] w[n,k]^T     tn: c[mo,ni] = a[p,mo]^T b[p,ni]          */
int prh_test_gemm_nt(const float* a, const float* w, float* c, int m, int n, int k,
                     void* workspace, size_t workspace_bytes, int device, void* stream);
size_t prh_test_gemm_tn_workspace_bytes(int p, int mo, int ni);
int prh_test_gemm_tn(const float* a, const float* b, float* c, float* colsum, int p, int mo,
                     int ni, void* workspace, size_t workspace_bytes, int device, void* stream);

/* Diagnostic: which XCD (XCC_ID) and CU (HW_ID) each workgroup of a `blocks` x 512-thread
 * launch with `lds_bytes` of dynamic LDS lands on; out[2*b] = XCC_ID, out[2*b+1] = HW_ID.
 * The GEMM kernels assume workgroup b runs on XCD b % 8 (xcd_remap); this checks it. */
int prh_test_xcc_map(int blocks, int lds_bytes, int* out, int device, void* stream);

/* Optional launch profiler used by bench.py: when enabled (capacity > 0) every GEMM launch is
 * bracketed by HIP events on the launch stream; prh_profile_read returns its duration and
 * the algorithmic FLOPs / bytes of that launch.  capacity 0 disables and frees the events. */
int prh_profile_enable(int capacity);
int prh_profile_count(void);
int prh_profile_reset(void);
int prh_profile_read(int i, char* name, int name_len, float* ms, double* flops, double* bytes);

const char* prh_last_error(void);
const char* prh_version(void);

#ifdef __cplusplus
}
#endif
#endif
