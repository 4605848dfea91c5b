/* mmskin.h -- C ABI of libmmskin_hip.so: the MI355X (gfx950) implementation of the
 * MultimodalModel forward/backward hot path of life-ufes/multimodal-model-skin-lesion-classifier.
 *
 * The reference has no FFI / plugin boundary of its own: its boundary is the Python nn.Module
 * contract of src/scripts/benchmark/models/multimodalIntraInterModal.py:13-416 (ctor :14-28,
 * forward :162).  This header is the build-defined C replacement for the ATen operators that module
 * invokes; each entry point names the reference call site whose arithmetic it replaces.  The Python
 * shim in multimodal-model-skin-lesion-classifier_amd/models/ binds these with ctypes (see
 * INTEGRATION.md).
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (no torch types, no allocation
 * inside compute calls); `stream` is a hipStream_t passed as void*; all calls are asynchronous on that
 * stream and return 0 on success, otherwise a MMSKIN_ERR_* code with mmskin_last_error() describing
 * it.  Matrices are row-major.  dtype selects the backbone compute type: MMSKIN_F32 = exact fp32
 * MFMA (parity mode), MMSKIN_BF16 = bf16 MFMA with fp32 accumulation (throughput mode).
 */
#ifndef MMSKIN_H
#define MMSKIN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMSKIN_F32 0
#define MMSKIN_BF16 1

const char* mmskin_last_error(void);
int mmskin_version(void);

/* ---------------------------------------------------------------------------------------------
 * Image encoder plan (resnet-18 / resnet-50).  Replaces `self.image_encoder(image)`
 * (multimodalIntraInterModal.py:167; factory loadImageModelClassifier.py:65-75) and its backward.
 * Parameters: ONE flat fp32 buffer in torchvision named_parameters() order (conv.weight OIHW,
 * bn.weight, bn.bias ...); buffers: ONE flat fp32 buffer (running_mean, running_var per BN).
 * tensor_info enumerates names/offsets/shapes so the host can expose them as state_dict entries. */
typedef struct mmskin_backbone* mmskin_backbone_t;
int mmskin_backbone_create(const char* arch, int batch, int height, int width, int dtype, mmskin_backbone_t* out);
void mmskin_backbone_destroy(mmskin_backbone_t h);
int mmskin_backbone_num_tensors(mmskin_backbone_t h, int kind /*0 params, 1 buffers*/);
int mmskin_backbone_tensor_info(mmskin_backbone_t h, int kind, int index, char* name, int name_cap,
                                int64_t* offset, int64_t* numel, int* ndim, int64_t* shape4);
int64_t mmskin_backbone_param_numel(mmskin_backbone_t h);
int64_t mmskin_backbone_buffer_numel(mmskin_backbone_t h);
int64_t mmskin_backbone_workspace_bytes(mmskin_backbone_t h);
int mmskin_backbone_feature_dim(mmskin_backbone_t h);
/* spatial size of the plan's output: 1x1 for pooled features; arch "densenet169-features" returns the norm5
 * feature map [N][1664][H/32][W/32] (fp32 NCHW), the `densenet.features` the reference's MD-Net keeps
 * (multimodalMDNet.py:72-76) */
int mmskin_backbone_feature_hw(mmskin_backbone_t h, int* out_h, int* out_w);
/* Introspection for parity tests: where unit `index` (conv+BN, in parameter order) keeps its raw conv
 * output x, its post-activation output y and its BN coefficient vectors inside the workspace.
 * info12 = {x_off, y_off, coef_off (bytes), rows, Cout, OH, OW, Cin, H, W, pool_off, scratch0_off}. */
int mmskin_backbone_num_units(mmskin_backbone_t h);
int mmskin_backbone_unit_info(mmskin_backbone_t h, int index, char* name, int name_cap, int64_t* info12);
/* Per-kernel-class timing with HIP events recorded on the launch stream (off by default).  Classes:
 * 0 conv fwd (implicit GEMM), 1 conv dgrad, 2 wgrad(+reduce), 3 BN fwd (finalize+apply), 4 BN bwd,
 * 5 weight staging, 6 stem pack/pool.  read() synchronises, returns the totals accumulated since
 * enable()/the last read(): milliseconds, algorithmic FLOPs, algorithmic bytes, launches; then resets. */
int mmskin_backbone_profile_enable(mmskin_backbone_t h, int on);
int mmskin_backbone_profile_read(mmskin_backbone_t h, double* ms7, double* flops7, double* bytes7, int64_t* launches7);
/* image_nchw: fp32 [batch,3,H,W]; features: fp32 [batch, feature_dim].  training!=0: batch-stat BN,
 * running stats updated in `buffers`, activations kept in `workspace` for the backward call. */
int mmskin_backbone_forward(mmskin_backbone_t h, const float* image_nchw, const float* params, float* buffers,
                            void* workspace, float* features, int training, void* stream);
/* param_grads: flat fp32, same layout as params; every element is written. */
/* Same forward from the uint8 NHWC batch the DataLoader decodes ([N][H][W][3]): Normalize + ToTensor of the
 * reference's transform (skinLesionDatasets.py:29,111-119: (u8/255 - mean)/std) run inside the stem packing
 * kernel, so the host->device copy is 1 byte per value instead of 4.  mean_std6: 6 HOST floats, mean rgb | std rgb. */
int mmskin_backbone_forward_u8(mmskin_backbone_t h, const uint8_t* image_nhwc, const float* mean_std6, const float* params,
                               float* buffers, void* workspace, float* features, int training, void* stream);
/* Grad-CAM / Grad-CAM++ consumers (src/services/XAI/models/cam.py:10-60, model_loader.py:36-41) hook the LAST nn.Conv2d
 * of image_encoder and differentiate the class score w.r.t. its output under model.eval().  With option
 * "keep_raw_eval" = 1 an eval forward keeps raw conv outputs (BatchNorm is then not folded into the convs);
 * last_conv_export returns that conv's output [N][C][OH][OW] (fp32) and last_conv_grad the gradient of the pooled
 * features w.r.t. it for a given d(score)/d(features) [N][C].  ResNet plans only. */
int mmskin_backbone_set_option(mmskin_backbone_t h, const char* key, int value);
/* option "reuse_staged" = 1 (serving): the caller vouches that parameters and BatchNorm buffers are unchanged since the
 * previous eval forward on this plan and workspace; the BN-folded staged weights and coefficient table already in the
 * workspace are then reused instead of rebuilt (ResNet plans; others restage).  Any training-mode or keep_raw_eval
 * forward invalidates the staged copy by itself. */
/* per-step device pointers: key "sd_mask" (EfficientNet plans) = fp32 [n_residual_blocks][batch] keep/scale factors of
 * torchvision's StochasticDepth(p, "row") for this training step (0 or 1/(1-p)); NULL disables stochastic depth */
int mmskin_backbone_set_pointer(mmskin_backbone_t h, const char* key, const void* device_ptr);
/* Gradient segments of the flat gradient arena, in the order backward completes them (ResNet: layer4, layer3,
 * layer2, layer1 + stem).  `wait_grad_segment` makes `stream` wait (hipStreamWaitEvent) until segment `index` of the
 * most recently enqueued backward is complete, so a data-parallel caller can all-reduce finished ranges under the rest
 * of backward.  Build-side addition (the reference is single-GPU, train_pad_20.py:509).  Plans that report
 * 0 segments are reduced as one range after backward. */
int mmskin_backbone_num_grad_segments(mmskin_backbone_t h, int* count);
int mmskin_backbone_grad_segment(mmskin_backbone_t h, int index, int64_t* offset, int64_t* numel);
int mmskin_backbone_wait_grad_segment(mmskin_backbone_t h, int index, void* stream);
int mmskin_backbone_last_conv_shape(mmskin_backbone_t h, int* C, int* OH, int* OW);
int mmskin_backbone_last_conv_export(mmskin_backbone_t h, const void* workspace, float* x_nchw, void* stream);
int mmskin_backbone_last_conv_grad(mmskin_backbone_t h, const float* dfeatures, const void* workspace, float* dx_nchw,
                                   void* stream);
int mmskin_backbone_backward(mmskin_backbone_t h, const float* dfeatures, const float* params, void* workspace,
                             float* param_grads, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Op-level convolution / batch-norm entry points (same kernels the plan uses; NCHW fp32 at the
 * boundary, converted to NHWC `dtype` inside `workspace`).  Used by the parity tests and by
 * consumers that need a single layer.  conv weight OIHW fp32, no bias (torchvision ResNet convs). */
int64_t mmskin_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad);
int mmskin_conv2d_forward(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, int kh,
                          int kw, int stride, int pad, int dtype, void* workspace, void* stream);
/* dx (may be null) and dw (may be null) from dy [N,Cout,OH,OW] */
int mmskin_conv2d_backward(const float* dy, const float* x, const float* w, float* dx, float* dw, int N, int Cin,
                           int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype, void* workspace,
                           void* stream);
/* bf16 data gradient with the consumer unit's BatchNorm-backward prologue fused into the epilogue (what the ResNet plan launches for every
 * conv -> BatchNorm -> ReLU unit, csrc/conv_gemm.hip profile 3 / csrc/conv3x3_c64.hip): dz = dgrad(dy) * (xc * scale + shift > 0) and
 * the per-row-block partial sums [rows][2][Cin] of dz and dz * xc.  Replaces conv_backward(input) + the ReLU / BatchNorm-backward
 * reductions autograd runs after it (train_pad_20.py:112).  Workspace: mmskin_conv2d_workspace_bytes. */
int mmskin_conv2d_dgrad_fused_rows(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad);
int mmskin_conv2d_dgrad_fused(const float* dy, const float* w, const float* xc, const float* scale, const float* shift, float* dz,
                              float* partial, int* rows_written, int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                              int pad, void* workspace, void* stream);
/* launches of the layer-1 all-taps 3x3 kernel (csrc/conv3x3_c64.hip) since load */
int64_t mmskin_conv3x3_c64_launches(void);
int64_t mmskin_stem7x7_launches(void);   /* ... of the direct 7x7 stem convolution (stem7x7.hip) */
/* timing helper for kernel tuning: average microseconds of the forward conv kernel over `iters` launches
 * on NHWC buffers carved from `workspace` (>= conv2d_workspace_bytes; contents irrelevant) */
double mmskin_conv2d_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                          int iters, void* workspace, void* stream);
double mmskin_conv2d_dgrad_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                                int iters, void* workspace, void* stream);
double mmskin_conv2d_wgrad_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                                int iters, void* workspace, void* stream);
/* number of conv / dgrad launches this process has sent to the 8-phase pipelined kernel (224 / 256-row tiles, csrc/conv_gemm.hip);
 * the parity tests assert through it that the kernel they mean to check is the one that ran */
int64_t mmskin_conv_pipe_launches(void);
/* launches of the ring weight-gradient kernel (wgrad_ring.hip) since load: tests assert the path they mean to cover ran */
int64_t mmskin_wgrad_ring_launches(void);
int64_t mmskin_wgrad3_ring_launches(void);   /* ... of the all-taps 3x3 ring kernel (wgrad3_ring.hip) */
/* Algebraic BatchNorm backward of an expanding 1x1 convolution x = conv(y, w) (csrc/abn.hip; bf16 operands): with
 * dz = cA g + cB x + cC per channel, dy = dz w and dw = dz^T y are computed from g, y, w and the coefficients alone
 * (dy = [g | y][cA w ; w^T diag(cB) w] + cC w, dw = cA (g^T y) + cB (w (y^T y)) + cC (x) colsum(y)): neither dz nor x is read.
 * Replaces autograd's backward through nn.BatchNorm2d + nn.Conv2d of torchvision's Bottleneck.conv3 (train_pad_20.py:112). */
int64_t mmskin_abn_workspace_bytes(int N, int Cw, int C4, int H, int W);
int mmskin_abn_backward(const float* g, const float* y, const float* w, const float* cA, const float* cB, const float* cC, int N, int Cw,
                        int C4, int H, int W, float* dy, float* dw, void* workspace, void* stream);
/* ... with the Gram matrix y^T y and colsum(y) taken by their own launch first and g^T y alone afterwards: the order of the two-pass
 * forward, whose first pass computes them for the statistics below and keeps them for this backward */
int mmskin_abn_backward_kept_gram(const float* g, const float* y, const float* w, const float* cA, const float* cB, const float* cC, int N,
                                  int Cw, int C4, int H, int W, float* dy, float* dw, void* workspace, void* stream);
/* Batch statistics of x = conv1x1(y, w) without x: stat_sum[o] = sum x[.,o] = sum_k w[o][k] colsum(y)[k], stat_sq[o] = sum x[.,o]^2 =
 * w_o^T (y^T y) w_o (bf16 operands, fp32 Gram matrix from the ring weight-gradient kernel, double-precision contraction).  The first
 * pass of the two-pass BatchNorm forward of Bottleneck.conv3 + bn3 (train-mode nn.BatchNorm2d statistics, train_pad_20.py:102);
 * workspace: mmskin_abn_workspace_bytes. */
int mmskin_conv1x1_gram_stats(const float* y, const float* w, int N, int Cw, int C4, int H, int W, float* stat_sum, float* stat_sq,
                              void* workspace, void* stream);
/* training-mode BatchNorm2d + optional ReLU on NCHW fp32 tensors (batch statistics) */
int64_t mmskin_batchnorm_workspace_bytes(int N, int C, int H, int W);
int mmskin_batchnorm_forward(const float* x, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float* y, float* save_mean, float* save_invstd, int N, int C, int H,
                             int W, float eps, float momentum, int relu, int dtype, void* workspace, void* stream);
int mmskin_batchnorm_backward(const float* dy, const float* x, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                              int N, int C, int H, int W, int relu, int dtype, void* workspace, void* stream);
/* the ResNet stem: conv7x7/2 (OIHW [64,3,7,7]) -> BN(train) -> ReLU -> maxpool3x3/2; y [N,64,PH,PW] */
int64_t mmskin_stem_workspace_bytes(int N, int H, int W);
int mmskin_stem_forward(const float* x, const float* w, const float* gamma, const float* beta, float* y, int N, int H,
                        int W, float eps, int dtype, void* workspace, void* stream);
int mmskin_stem_backward(const float* dy, const float* x, const float* w, const float* gamma, const float* beta,
                         float* dw, float* dgamma, float* dbeta, int N, int H, int W, float eps, int dtype,
                         void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fusion-head operators (fp32).  Replace nn.Linear / nn.LayerNorm / nn.MultiheadAttention(L=1) /
 * torch.sigmoid gates / MetaBlock / GatedAlteredResidualBlock pointwise math of
 * multimodalIntraInterModal.py:172-412, metablock.py:27-32, gatedResidualBlock.py:12-17. */
/* y[M,N] = x[M,K] @ w[N,K]^T + b (b may be null) ; relu!=0 applies max(.,0) */
int mmskin_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int relu,
                          void* stream);
/* dy[M,N]; if y_relu is given dy is first masked by (y_relu > 0).  Any of dx[M,K], dw[N,K], db[N] may be
 * null.  dy_scratch[M,N] is required when y_relu is given (holds the masked dy). */
int mmskin_linear_backward(const float* dy, const float* x, const float* w, const float* y_relu, float* dy_scratch,
                           float* dx, float* dw, float* db, int M, int K, int N, void* stream);
/* Backward of h = gelu(x w^T + b) (nn.Linear -> nn.GELU, the first half of timm's Mlp) from dh and the saved pre-activation z [M][N]:
 * gelu'(z) is applied inside the pass that converts the gradient to bf16 for the dgrad / wgrad GEMMs and sums it for db, so the fp32
 * gradient of the pre-activation is never written.  dy_scratch [M][N] is used only off the bf16-operand large-GEMM path. */
int mmskin_linear_gelu_backward(const float* dh, const float* x, const float* w, const float* z, float* dy_scratch, float* dx, float* dw,
                                float* db, int M, int K, int N, void* stream);
/* The bf16-operand large-GEMM path converts x to bf16 for the forward GEMM and again for the backward's weight-gradient GEMM.
 * mmskin_linear_x16_pitch > 0 (the row pitch in elements; 0: this shape / mode makes no such copy) lets a caller keep the forward's
 * copy instead: _forward_keep writes it to x16_keep [M][pitch] (bf16), _backward_keep takes it in x's place (y_relu / z_gelu: the saved
 * output of a fused ReLU or pre-activation of a following GELU, at most one, as in mmskin_linear_backward / _gelu_backward).
 * Same arithmetic as mmskin_linear_forward / _backward: the kept copy is the tensor the scratch conversion would have produced. */
int mmskin_linear_x16_pitch(int M, int K, int N);
int mmskin_linear_forward_keep(const float* x, const float* w, const float* b, const float* res, float* y, void* x16_keep, int M, int K,
                               int N, int relu, void* stream);   /* res (optional, fp32 [M][N]): y = res + act(x w^T + b) */
int mmskin_linear_backward_keep(const float* dy, const void* x16, const float* w, const float* y_relu, const float* z_gelu, float* dy_scratch,
                                float* dx, float* dw, float* db, int M, int K, int N, void* stream);
/* A transformer MLP (timm Mlp: fc1 -> GELU -> fc2) with gradients, without gelu(z) in fp32: mmskin_gelu_forward_bf16 writes
 * h16 [rows][cols_pad] = bf16(gelu(z)) (zero pad columns; cols_pad = mmskin_linear_x16_pitch of the second Linear), and
 * mmskin_linear_forward_x16 runs the second Linear on it; the backward hands h16 to mmskin_linear_backward_keep. */
int mmskin_gelu_forward_bf16(const float* z, void* h16, int64_t rows, int cols, int cols_pad, void* stream);
int mmskin_linear_forward_x16(const void* x16, const float* w, const float* b, const float* res, float* y, int M, int K, int N, int relu,
                              void* stream);                     /* res as above: the block's skip connection in the same pass */
/* y = LN(x)*g + b over the last dim, optional fused ReLU; mean/rstd [M] saved for backward */
int mmskin_layernorm_forward(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd,
                             int M, int N, float eps, int relu, void* stream);
int mmskin_layernorm_backward(const float* dy, const float* x, const float* g, const float* b, const float* mean,
                              const float* rstd, float* dx, float* dg, float* db, int M, int N, int relu, void* stream);
/* out = sigmoid(z) * v                                  (multimodalIntraInterModal.py:219-235) */
int mmskin_sigmoid_gate_forward(const float* z, const float* v, float* out, int64_t n, void* stream);
int mmskin_sigmoid_gate_backward(const float* dout, const float* z, const float* v, float* dz, float* dv, int64_t n,
                                 void* stream);
/* out = g*a + (1-g)*q, g = sigmoid(z)                   (gatedResidualBlock.py:15-16) */
int mmskin_gated_mix_forward(const float* z, const float* a, const float* q, float* out, int64_t n, void* stream);
int mmskin_gated_mix_backward(const float* dout, const float* z, const float* a, const float* q, float* dz, float* da,
                              float* dq, int64_t n, void* stream);
/* out = sigmoid(tanh(V*t1) + t2)                        (metablock.py:31) */
int mmskin_metablock_gate_forward(const float* V, const float* t1, const float* t2, float* out, int64_t n, void* stream);
int mmskin_metablock_gate_backward(const float* dout, const float* V, const float* t1, const float* t2, float* dV,
                                   float* dt1, float* dt2, int64_t n, void* stream);
/* One Adam step (torch.optim.Adam: g += weight_decay p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)) over one flat fp32 range of n elements: parameters, gradient,
 * both moments; step counts from 1; 16-byte aligned pointers.  The optimizer step of train_pad_20.py:54,113 for a parameter arena. */
int mmskin_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int64_t step, void* stream);
/* inverted dropout with a counter-based generator; mask[n] bytes (1 keep / 0 drop) */
int mmskin_dropout_forward(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                           uint64_t offset, void* stream);
/* dropout on attention probabilities [rows][L] (attention_probs_dropout_prob / nn.MultiheadAttention(dropout=p)): the (row, key) generator
 * every attention kernel of the library applies or regenerates (fused forward / backward, one-wave-per-row kernels), so the unfused
 * chain softmax -> dropout -> bmm drops the same probabilities; mask as mmskin_dropout_forward (backward: mmskin_dropout_backward) */
int mmskin_attn_dropout_forward(const float* x, float* y, uint8_t* mask, int64_t n, int L, float p, uint64_t seed, uint64_t offset,
                                void* stream);
int mmskin_dropout_backward(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream);
/* out[M, Na+Nb] = [a | b]; backward splits */
int mmskin_concat2_forward(const float* a, const float* b, float* out, int M, int Na, int Nb, void* stream);
int mmskin_concat2_backward(const float* dout, float* da, float* db, int M, int Na, int Nb, void* stream);
/* softmax attention for the metadata TabTransformer / generic L>1 attention:
 * q,k,v [B,H,L,Dh] -> o [B,H,L,Dh], softmax probs p [B,H,L,L] saved for backward.  drop_p > 0 applies
 * training-mode dropout to the probabilities (nn.MultiheadAttention(dropout=...) inside
 * nn.TransformerEncoderLayer, tab_transformer.py:20-27) with the counter-based generator of
 * mmskin_dropout_forward: backward must be given the same (drop_p, seed, offset). */
int mmskin_attention_forward(const float* q, const float* k, const float* v, float* o, float* p, int B, int H, int L,
                             int Dh, float drop_p, uint64_t seed, uint64_t offset, void* stream);
int mmskin_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* p,
                              float* dq, float* dk, float* dv, int B, int H, int L, int Dh, float drop_p,
                              uint64_t seed, uint64_t offset, void* stream);
/* MD-Net fusion head (multimodalMDNet.py:21-29 MetaNet gate, :47-55 MetaBlock, :94-100 sum + GAP):
 * pooled[nc] = mean_hw(sigmoid(z[nc]) * feat[nc][hw] + sigmoid(tanh(feat[nc][hw] * t1[nc]) + t2[nc])), feat NCHW */
int mmskin_mdnet_fuse_forward(const float* feat, const float* z, const float* t1, const float* t2, float* pooled,
                              int64_t NC, int HW, void* stream);
int mmskin_mdnet_fuse_backward(const float* dpooled, const float* feat, const float* z, const float* t1, const float* t2,
                               float* dfeat /* may be NULL */, float* dz, float* dt1, float* dt2, int64_t NC, int HW,
                               void* stream);
/* Encoder-layer pieces for the HuggingFace text encoders (loadImageModelClassifier.py:170-181, bert-base-uncased):
 * strided batched GEMM  C[b](m,n) = sum_k A[b](m,k) * B[b](n,k)  with A[b](m,k) = a[b*sab + m*sam + k*sak] (likewise B),
 * C[b] = c + b*scb with row pitch ldc -- QK^T, PV and their gradients;
 * row softmax with scale and an additive per-batch key mask (mask_add [batch][L], may be NULL); exact (erf) GELU. */
int mmskin_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int64_t sam, int64_t sak, int64_t sab,
               int64_t sbn, int64_t sbk, int64_t sbb, int64_t ldc, int64_t scb, void* stream);
int mmskin_softmax_forward(const float* x, const float* mask_add, const float* bias /* [rows_per_batch][L] or NULL */,
                           float* y, int64_t rows, int L, int64_t rows_per_batch, float scale,
                           int causal /* GPT-2: key j <= query i */, void* stream);
int mmskin_softmax_backward(const float* dy, const float* y, float* dx, int64_t rows, int L, float scale, void* stream);
/* pieces of the timm transformer blocks (BEiT: LayerScale residuals, mean pooling over patch tokens, bias gradients) */
int mmskin_colsum(const float* x, float* out, int M, int N, void* stream);
int mmskin_scale_add_forward(const float* x, const float* b, const float* gamma, float* y, int64_t n, int C, void* stream);
int mmskin_scale_mul(const float* dy, const float* v, float* out, int64_t n, int C, int per_channel, void* stream);
int mmskin_token_mean_forward(const float* x, float* out, int B, int L, int E, int start, void* stream);
int mmskin_token_mean_backward(const float* dout, float* dx, int B, int L, int E, int start, void* stream);
/* Fused attention, bf16 MFMA with fp32 accumulation / softmax / I/O:  o = dropout(softmax(q k^T * scale + bias + mask)) v  in one
 * kernel; the [B, H, L, L] scores never reach memory.  Replaces the QK^T GEMM -> softmax -> dropout -> PV GEMM chain of the
 * transformer encoders (timm ViT / BEiT blocks, transformers' BertSelfAttention / GPT2Attention; loadImageModelClassifier.py
 * :117-121, :170-181) in bf16-operand mode.  q, k, v, o: io_dtype (MMSKIN_F32 or MMSKIN_BF16), element (b, h, l, d) at index b*sb + h*sh + l*sl + d, the twelve
 * ELEMENT strides given as strides12 = {q_sb, q_sh, q_sl, k_*, v_*, o_*} (q / k / v: multiples of 16 bytes; pointers 16-byte aligned), so the output of a
 * fused qkv Linear is read in place.  mask_add [B][L] additive key mask or NULL; bias [H][L][L] additive score bias or NULL
 * (BEiT's relative-position bias); causal != 0: key j > query i masked; drop_p: dropout on the probabilities with the library's
 * counter-based generator on the element index of the [B, H, L, L] tensor (seed, offset as mmskin_dropout_forward);
 * lse (optional) [B][H][L] = log-sum-exp of the scaled, biased scores.  Dh in {32, 64}. */
int mmskin_flash_attention_forward(const void* q, const void* k, const void* v, const float* mask_add, const float* bias,
                                   void* o, float* lse, int B, int H, int L, int Dh, const int64_t* strides12, int io_dtype,
                                   float scale, int causal, float drop_p, uint64_t seed, uint64_t offset, void* stream);
/* Backward of the fused attention for TRAINABLE transformer blocks in bf16-operand mode (csrc/flash_attn_bwd.hip): dq, dk, dv from
 * q, k, v, o, d(o) and the forward's lse, the probabilities recomputed tile by tile (no [B, H, L, L] tensor is kept).  All tensors fp32;
 * strides15 = element strides (batch, head, token) of q, k, v, o (= dout) and of dq / dk / dv (shared), last dims contiguous; bias
 * [H][L][L] needs its transpose biasT [H][key][query] beside it; delta [B][H][L] is scratch; ds_out (optional) [B][H][L][L] receives
 * dS = d loss / d(scaled, biased score) so the caller can sum the bias gradient over the batch; dropout regenerated from (seed, offset)
 * exactly as the forward drew it.  Replaces autograd through timm / transformers attention (loadImageModelClassifier.py:117-131,170-181). */
int mmskin_flash_attention_backward(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                                    const float* mask_add, const float* bias, const float* biasT, float* delta, float* dq, float* dk,
                                    float* dv, float* ds_out, int B, int H, int L, int Dh, const int64_t* strides15, float scale,
                                    int causal, float drop_p, uint64_t seed, uint64_t offset, void* stream);
/* Inference lane of the transformer encoders in bf16-operand mode (no gradient flows: frozen encoders, evaluation): activations pass
 * between layers as bf16, so no fp32 <-> bf16 conversion passes run.  linear_forward_ex: x [M][K] and y [M][N] are fp32
 * (dtype MMSKIN_F32) or bf16 (MMSKIN_BF16) independently; w [N][K], b [N] fp32; act 0 none / 1 ReLU / 2 exact GELU, fused into the GEMM
 * epilogue; shapes off the large-GEMM path are computed through the fp32 entry point.  layernorm_forward_mixed: nn.LayerNorm with the
 * result written as fp32 (y_f32) and / or bf16 (y_bf16); N % 4 == 0, N <= 2048. */
int mmskin_linear_forward_ex(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, int M, int K, int N,
                             int act, void* stream);
/* Small-sequence attention, one wave per (batch, head): L <= 64, Dh 32 or 64, fp32, softmax(q k^T * scale) v with optional dropout on the
 * probabilities (generator of mmskin_dropout_forward on element ((b*H + h)*L + i)*L + j).  q / k / v -- and dq / dk / dv -- are addressed by
 * the (batch, head, token) ELEMENT strides qkv_strides[3], o / dO by o_strides[3] (rows 16-byte aligned, Dh contiguous), so the packed
 * [B, L, 3, H, Dh] output of a fused qkv Linear is read in place and o is written token-major.  lse [B*H][L] (row log-sum-exp) is all the
 * backward needs besides q, k, v, o: no probability tensor is stored.  Replaces timm's window / global Attention.forward and
 * nn.MultiheadAttention's core for those shapes with gradients (DaViT WindowAttention: 49 tokens, Dh 32; hip_davit.py). */
int mmskin_attention_rows_forward(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L, int Dh,
                                  const int64_t* qkv_strides, const int64_t* o_strides, float scale, float drop_p, uint64_t seed,
                                  uint64_t offset, void* stream);
int mmskin_attention_rows_backward(const float* dO, const float* q, const float* k, const float* v, const float* o, const float* lse,
                                   float* dq, float* dk, float* dv, int B, int H, int L, int Dh, const int64_t* qkv_strides,
                                   const int64_t* o_strides, float scale, float drop_p, uint64_t seed, uint64_t offset, void* stream);
/* Window attention on an image-major token grid [B][nwy*ws][nwx*ws] (ws*ws <= 64, Dh 32 or 64): window (image, wy, wx) attends over its
 * ws*ws tokens WHERE THEY SIT -- the kernels of mmskin_attention_rows_* with the window -> token map in their addressing, so timm's
 * window_partition / window_reverse (davit.py SpatialBlock.forward; reached through loadImageModelClassifier.py:117-131) are index
 * arithmetic, not copies.  q_tok / q_head: element strides between tokens / heads of q, k, v (and dq, dk, dv); o_tok / o_head of o
 * (and dO).  lse [B*nwy*nwx][H][ws*ws].  Dropout as mmskin_attention_rows_forward with batch = window index. */
int mmskin_window_attention_forward(const float* q, const float* k, const float* v, float* o, float* lse, int B, int nwy, int nwx, int ws,
                                    int H, int Dh, int64_t q_tok, int64_t q_head, int64_t o_tok, int64_t o_head, float scale,
                                    float drop_p, uint64_t seed, uint64_t offset, void* stream);
int mmskin_window_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* o, const float* lse,
                                     float* dq, float* dk, float* dv, int B, int nwy, int nwx, int ws, int H, int Dh, int64_t q_tok,
                                     int64_t q_head, int64_t o_tok, int64_t o_head, float scale, float drop_p, uint64_t seed,
                                     uint64_t offset, void* stream);
/* Channel attention of DaViT's ChannelBlock (timm davit.py ChannelAttention.forward; loadImageModelClassifier.py:117-131), fp32, on
 * token-major operands: per (batch, group of Dh = 32 channels)  A = softmax(scale * q^T k) [32 x 32] (the reduction runs over the N tokens),
 * x[n][i] = sum_j A[i][j] v[n][j].  q / k / v (and dq / dk / dv): element strides q_tok between tokens and q_b between batches, group g
 * at channel offset 32 g -- the packed [B, N, 3, G, 32] output of a fused qkv Linear is read in place; x / dO likewise with o_tok / o_b.
 * attn [B*G][32][32] is written by the forward and read by the backward (may be null when no backward follows).
 * scratch: mmskin_channel_attention_scratch_floats(B, G, N) floats (0 for N <= 256: may be null) -- per-chunk partial products. */
int64_t mmskin_channel_attention_scratch_floats(int B, int G, int N);
int mmskin_channel_attention_forward(const float* q, const float* k, const float* v, float* x, float* attn, float* scratch, int B, int G,
                                     int N, int Dh, int64_t q_tok, int64_t q_b, int64_t o_tok, int64_t o_b, float scale, void* stream);
int mmskin_channel_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* attn, float* dq,
                                      float* dk, float* dv, float* scratch, int B, int G, int N, int Dh, int64_t q_tok, int64_t q_b,
                                      int64_t o_tok, int64_t o_b, float scale, void* stream);
/* linear_lane: the lane Linear with a frozen transformer block's elementwise tail fused into the GEMM epilogue,
 *   y = residual + gamma * dropout(act(x w^T + b))     (residual fp32 [M][N], gamma fp32 [N], dropout: the generator of mmskin_dropout_forward
 *   on element row * N + column; each optional).  w is fp32 or an already-converted bf16 copy (w_dtype).  Replaces the nn.Linear ->
 *   nn.Dropout -> residual add chain of transformers' BertSelfOutput / BertOutput and timm's `x + gamma * proj(...)` (hip_bert.py, hip_beit.py).
 *   Only for shapes on the large bf16 GEMM path (rows >= 2048, 64-multiple widths, bf16-operand mode): MMSKIN_ERR_ARG otherwise. */
int mmskin_linear_lane(const void* x, int x_dtype, const void* w, int w_dtype, const float* b, const float* gamma,
                       const float* residual, float drop_p, uint64_t seed, uint64_t offset, void* y, int y_dtype, int M, int K,
                       int N, int act, void* stream);
int mmskin_layernorm_forward_mixed(const float* x, const float* g, const float* b, float* y_f32, void* y_bf16, int M, int N,
                                   float eps, void* stream);
/* Operand type of the large Linear GEMMs (rows >= 2048, 64-multiple widths: the transformer backbones' and BERT's
 * projections): MMSKIN_F32 = exact-f32 MFMA (default, parity mode), MMSKIN_BF16 = bf16 operands with fp32 accumulation
 * (BASELINE configs[3] is quoted in bf16); in that mode widths that are only multiples of 8 (DaViT's 96 / 288) run on the same kernels
 * through zero-padded bf16 operand copies (mmskin_linear_forward / _backward; exact zeros from the pad columns).  Process-wide; the
 * environment variable MMSKIN_LINEAR_DTYPE sets the initial value. */
int mmskin_set_linear_dtype(int dtype);
int mmskin_get_linear_dtype(void);
/* y = a + b, b broadcast over the leading dimension when nb < n (residual sums, position embeddings) */
int mmskin_add(const float* a, const float* b, float* y, int64_t n, int64_t nb, void* stream);
int mmskin_gelu_forward(const float* x, float* y, int64_t n, void* stream);
int mmskin_gelu_backward(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int mmskin_gelu_tanh_forward(const float* x, float* y, int64_t n, void* stream);      /* GPT-2 "gelu_new" */
int mmskin_gelu_tanh_backward(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* fp32 NHWC depthwise 3x3 convolution (stride 1, pad 1) with its gradients -- ConvPosEnc of timm's DaViT blocks */
int64_t mmskin_dwconv3_scratch_floats(int N, int H, int W, int C);
int mmskin_dwconv3_forward(const float* x, const float* w, float* w_stage, float* y, int N, int H, int W, int C, void* stream);
int mmskin_dwconv3_backward(const float* dy, const float* x, const float* w, float* w_stage, float* scratch, float* dx, float* dw,
                            int N, int H, int W, int C, void* stream);
/* DaViT's convolutional position encoding (timm davit.py ConvPosEnc.forward: x + proj(x), proj = depthwise 3x3 with bias) in one pass each
 * way: y = x + dwconv3(x, w) + b; backward dx = dy + dgrad(dy), dw, db = sum(dy) (from the weight-gradient pass: db needs dw).
 * Staging / scratch as mmskin_dwconv3_*. */
int mmskin_conv_pos_enc_forward(const float* x, const float* w, const float* b, float* w_stage, float* y, int N, int H, int W, int C,
                                void* stream);
int mmskin_conv_pos_enc_backward(const float* dy, const float* x, const float* w, float* w_stage, float* scratch, float* dx, float* dw,
                                 float* db, int N, int H, int W, int C, void* stream);
/* embedding gather for categorical metadata columns: table [ncols, card, E]; ids [B, ncols] int64 */
int mmskin_embedding_forward(const float* table, const int64_t* ids, float* out, int B, int ncols, int card, int E,
                             void* stream);
int mmskin_embedding_backward(const float* dout, const int64_t* ids, float* dtable, int B, int ncols, int card, int E,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input side of the path: what the reference's Dataset does on the host before `model(image, metadata)`.
 * resize_u8: the val/test transform's A.Resize(h, w) (skinLesionDatasets.py:116-120 = cv2.resize INTER_LINEAR on the
 * decoded uint8 HWC image): [N][src_h][src_w][3] -> [N][dst_h][dst_w][3], uint8; Normalize + ToTensor then run inside
 * mmskin_backbone_forward_u8.
 * metadata_encode: OneHotEncoder(handle_unknown='ignore') + StandardScaler transform (skinLesionDatasets.py:133-183):
 * codes int32 [batch][n_cat] = index of the value among the column's fitted (sorted) categories, -1 = unseen;
 * col_offset int32 [n_cat + 1] = first one-hot slot of every column (col_offset[n_cat] = onehot_width); numeric fp32
 * [batch][n_num] (NaN = missing -> nan_fill, the reference's fillna(-1)); out fp32 [batch][onehot_width + n_num] =
 * one-hot blocks | (numeric - mean) / scale -- the tensor the reference hands to text_fc. */
/* Patch extraction for the patch-embedding GEMMs of the timm backbones (PatchEmbed 16x16/16 of ViT / BEiT, DaViT's 7x7/4
 * stem and 2x2/2 downsample convolutions; loadImageModelClassifier.py:117-121): cols [N*OH*OW][C*k*k], columns ordered
 * (c, ky, kx) like conv.weight.flatten(1), zeros for padding taps; x is fp32 NCHW (channels_last = 0) or NHWC (= 1).
 * backward: dx (same layout as x, every element written) = the transpose (sum over the windows containing a pixel). */
int mmskin_im2col_forward(const float* x, int N, int C, int H, int W, int k, int stride, int pad, int channels_last, float* cols,
                          void* stream);
int mmskin_im2col_backward(const float* dcols, int N, int C, int H, int W, int k, int stride, int pad, int channels_last, float* dx,
                           void* stream);
int mmskin_resize_u8(const uint8_t* src_nhwc, int N, int src_h, int src_w, uint8_t* dst_nhwc, int dst_h, int dst_w,
                     void* stream);
int mmskin_metadata_encode(const int32_t* codes, int n_cat, const int32_t* col_offset, int onehot_width, const float* numeric,
                           int n_num, const float* mean, const float* scale, float nan_fill, float* out, int batch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * `custom-cnn` image encoder pieces (loadImageModelClassifier.py:50-60): small direct kernels for
 * shapes the MFMA implicit GEMM does not cover (Cin=3, Cout=16).  NCHW fp32. */
int mmskin_direct_conv2d_forward(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                                 int Cout, int kh, int kw, int stride, int pad, int relu, void* stream);
int mmskin_direct_conv2d_backward(const float* dy, const float* x, const float* y_relu, float* dw, float* db, int N,
                                  int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, void* stream);
/* maxpool(k, stride=k) followed by global average pool: y[N,C]; idx for backward */
int mmskin_pool_gap_forward(const float* x, float* y, int32_t* idx, int N, int C, int H, int W, int k, void* stream);
int mmskin_pool_gap_backward(const float* dy, const int32_t* idx, float* dx, int N, int C, int H, int W, int k,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSKIN_H */
