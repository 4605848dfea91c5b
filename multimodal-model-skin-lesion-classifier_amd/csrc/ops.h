// HBM-bound NHWC kernels around the conv GEMMs: BatchNorm (train/eval, fwd/bwd), ReLU/residual,
// stem packing + max-pool, global average pool, weight staging.  All loads/stores are 16 bytes per
// lane; reductions are deterministic (per-block partial slabs + a finalize kernel, no atomics).
#pragma once
#include "common.h"

// ---- BatchNorm forward
#ifdef __HIPCC__
// channel c's training-mode BatchNorm coefficients from its sum and sum of squares (double), running statistics updated
__device__ __forceinline__ void bn_fwd_coeffs(int c, double s, double q, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                              float* save_mean, float* save_invstd) {
  double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (save_mean) { save_mean[c] = (float)mean; save_invstd[c] = invstd; }
  if (running_mean) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}
#endif
// Reduce conv-epilogue partials [nrows][C] -> batch mean/var -> scale/shift (+ running stats).
int bn_finalize(const float* stat_sum, const float* stat_sq, int nrows, int C, double count,
                const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                double* scratch, hipStream_t st);
// scratch for the two finalize calls: 2 * 64 * C doubles (may be null: single-stage reduction)
static inline size_t bn_reduce_scratch_bytes(int C) { return (size_t)2 * 64 * C * sizeof(double); }
// in[nrows][cols] -> out[G][cols] (and in1 -> out[G..2G) when given)
template <typename OUT>
int partial_reduce(const float* in0, const float* in1, int nrows, int cols, int G, OUT* out, hipStream_t st);
int bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, float* scale, float* shift, hipStream_t st);
// y = [relu](x*scale[c] + shift[c] [+ res | + res*rscale[c] + rshift[c]]);  rows*C elements
template <typename T>
int bn_apply(const T* x, const T* res, const float* scale, const float* shift, const float* rscale,
             const float* rshift, T* y, size_t rows, int C, bool relu, hipStream_t st,
             uint8_t* mask_bits = nullptr,    // optional: one byte per 16-byte chunk, bit e = (y[e] > 0)
             float relu_cap = 0.f);           // > 0: ReLU6-style clamp min(relu(.), cap) (MobileNet); < 0 (with relu): SiLU
// Column partial sums of an arbitrary NHWC tensor (used where no conv epilogue produced them).
template <typename T>
int column_stats(const T* x, size_t rows, int C, float* stat_sum, float* stat_sq, int* nrows_out,
                 hipStream_t st);
int column_stats_rows(size_t rows, int C);

// ---- BatchNorm backward
enum { MASK_NONE = 0, MASK_FROM_X = 1, MASK_FROM_Y = 2, MASK_FROM_Y6 = 3,   // Y6: 0 < y < 6 (ReLU6)
       MASK_SILU_X = 4 };   // dz = dy * silu'(x*scale + shift)
int bn_bwd_partial_rows(size_t rows, int C);  // upper bound over dtypes (for sizing only)
// partial[blk][0][C] = sum dz, partial[blk][1][C] = sum dz*x   with dz = dy * mask
template <typename T>
int bn_bwd_reduce(const T* dy, const T* x, const T* ymask, const float* scale, const float* shift,
                  int mask_mode, size_t rows, int C, float* partial, int* nrows_out, hipStream_t st);
// -> dgamma, dbeta (may be null) and dx = cA*dz + cB*x + cC coefficient vectors
// n_grad (default C): dgamma / dbeta are written for channels < n_grad only (zero-padded channel tails)
int bn_bwd_finalize(const float* partial, int nrows, int C, double count, const float* gamma,
                    const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                    float* cA, float* cB, float* cC, double* scratch, hipStream_t st, int n_grad = -1,
                    bool accumulate_bc = false,    // accumulate_bc: cB / cC are ADDED to (running sums over the consumers of one input)
                    const float* sum_dz_x = nullptr);   // [C]: use this as sum dz*x instead of the partial rows' second half (abn.hip, second phase)
// dx = cA*dz + cB*x + cC ; optionally also writes dz (masked dy) to dz_out
template <typename T>
int bn_bwd_apply(const T* dy, const T* x, const T* ymask, const float* scale, const float* shift,
                 int mask_mode, const float* cA, const float* cB, const float* cC, T* dx, T* dz_out,
                 size_t rows, int C, hipStream_t st);

// ---- stem
// NCHW fp32 image -> zero-padded NHWC4 T image [N][Hp][Wp][4] (3 px top/left border)
template <typename T>
int stem_pack(const float* img, int N, int H, int W, int Hp, int Wp, T* img4, hipStream_t st);
// same from a uint8 NHWC image (what the DataLoader decodes), normalised on the fly:
// v = (u8 / 255 - mean[c]) / std[c]  (A.Normalize + ToTensorV2, skinLesionDatasets.py:29,111-119); norm6 = mean rgb | std rgb (host)
template <typename T>
int stem_pack_u8(const uint8_t* img_nhwc, int N, int H, int W, int Hp, int Wp, const float* norm6, T* img4,
                 hipStream_t st);
// y = maxpool3x3s2p1(relu(x*scale+shift)); idx = argmax tap (first max, row-major), 0..8
template <typename T>
int stem_bn_relu_pool(const T* x, const float* scale, const float* shift, int N, int H, int W, int C,
                      T* y, uint8_t* idx, hipStream_t st);

// Stem backward without the full-resolution un-pooled gradient: dz = (sum of dpool over the pooling windows whose argmax is (h,w))[c] * (x*scale+shift > 0) is
// recomputed from the pooled gradient + argmax bytes inside the BatchNorm-backward reduce and apply passes
// (saves one 411 MB write and two reads of it at batch 256).
// the same partial sums from the pooled tensors alone (dpool, the pooled forward output ypool): x at a window's argmax is (y - shift) / scale
template <typename T>
int stem_pool_bwd_sums(const T* dpool, const T* ypool, const uint8_t* idx, const T* x, const float* scale, const float* shift, int N, int H, int W,
                       int C, float* partial, int* nrows_out, hipStream_t st);
template <typename T>
int stem_pool_bn_bwd_reduce(const T* dpool, const uint8_t* idx, const T* x, const float* scale, const float* shift, int N,
                            int H, int W, int C, float* partial, int* nrows_out, hipStream_t st);
template <typename T>
int stem_pool_bn_bwd_apply(const T* dpool, const uint8_t* idx, const T* x, const float* scale, const float* shift,
                           const float* cA, const float* cB, const float* cC, int N, int H, int W, int C, T* dx, hipStream_t st);

// ---- global average pool
template <typename T>
int avgpool_fwd(const T* x, int N, int HW, int C, float* feat, hipStream_t st);
template <typename T>
int avgpool_bwd(const float* dfeat, int N, int HW, int C, T* dx, hipStream_t st);

// ---- weight staging (fp32 OIHW master weights -> T, K-contiguous GEMM operands)
struct StageDesc {
  int64_t src_off;   // element offset of the OIHW fp32 weight in the flat param buffer
  int64_t fwd_off;   // element offset in the forward staging buffer  [Cout][taps][Cin]
  int64_t dgrad_off; // element offset in the dgrad staging buffer    [Cin][taps][Cout]
  int Cout, Cin, taps;
  int stem;          // 1: conv1 7x7 -> virtual [64][8][32] (fwd only)
  int Cout_pad, Cin_pad;   // staged dims (0 = same as Cout / Cin): extra rows / channels are zero
  // BatchNorm that follows the conv (inference folding): flat param / buffer offsets, coefficient slot in the workspace
  int64_t bn_g_off, bn_b_off, bn_rm_off, bn_rv_off, coef_off;
  int has_bn;
};
// fold_buffers != nullptr (inference): the forward-staged weights of every layer with has_bn are scaled by
// gamma / sqrt(running_var + eps) per output channel, so the conv epilogue only adds the shift
template <typename T>
int stage_weights(const StageDesc* table_dev, int nlayers, int max_elems, const float* params, T* wfwd,
                  T* wdgrad, bool need_dgrad, hipStream_t st, const float* fold_buffers = nullptr, float eps = 1e-5f);
// eval-mode scale / shift of every table layer with has_bn in ONE launch: coef (floats at ws + coef_off) = scale | shift
int bn_eval_table(const StageDesc* table_dev, int nlayers, int maxC, const float* params, const float* buffers,
                  unsigned char* ws, float eps, hipStream_t st);
// dwv[64][8][32] -> OIHW [64][3][7][7]
int stem_wgrad_unpack(const float* dwv, float* dw, hipStream_t st);

// ---- channel-slice kernels for concatenating backbones (DenseNet): `cat` is [rows][pitch]
// out[r][0..Cp) = c < C ? f(in[r*pitch + c]) : 0 with f = relu(x*scale[c]+shift[c]) when scale is given,
// else identity.  `in` may point at a channel offset inside a row.
template <typename T>
int slice_pack(const T* in, int pitch, int C, int Cp, size_t rows, const float* scale, const float* shift, T* out,
               hipStream_t st);
// dst[r*pitch + c] = src[r*srcC + c] for c < C  (dst may point at a channel offset inside a row)
template <typename T>
int slice_scatter(const T* src, int srcC, int C, T* dst, int pitch, size_t rows, hipStream_t st);
// dcat[r*pitch + c] (+)= cA[c]*dz[r*Cp + c] + cB[c]*x[r*pitch + c] + cC[c]   for c < C
template <typename T>
int slice_bn_bwd_accumulate(T* dcat, const T* x, int pitch, int C, const T* dz, int Cp, const float* cA,
                            const float* cB, const float* cC, size_t rows, hipStream_t st);
// deferred form (DenseNet): the x / constant terms of all consumers of a channel are added once, when its gradient is consumed
template <typename T>
int slice_accumulate_scaled(T* dcat, int pitch, int C, const T* dz, int Cp, const float* cA, size_t rows, hipStream_t st);
template <typename T>
int slice_pack_deferred(const T* d, const T* x, int pitch, int C, int Cp, size_t rows, const float* sB, const float* sC, T* out, hipStream_t st);
template <typename T>
int slice_affine_inplace(T* d, const T* x, int pitch, int C, size_t rows, const float* sB, const float* sC, hipStream_t st);
// column partial sums of x[r*pitch + c], c < C -> stat_sum/stat_sq [nrows][C]
template <typename T>
int slice_stats(const T* x, int pitch, int C, size_t rows, float* stat_sum, float* stat_sq, int* nrows_out,
                hipStream_t st);
// partial slabs [nrows][stride] (first C columns used) -> batch mean / biased variance
int bn_table_finalize(const float* stat_sum, const float* stat_sq, int nrows, int stride, int C, double count,
                      float* mean, float* var, double* scratch, hipStream_t st);
// BatchNorm coefficients for a consumer of table channels [0, C): coef = scale|shift|mean|invstd|gamma,
// each Cp long with zeros beyond C.  training: batch stats from the table (+ running-stat update);
// eval: running stats.
int bn_coef_from_table(const float* mean_tab, const float* var_tab, int C, int Cp, const float* gamma,
                       const float* beta, float eps, float momentum, double count, float* running_mean,
                       float* running_var, bool training, float* coef, hipStream_t st);
// 2x2 stride-2 average pool of compact [N][H][W][C] into dst rows of `pitch` channels; and its backward
template <typename T>
int avgpool2_fwd(const T* x, int N, int H, int W, int C, T* dst, int pitch, hipStream_t st);
template <typename T>
int avgpool2_bwd(const T* dpool, int pitch, int N, int H, int W, int C, T* dx, hipStream_t st);

// ---- VGG pieces (conv + bias + ReLU stacks with 2x2 max-pools, torchvision vgg16 `features` + `avgpool`)
// image (fp32 NCHW, or uint8 NHWC with norm6 != nullptr) -> zero-bordered NHWC8 [N][H+2][Wp][8], pixel (h,w) at (h+1, w+1)
template <typename T>
int pack_nhwc8(const void* img, const float* norm6, int N, int H, int W, int Hp, int Wp, T* out, hipStream_t st);
// [64][3][3][3] OIHW fp32 -> virtual-conv operand [64][4][32]: (o, r, s*8 + c); tap 3 and unused slots stay zero
template <typename T>
int vgg_stage_first(const float* w, T* wv, hipStream_t st, int cout = 64);   // cout <= 64 real output channels
int vgg_wgrad_unpack_first(const float* dwv, float* dw, hipStream_t st, int cout = 64);
// 2x2 stride-2 max pool (floor), idx = argmax tap 0..3 per element (first max, row-major)
template <typename T>
int maxpool2_fwd(const T* x, int N, int H, int W, int C, T* y, uint8_t* idx, hipStream_t st);
// dz[n][h][w][c] = dpool at the window's argmax where the (post-ReLU) activation y is > 0, else 0
template <typename T>
int maxpool2_bwd_relu(const T* dpool, const uint8_t* idx, const T* y, int N, int H, int W, int C, T* dz, hipStream_t st);
// AdaptiveAvgPool2d(OH, OW): NHWC T -> NCHW fp32, and its backward
template <typename T>
int adaptive_avgpool_fwd(const T* x, int N, int H, int W, int C, int OH, int OW, float* out_nchw, hipStream_t st);
template <typename T>
int adaptive_avgpool_bwd(const float* dout_nchw, int N, int H, int W, int C, int OH, int OW, T* dx, hipStream_t st);
// db[c] = sum over rows of partial[r*stride + c]   (conv bias gradient from dgrad-epilogue / column_stats partials)
int bias_grad_finalize(const float* partial, int nrows, int stride, int C, float* db, double* scratch, hipStream_t st);

// Grad-CAM: dx[n][c][hw] (NCHW fp32) = dfeat[n][c] / HW * (y[n][hw][c] > 0) * scale[c]  -- the gradient of the pooled
// features w.r.t. the raw output of the last conv under eval-mode BatchNorm (y = relu(x*scale + shift + skip), NHWC T)
template <typename T>
int gap_relu_bn_grad(const float* dfeat, const T* y, const float* scale, int N, int HW, int C, float* dx_nchw, hipStream_t st);

// ---- depthwise 3x3 convolution (pad 1, stride 1 or 2) on NHWC T with C a multiple of the chunk width (MobileNet)
// w_tc: staged weights [9][C] (tap-major) in T
// ksize 3 or 5 (EfficientNet), padding ksize/2; staged weights [ksize*ksize][C]
template <typename T>
int dw_stage_weights(const float* w_oihw, int C, int Cp, T* w_tc, hipStream_t st, int ksize = 3);
template <typename T>
int dwconv3_fwd(const T* in, const T* w_tc, int N, int H, int W, int C, int stride, T* out, hipStream_t st, int ksize = 3,
                const float* bias = nullptr, bool residual = false);   // residual (stride 1): out = in + conv(in) + bias
template <typename T>
int dwconv3_dgrad(const T* dout, const T* w_tc, int N, int H, int W, int C, int stride, T* din, hipStream_t st, int ksize = 3,
                  bool residual = false);                              // residual: din = dout + dgrad(dout)
// dw[c][tap] (OIHW [C][1][3][3], first Cv channels written) = sum over output pixels of dout * shifted input;
// partial: scratch of dwconv3_wgrad_partial_floats() floats
size_t dwconv3_wgrad_partial_floats(int N, int H, int W, int C, int stride, int ksize = 3);
template <typename T>
int dwconv3_wgrad(const T* dout, const T* in, int N, int H, int W, int C, int stride, float* partial, float* dw,
                  int Cv, hipStream_t st, int ksize = 3, float* db = nullptr);   // db (3x3 / stride 1): sum of dout per channel, same pass

// ---- squeeze-excitation and stochastic depth (EfficientNet MBConv)
// y[n][hw][c] = x[n][hw][c] * gate[n][c]
template <typename T>
int se_scale_fwd(const T* x, const float* gate, int N, int HW, int C, T* y, hipStream_t st);
// out[n][c] = scale * sum_hw a[n][hw][c] * (b ? b[n][hw][c] : 1): block-parallel over the pixels (the per-thread loops of
// avgpool_fwd / se_dgate are sized for 7x7 maps; squeeze-excitation pools 112x112 ones)
template <typename T>
int gap_reduce(const T* a, const T* b, int N, int HW, int C, float scale, float* out, hipStream_t st);
// dgate[n][c] = sum_hw dy * x
template <typename T>
int se_dgate(const T* dy, const T* x, int N, int HW, int C, float* dgate, hipStream_t st);
// dx = dy * gate + dpool[n][c] / HW   (dpool = gradient that reached the squeezed (average-pooled) input)
template <typename T>
int se_dx(const T* dy, const float* gate, const float* dpool, int N, int HW, int C, T* dx, hipStream_t st);
// small fp32 vectors: mode 0 silu, 1 sigmoid (forward);  backward: dz = dout * f'(z)
int ew_act_fwd(const float* z, float* out, int64_t n, int mode, hipStream_t st);
int ew_act_bwd(const float* dout, const float* z, float* dz, int64_t n, int mode, hipStream_t st);
// fp32 matrix padding helpers: dst[r][c] = (r < rows && c < cols) ? src[r*cols + c] : 0 for a rows_p x cols_p dst
int pad_matrix(const float* src, int rows, int cols, int rows_p, int cols_p, float* dst, hipStream_t st);
// stochastic depth (row mode): y = branch * mask[n] + res;  out = dy * mask[n]
template <typename T>
int sd_residual_add(const T* branch, const T* res, const float* mask, int N, size_t per_sample, T* y, hipStream_t st);
// out = a + b  (n elements, multiple of the chunk width)
template <typename T>
int ew_add(const T* a, const T* b, T* out, size_t n, hipStream_t st);
template <typename T>
int sd_row_scale(const T* dy, const float* mask, int N, size_t per_sample, T* out, hipStream_t st);

// ---- layout converters used by the op-level C ABI (tests / small tensors)
template <typename T>
int nchw_to_nhwc(const float* src, int N, int C, int H, int W, T* dst, hipStream_t st);
template <typename T>
int nhwc_to_nchw(const T* src, int N, int C, int H, int W, float* dst, hipStream_t st);
