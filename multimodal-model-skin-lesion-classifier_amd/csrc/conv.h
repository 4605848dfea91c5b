// Implicit-GEMM convolution on NHWC activations: host-side descriptors + launchers.
#pragma once
#include "common.h"

// One conv layer (NHWC activations, square or rectangular kernel, symmetric padding).
struct ConvShape {
  int N, H, W, Cin, Cout, kh, kw, stride, pad;
  int OH() const { return (H + 2 * pad - kh) / stride + 1; }
  int OW() const { return (W + 2 * pad - kw) / stride + 1; }
};

#define MMSKIN_MAX_TAPS 12
// A "tap class": a set of output pixels that all gather from the same list of taps.
// Forward conv has one class (all kh*kw taps).  Strided dgrad has stride*stride classes, one per
// output-pixel parity, each with only the taps that land on an integer source pixel.
struct TapClass {
  int a_dim, b_dim;    // class pixel grid (rows of this class = N * a_dim * b_dim)
  int ph, pw;          // output pixel = (a*OS + ph, b*OS + pw)
  int ntaps;
  int mblk_start;      // first row-block index of this class in the launch
  int rows;
  int pad_;
  int8_t offy[MMSKIN_MAX_TAPS], offx[MMSKIN_MAX_TAPS], wtap[MMSKIN_MAX_TAPS];
};

struct ConvGemmArgs {
  const void* in;      // gather source  [N][IH][IW][Cpitch]
  const void* w;       // staged weights [Cout][wtaps][C]   (K contiguous)
  void* out;           // [N][OHf][OWf][Cout]
  const void* addend;  // optional, same layout as out: out = acc + addend (may alias out)
  float* stat_sum;     // optional per-row-block partial column sums   [total_mblk][stat_stride]
  float* stat_sq;      // optional per-row-block partial: sum v*v, or sum v*ep_x when ep_x is set
  // ---- fused BatchNorm-backward prologue of the CONSUMER of `out` (dgrad launches only):
  // v = acc + addend; v *= mask; out = v; stat_sum += v; stat_sq += v*ep_x; stat_b_sq += v*ep_x2
  const void* ep_mask_y;   // mask = (ep_mask_y > 0)      (ReLU after a residual add: mask from the block output)
  const uint8_t* ep_mask_bits;  // or: the same mask as one byte per 16-byte chunk (written by bn_apply)
  const void* ep_x;        // raw BN input x of the consumer unit (same shape as out)
  int ep_x_pitch;          // 0: ep_x rows are Cout elements apart (compact); else the row pitch of ep_x in elements
  const float* ep_scale;   // with ep_shift: mask = (ep_x*scale + shift > 0)   (plain BN+ReLU)
  const float* ep_shift;
  const float* ep_bias;    // inference: per-output-channel constant added to acc (+ addend) -- the folded BatchNorm shift
  int ep_relu;             // 1: ReLU on the stored value (folded-BN inference, Linear + ReLU); 2: exact GELU (transformer MLPs)
  float* out_f32;          // light-epilogue launches only: store the tile as fp32 HERE instead of T at `out` (Linear ops at the fp32 boundary)
  // ---- transformer-residual epilogue (profile 6; Linear layers of frozen encoders): out_f32 = res_f32 + gamma * dropout(act(acc + bias))
  const float* ep_gamma;   // optional per-output-channel multiplier (BEiT / DaViT layer scale)
  const float* res_f32;    // optional fp32 residual stream, [rows][Cout]
  float ep_drop_p;         // > 0: inverted dropout with the counter-based generator of the dropout op (element = row * Cout + column)
  uint64_t ep_seed_mix, ep_offset;   // mix64(seed) and the call's counter offset
  const void* ep_x2;       // second raw tensor (downsample BN of the same block) -> stat_b_*
  float* stat_b_sum;       // = sum v   (again, so the downsample finalize sees the same slab layout)
  float* stat_b_sq;        // = sum v*ep_x2
  int stat_stride;         // floats per partial row (Cout for forward stats, 2*Cout for the bwd layout)
  int N, IH, IW, C, Cpitch;
  int Cout, wrow;      // wrow = wtaps*C = elements per weight row
  int Sy, Sx, OS;
  int OHf, OWf;
  int ncls, total_mblk, nblk_n;
  int simple_src;      // 1x1 / stride 1 / no padding, one tap class: tile row m IS source pixel m (no per-row index arithmetic in the prologue)
  uint64_t in_bytes;   // size of the gather source in bytes (the pipelined kernel's buffer descriptor: reads past it return zeros); 0 = not set
  int bm_step;         // valid rows per row block (= the tile height except for the pipelined kernel's 196-of-224-row tiles); set by finish_classes
  int pipe_ok;         // the caller can take a row-block count that differs from ceil(rows / 128) (statistics slabs), so any tile height may be chosen
  int group_m;         // > 1: tiles are ordered in groups of group_m row blocks x all column blocks, row block fastest (Linear GEMMs whose
                       // weight matrix exceeds an XCD's 4 MB L2: the group's A rows stay resident while the weight streams through ONCE per group)
  // ---- two-source A operand (algebraic BatchNorm backward of an expanding 1x1 convolution, backbone.hip): reduction indices
  // [0, K1) come from `in` (row pitch Cpitch), [K1, C) from `in2` (row pitch Cpitch >> pitch2_shift).  simple_src launches of the
  // 128-row kernels only.
  const void* in2;
  int K1, pitch2_shift;
  int addend_sub;          // 2: addend is the stride-2 SAMPLED tensor [N][ceil(OHf/2)][ceil(OWf/2)][Cout] (a stride-2 1x1 convolution's data gradient,
                           // kept compact): only even (y, x) output pixels have an addend row.  1x1 / stride 1 launches only.
  const float* ep_mul;     // light-epilogue launches: per-output-channel multiplier applied to the accumulator BEFORE bias / addend (train-mode BatchNorm scale)
  uint8_t* ep_mask_out;    // light-epilogue launches: also write the ReLU mask of the stored tile, one byte per 16-byte chunk, bit e = (stored value e > 0)
                           // (the forward's second pass of a two-pass conv + BatchNorm + residual + ReLU, backbone.hip)
  int ablate;          // -DMMSKIN_ABLATE builds only (`make ablate`): bit0 skip A DMA, bit1 skip B DMA, bit2 skip MFMA, bit3 skip stores; always 0 in the production library
#ifdef MMSKIN_ABLATE
  unsigned long long* stamps;   // in-kernel phase stamps (scripts/conv_stamps.py): [workgroup][8] s_memtime values, or null
#endif
  TapClass cls[4];
};

// wgrad: dW[cout][tap][c] = sum_m dY[m][cout] * gather(in)[m][tap][c]
struct WgradArgs {
  const void* dy;      // [M][Cout]   (M = N*OH*OW, NHWC flattened)
  const void* in;      // gather source [N][IH][IW][Cpitch]
  float* slab;         // [nsplit][Cout][Ktot] fp32 partials
  int N, IH, IW, C, Cpitch;
  int OH, OW;          // rows m = (img, oy, ox)
  int Cout, Ktot;      // Ktot = ntaps*C
  int Sy, Sx;
  int ntaps;
  int M, m_per_split, nsplit;
  int nblk_o, nblk_k;  // tiles over Cout / Ktot
  int simple1x1;       // host-side selector: 1x1 / stride 1 / no padding (gathered-input row m IS pixel m)
  // ---- ring kernel only: Gram tiles (algebraic BatchNorm backward, backbone.hip).  Cout tiles ob >= nblk_o_main take their
  // "dY" rows from the INPUT tensor instead (columns (ob - nblk_o_main) * BO .. of its first gram_cols channels, zeros beyond), so
  // the launch also produces in^T in (slab rows Cout ..) and, through one extra MFMA against a ones fragment, the column sums of
  // `in` (colsum[(split * G + group)][gram tile columns]).  nblk_o_main == nblk_o: plain weight gradient.
  int nblk_o_main, gram_cols;
  float* colsum;
  int ablate;          // timing experiments only: bit0 skip X loads, bit1 skip Y loads, bit2 skip MFMA+LDS reads, bit3 skip slab store, bit4 skip LDS writes
  int8_t offy[MMSKIN_MAX_TAPS], offx[MMSKIN_MAX_TAPS];
};

// ring form of the weight-gradient GEMM (wgrad_ring.hip): tile = 64 wo x 64 wk, 8 waves, 8 / (wo wk) pixel groups per workgroup
struct WgradRingPlan { int wo, wk, nsplit, mps; bool s1; int gram_tiles; };
bool wgrad_ring_tile(int M, int Cout, int Ktot, WgradRingPlan& r, bool simple);   // tile shape + split count from the GEMM dimensions (simple: 1x1 / stride 1)
bool wgrad_ring_plan(const WgradArgs& a, WgradRingPlan& r);              // false: the shape stays on the register-staged kernel of wgrad.hip
int wgrad_ring_launch(WgradArgs& a, const WgradRingPlan& r, hipStream_t st);   // fills a.nsplit / m_per_split / nblk_*; the caller reduces the slabs
int wgrad_ring_launch_count();
// dY^T in plus in^T in and colsum(in) in one launch (1x1 / stride 1 layers whose Cout x Cin the ring kernel tiles): the slab rows are
// [Cout main | gram tile rows]; returns the plan through r, reduces the slabs into s_out [Cout + gram rows][Cin] and the column sums
// into colsum_out [Cin].  scratch: wgrad_gram_scratch_floats() floats behind the caller's slab.
// mode 1: dY^T in only (no Gram rows, colsum_out untouched); mode 2: in^T in + colsum(in) only (g unused; s_out = [gram rows][Cin]).
bool wgrad_gram_plan(int M, int Cout, int Cin, WgradRingPlan& r, int mode = 0);
size_t wgrad_gram_slab_bytes(int M, int Cout, int Cin, int mode = 0);
int launch_wgrad_gram(int N, int H, int W, int Cin, int Cout, const bf16_t* g, const bf16_t* in, float* slab, float* s_out, float* colsum_out,
                      hipStream_t st, int mode = 0);

// Inference epilogue (eval-mode BatchNorm folded into the conv): out = [relu](acc + bias[c] + addend)
struct FwdFuse {
  const float* bias = nullptr;    // [Cout] fp32
  const void* addend = nullptr;   // optional residual, same layout as out
  bool relu = false;
  bool gelu = false;              // exact GELU instead of ReLU
  float* out_f32 = nullptr;       // write the result as fp32 here (the T output pointer is then unused)
  // transformer-residual epilogue: out = res_f32 + gamma * dropout(act(acc + bias))
  const float* gamma = nullptr;   // [Cout]
  const float* res_f32 = nullptr; // [rows][Cout] fp32
  float drop_p = 0.f;
  uint64_t seed = 0, offset = 0;
  uint8_t* mask_out = nullptr;    // write the ReLU mask of the result (one byte per 16-byte chunk) beside it
  const float* mul = nullptr;     // [Cout] fp32: out = act(acc * mul + bias + addend)
};
template <typename T>
int launch_conv_fwd(const ConvShape& s, const T* in, const T* w_staged, T* out, float* stat_sum,
                    float* stat_sq, hipStream_t st, const FwdFuse* fuse = nullptr, int* stat_rows_out = nullptr);
// number of stat partial rows the forward launch produces (rows of stat_sum / stat_sq) when the caller passes no stat_rows_out;
// with stat_rows_out the launcher may pick a taller tile and reports the (smaller) row count there -- the slabs sized for
// conv_fwd_stat_rows() always suffice
int conv_fwd_stat_rows(const ConvShape& s);

// Optional epilogue fusion for launch_conv_dgrad (see ConvGemmArgs::ep_*): partial slabs use the
// bn_bwd layout [row][2][C]; rows = conv_dgrad_partial_rows(s).
struct DgradFuse {
  const void* mask_y = nullptr;
  const uint8_t* mask_bits = nullptr;   // alternative to mask_y: 1 bit per element (1/16 of the bytes)
  const void* x = nullptr;
  int x_pitch = 0;              // row pitch of x in elements when x is a channel prefix of a wider tensor (0 = compact)
  const float* scale = nullptr;
  const float* shift = nullptr;
  const void* x2 = nullptr;
  float* partial = nullptr;     // [rows][2][Cin]: sum dz, sum dz*x
  float* partial_b = nullptr;   // [rows][2][Cin]: sum dz, sum dz*x2
  int rows_written = 0;         // out: partial rows the launch produced
  // second gradient-side operand (1x1 / stride 1 only): din = [dout | in2] x wt_staged^T + bias, where wt_staged rows are
  // s.Cout + k2 long and in2 is [rows][k2] (k2 * 2^j = s.Cout)
  const void* in2 = nullptr;
  int k2 = 0;
  const float* bias = nullptr;  // [Cin] fp32, added before the mask
  bool addend_s2 = false;       // the addend is compact: [N][ceil(H/2)][ceil(W/2)][Cin], added at the even pixels (see ConvGemmArgs::addend_sub)
};
int conv_dgrad_partial_rows(const ConvShape& s);

// din[N][H][W][Cin] = dgrad(dout[N][OH][OW][Cout]); wt_staged is [Cin][kh*kw][Cout].
// addend (optional, may alias din) is added in the epilogue.  accumulate_only_touched: for classes
// with no taps (e.g. 1x1 stride 2) leave din untouched when addend aliases din, else write zeros.
template <typename T>
int launch_conv_dgrad(const ConvShape& s, const T* dout, const T* wt_staged, T* din, const T* addend,
                      hipStream_t st, DgradFuse* fuse = nullptr);

// The stem conv (7x7 s2 p3, Cin=3) runs as a "virtual" 8x1-tap conv with 32 channels over the padded
// NHWC4 image produced by stem_pack (see stem.hip).  out = [N][OH][OW][64].
template <typename T>
int launch_stem_conv_fwd(int N, int OH, int OW, int Hp, int Wp, const T* img4, const T* wv, T* out,
                         float* stat_sum, float* stat_sq, hipStream_t st, int* stat_rows_out = nullptr);
int stem_conv_stat_rows(int N, int OH, int OW);   // upper bound of the statistics rows a launch writes (allocation); the launch reports its own count
// direct 7x7 / stride 2 convolution from an LDS-resident input window (stem7x7.hip; bf16, 112-wide output rows): launch_stem_conv_fwd routes to it
bool stem7x7_takes(int OH, int OW, int Hp, int Wp);
int stem7x7_stat_rows(int N, int OH);
int launch_stem7x7_fwd(int N, int OH, int OW, int Hp, int Wp, const bf16_t* img4, const bf16_t* wv, bf16_t* out, float* stat_sum, float* stat_sq,
                       hipStream_t st);

// VGG's first conv (3x3 s1 p1, Cin=3) as a virtual conv over the zero-bordered NHWC8 image of pack_nhwc8: one tap
// per kernel row, each reading 32 contiguous elements (4 pixels x 8 ch; 3 x 3 of them carry weights), plus one
// all-zero tap so K = 4 x 32 = 128.  wv / dwv are [64][4][32].
template <typename T>
int launch_vgg_first_conv_fwd(int N, int H, int W, int Hp, int Wp, const T* img8, const T* wv, T* out,
                              const FwdFuse* fuse, hipStream_t st, int stride = 1, float* stat_sum = nullptr,
                              float* stat_sq = nullptr);
size_t vgg_first_wgrad_slab_bytes(int N, int H, int W);   // H, W = OUTPUT size
template <typename T>
int launch_vgg_first_conv_wgrad(int N, int H, int W, int Hp, int Wp, const T* dout, const T* img8, float* slab,
                                float* dwv, hipStream_t st, int stride = 1);

// 3x3 / stride 1 / pad 1, 64 -> 64 channels at 56 x 56 (ResNet-50 layer1 conv2): all nine taps from one staged input window, weights
// resident in LDS, one workgroup per image (conv3x3_c64.hip).  launch_conv_fwd / launch_conv_dgrad route to it when it takes the shape.
bool conv3x3_c64_takes(const ConvShape& s, bool bf16);
int launch_conv3x3_c64_fwd(const ConvShape& s, const bf16_t* in, const bf16_t* w_staged, bf16_t* out, float* stat_sum, float* stat_sq,
                           int stat_stride, hipStream_t st);
int launch_conv3x3_c64_dgrad(const ConvShape& s, const bf16_t* dout, const bf16_t* wt_staged, bf16_t* din, DgradFuse* fuse, hipStream_t st);

size_t conv_wgrad_slab_bytes(const ConvShape& s);
// dw: fp32 OIHW [Cout][Cin][kh][kw], reduced over the split slabs.  cout_valid / cin_valid (0 = all): when the
// GEMM operands carry zero padding (s.Cout / s.Cin rounded up to 64), dw is the UNPADDED
// [cout_valid][cin_valid][kh][kw] tensor and the padded rows / channels are dropped.
template <typename T>
int launch_conv_wgrad(const ConvShape& s, const T* dout, const T* in, float* slab, float* dw,
                      hipStream_t st, int cout_valid = 0, int cin_valid = 0);
size_t stem_wgrad_slab_bytes(int N, int OH, int OW);
template <typename T>
int launch_stem_conv_wgrad(int N, int OH, int OW, int Hp, int Wp, const T* dout, const T* img4,
                           float* slab, float* dwv, hipStream_t st);

// all-taps 3x3 / stride 1 / pad 1 weight gradient, ring form (wgrad3_ring.hip): launches the GEMM; the caller reduces *nsplit slabs [Cout][9 Cin]
bool wgrad3_ring_takes(const ConvShape& s, int* nsplit);
int launch_wgrad3_ring(const ConvShape& s, const bf16_t* dout, const bf16_t* in, float* slab, hipStream_t st, int* nsplit);

// algebraic BatchNorm backward of an expanding 1x1 convolution (abn.hip)
int abn_prep(const float* W, const float* cA, const float* cB, const float* cC, int C4, int Cw, bf16_t* wd, float* bias, float* coef_copy,
             hipStream_t st);
int abn_wgrad_finalize(const float* S, const float* colsum, const float* W, const float* coef, int C4, int Cw, float* dW, hipStream_t st,
                       const float* gram = nullptr);   // gram: y^T y kept from the forward pass (else the rows behind S's C4 main rows)
// statistics (sum, sum of squares per output channel, one-row table) of x = y W^T from gram = y^T y [Cw][Cw] and colsum(y)
int gram_stats(const float* gram, const float* colsum, const float* W, int C4, int Cw, float* stat_sum, float* stat_sq, hipStream_t st);
int gram_stats_finalize(const float* gram, const float* colsum, const float* W, int C4, int Cw, double count, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                        hipStream_t st);   // ... with bn_finalize's coefficients and running statistics from the same launch
int abn_sgx(const float* S, const float* W, int C4, int Cw, float* sgx, hipStream_t st);
