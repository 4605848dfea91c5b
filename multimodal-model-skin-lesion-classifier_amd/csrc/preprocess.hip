// Input side of the hot path: the work the reference does on the host between decoding a sample and
// `model(image, metadata)` -- the val/test transform's A.Resize (skinLesionDatasets.py:116-120) and the
// OneHotEncoder + StandardScaler metadata encoding (skinLesionDatasets.py:133-183) -- as HBM-bound gfx950 kernels, so a
// batch travels over PCIe as raw uint8 pixels and integer category codes.
//
// Resize follows the published algorithm of cv2.resize(INTER_LINEAR) on 8-bit images (what albumentations' A.Resize calls):
// half-pixel centres, 11-bit fixed-point coefficients, horizontal pass in int, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2, replicated borders.  cv2 is not installed in the build container, so this
// is "parity unpinned" against cv2 itself; oracle/preprocess.py restates the same algorithm in numpy.
#include "../../include/mmskin.h"
#include "common.h"

namespace {

// source index + 11-bit coefficients of one destination coordinate (cv2 resize.cpp, linear branch)
__device__ __forceinline__ void lin_coef(int d, int dsize, int ssize, bool horizontal, int& s0, int& s1, int& c0, int& c1) {
  const double scale = (double)ssize / (double)dsize;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (horizontal) {   // the horizontal pass zeroes the fraction at the borders; the vertical one replicates rows
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  c0 = __float2int_rn((1.f - f) * 2048.f);
  c1 = __float2int_rn(f * 2048.f);
  s0 = min(max(s, 0), ssize - 1);
  s1 = min(max(s + 1, 0), ssize - 1);
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int N, int Hs, int Ws,
                                                        uint8_t* __restrict__ dst, int H, int W) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  if (w >= W) return;
  int y0, y1, b0, b1, x0, x1, a0, a1;
  lin_coef(h, H, Hs, false, y0, y1, b0, b1);
  lin_coef(w, W, Ws, true, x0, x1, a0, a1);
  const uint8_t* r0 = src + ((size_t)n * Hs + y0) * Ws * 3;
  const uint8_t* r1 = src + ((size_t)n * Hs + y1) * Ws * 3;
  uint8_t* out = dst + (((size_t)n * H + h) * W + w) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = (int)r0[x0 * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
    const int S1 = (int)r1[x0 * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    out[c] = (uint8_t)min(max(v, 0), 255);
  }
}

// one thread per output element: columns [0, onehot_width) are the one-hot blocks, the rest the standardised numerics
__global__ __launch_bounds__(256) void metadata_encode_kernel(const int32_t* __restrict__ codes, int n_cat,
                                                              const int32_t* __restrict__ col_offset,
                                                              const float* __restrict__ numeric, int n_num,
                                                              const float* __restrict__ mean, const float* __restrict__ scale,
                                                              float nan_fill, float* __restrict__ out, int B, int width) {
  const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (i >= (int64_t)B * width) return;
  const int b = (int)(i / width), j = (int)(i - (int64_t)b * width);
  const int onehot_width = width - n_num;
  float v;
  if (j < onehot_width) {
    int col = 0;   // the column this one-hot slot belongs to (n_cat is a handful: linear scan)
    while (col + 1 < n_cat && col_offset[col + 1] <= j) ++col;
    const int code = codes[(int64_t)b * n_cat + col];          // < 0: category unseen at fit time -> all zeros (handle_unknown='ignore')
    v = (code == j - col_offset[col]) ? 1.f : 0.f;
  } else {
    const int k = j - onehot_width;
    float x = numeric[(int64_t)b * n_num + k];
    if (x != x) x = nan_fill;                                   // pd.to_numeric(errors="coerce").fillna(-1)
    v = (x - mean[k]) / scale[k];
  }
  out[i] = v;
}

// ---- patch extraction (im2col) for the patch-embedding GEMMs of the transformer backbones (timm PatchEmbed / DaViT stem and
// downsample convolutions, reached through loadImageModelClassifier.py:117-121): rows = (n, oy, ox), columns = (c, ky, kx) --
// the order of conv.weight.flatten(1) -- so the convolution is ONE Linear over the rows.  The input is addressed through
// element strides, so NCHW images and NHWC token maps both work without a layout pass.  Out-of-image taps read zeros
// (padding); rows / columns of the input beyond the last full stride window are simply never read (timm crops / pads the same way).
struct PatchGeom { int N, C, H, W, k, stride, pad, OH, OW; int64_t sn, sc, sh, sw; };

__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, PatchGeom g, float* __restrict__ cols) {
  const int K = g.C * g.k * g.k;
  const int64_t total = (int64_t)g.N * g.OH * g.OW * K;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % K);
    const int64_t row = i / K;
    const int kx = col % g.k, ky = (col / g.k) % g.k, c = col / (g.k * g.k);
    const int ox = (int)(row % g.OW), oy = (int)((row / g.OW) % g.OH), n = (int)(row / ((int64_t)g.OW * g.OH));
    const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) v = x[n * g.sn + c * g.sc + iy * g.sh + ix * g.sw];
    cols[i] = v;
  }
}

// dx[n][c][iy][ix] = sum over the windows that contain the pixel: a gather, so no atomics and a deterministic sum
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, PatchGeom g, float* __restrict__ dx) {
  const int K = g.C * g.k * g.k;
  const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i enumerates the input in (n, iy, ix, c) order when the channel stride is 1 (NHWC), else (n, c, iy, ix)
    int n, c, iy, ix;
    if (g.sc == 1) { c = (int)(i % g.C); ix = (int)((i / g.C) % g.W); iy = (int)((i / ((int64_t)g.C * g.W)) % g.H); n = (int)(i / ((int64_t)g.C * g.W * g.H)); }
    else { ix = (int)(i % g.W); iy = (int)((i / g.W) % g.H); c = (int)((i / ((int64_t)g.W * g.H)) % g.C); n = (int)(i / ((int64_t)g.W * g.H * g.C)); }
    float s = 0.f;
    for (int ky = (iy + g.pad) % g.stride; ky < g.k; ky += g.stride) {
      const int oy = (iy + g.pad - ky) / g.stride;
      if (iy + g.pad - ky < 0 || oy >= g.OH) continue;
      for (int kx = (ix + g.pad) % g.stride; kx < g.k; kx += g.stride) {
        const int ox = (ix + g.pad - kx) / g.stride;
        if (ix + g.pad - kx < 0 || ox >= g.OW) continue;
        s += dcols[(((int64_t)n * g.OH + oy) * g.OW + ox) * K + (c * g.k + ky) * g.k + kx];
      }
    }
    dx[n * g.sn + c * g.sc + iy * g.sh + ix * g.sw] = s;
  }
}

static int patch_geom(PatchGeom& g, int N, int C, int H, int W, int k, int stride, int pad, int channels_last) {
  ARG_CHECK(N > 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0 && H + 2 * pad >= k && W + 2 * pad >= k,
            "im2col: bad shape N=%d C=%d %dx%d k=%d stride=%d pad=%d", N, C, H, W, k, stride, pad);
  g.N = N; g.C = C; g.H = H; g.W = W; g.k = k; g.stride = stride; g.pad = pad;
  g.OH = (H + 2 * pad - k) / stride + 1; g.OW = (W + 2 * pad - k) / stride + 1;
  if (channels_last) { g.sc = 1; g.sw = C; g.sh = (int64_t)W * C; g.sn = (int64_t)H * W * C; }
  else { g.sw = 1; g.sh = W; g.sc = (int64_t)H * W; g.sn = (int64_t)C * H * W; }
  return MMSKIN_OK;
}

}  // namespace

extern "C" {

int mmskin_im2col_forward(const float* x, int N, int C, int H, int W, int k, int stride, int pad, int channels_last, float* cols,
                          void* stream) {
  ARG_CHECK(x && cols, "im2col_forward: null argument");
  PatchGeom g;
  int rc = patch_geom(g, N, C, H, W, k, stride, pad, channels_last);
  if (rc) return rc;
  const int64_t total = (int64_t)N * g.OH * g.OW * C * k * k;
  int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)(blocks > (1 << 20) ? (1 << 20) : blocks)), dim3(256), 0, (hipStream_t)stream, x, g, cols);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_im2col_backward(const float* dcols, int N, int C, int H, int W, int k, int stride, int pad, int channels_last, float* dx,
                           void* stream) {
  ARG_CHECK(dcols && dx, "im2col_backward: null argument");
  PatchGeom g;
  int rc = patch_geom(g, N, C, H, W, k, stride, pad, channels_last);
  if (rc) return rc;
  const int64_t total = (int64_t)N * C * H * W;
  int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)(blocks > (1 << 20) ? (1 << 20) : blocks)), dim3(256), 0, (hipStream_t)stream, dcols, g, dx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_resize_u8(const uint8_t* src_nhwc, int N, int src_h, int src_w, uint8_t* dst_nhwc, int dst_h, int dst_w,
                     void* stream) {
  ARG_CHECK(src_nhwc && dst_nhwc, "resize_u8: null argument");
  ARG_CHECK(N > 0 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && N <= 65535 && dst_h <= 65535,
            "resize_u8: bad shape %d x %dx%d -> %dx%d", N, src_h, src_w, dst_h, dst_w);
  hipLaunchKernelGGL(resize_u8_kernel, dim3(ceil_div(dst_w, 256), dst_h, N), dim3(256), 0, (hipStream_t)stream, src_nhwc, N,
                     src_h, src_w, dst_nhwc, dst_h, dst_w);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_metadata_encode(const int32_t* codes, int n_cat, const int32_t* col_offset, int onehot_width, const float* numeric,
                           int n_num, const float* mean, const float* scale, float nan_fill, float* out, int batch, void* stream) {
  ARG_CHECK(out && col_offset && (codes || n_cat == 0) && (n_num == 0 || (numeric && mean && scale)), "metadata_encode: null argument");
  ARG_CHECK(batch > 0 && n_cat >= 0 && n_num >= 0 && onehot_width >= n_cat && onehot_width + n_num > 0, "metadata_encode: bad shape");
  const int width = onehot_width + n_num;
  const int64_t total = (int64_t)batch * width;
  hipLaunchKernelGGL(metadata_encode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes,
                     n_cat, col_offset, numeric, n_num, mean, scale, nan_fill, out, batch, width);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

}  // extern "C"
