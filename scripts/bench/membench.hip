// Streaming-kernel micro-benchmark for MI355X: how fast can a 2-reads-1-write elementwise pass (the BN-backward
// apply shape: dx = cA*dz + cB*x + cC on bf16 NHWC) go, as a function of grid size, chunks in flight per thread and
// store policy?   hipcc --offload-arch=gfx950 -O3 membench.hip -o membench && ./membench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ uint32_t pk(float a, float b) {
  __bf16 x = (__bf16)a, y = (__bf16)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

template <int U, int NT, int READS>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o,
                                         const float* __restrict__ cA, const float* __restrict__ cB, size_t n, int cpr) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = blockIdx.x * (size_t)256 + threadIdx.x; i0 < n; i0 += stride * U) {
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i = i0 + u * stride;
      if (i < n) { va[u] = a[i]; if (READS > 1) vb[u] = b[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i = i0 + u * stride;
      if (i < n) {
        const int c0 = (int)(i % cpr) * 8;
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x0 = lo(va[u][e]), x1 = hi(va[u][e]);
          float y0 = READS > 1 ? lo(vb[u][e]) : 0.f, y1 = READS > 1 ? hi(vb[u][e]) : 0.f;
          r[e] = pk(cA[c0 + 2 * e] * x0 + cB[c0 + 2 * e] * y0 + 1.f, cA[c0 + 2 * e + 1] * x1 + cB[c0 + 2 * e + 1] * y1 + 1.f);
        }
        if (NT) __builtin_nontemporal_store(r, &o[i]); else o[i] = r;
      }
    }
  }
}

int main() {
  const size_t rows = 802816, C = 256;           // layer-1 conv3 output at batch 256
  const size_t n = rows * C / 8;                 // 16-byte chunks
  u32x4 *a, *b, *o; float *cA, *cB;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, n * 16);
  hipMalloc(&cA, C * 4); hipMalloc(&cB, C * 4);
  hipMemset(a, 0x11, n * 16); hipMemset(b, 0x22, n * 16); hipMemset(cA, 0, C * 4); hipMemset(cB, 0, C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
#define RUN(U, NT, READS)                                                                      \
  for (int g : grids) {                                                                        \
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<U, NT, READS>), dim3(g), dim3(256), 0, 0, a, b, o, cA, cB, n, (int)(C / 8)); \
    hipEventRecord(e0, 0);                                                                     \
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<U, NT, READS>), dim3(g), dim3(256), 0, 0, a, b, o, cA, cB, n, (int)(C / 8)); \
    hipEventRecord(e1, 0); hipEventSynchronize(e1);                                            \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                \
    double us = ms * 100.0;                                                                    \
    printf("reads=%d U=%d nt=%d grid=%6d  %7.1f us  %5.2f TB/s\n", READS, U, NT, g, us, (READS + 1) * n * 16 / us / 1e6); \
  }
  RUN(1, 0, 2) RUN(2, 0, 2) RUN(4, 0, 2) RUN(2, 1, 2) RUN(4, 1, 2)
  RUN(1, 0, 1) RUN(2, 0, 1) RUN(4, 0, 1) RUN(4, 1, 1)
  return 0;
}
