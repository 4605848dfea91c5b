// Probe (MI355X): semantics of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) and the cost of s_barrier.
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/blds_probe scripts/bench/blds_probe.hip ; ./build_ab/blds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// each wave: one DMA of 64 x 16 B from src + soff + lane * 16 (lanes >= oob_from get an out-of-range offset) into LDS at lds_off,
// then every thread copies its 16 bytes from LDS to out
__global__ void probe(const uint4* src, uint32_t nbytes, uint32_t lds_off, uint32_t soff, int oob_from, uint4* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 160 * 1024 / 16 - 64; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  const uint64_t b = (uint64_t)(uintptr_t)src;
  srd_t srd = {(uint32_t)b, (uint32_t)(b >> 32) & 0xffffu, nbytes, 0x00020000u};
  uint32_t voff = lane < oob_from ? lane * 16u : 0xF0000000u;
  const uint32_t dst = (uint32_t)(uintptr_t)(lds_ptr_t)smem + lds_off;
  uint32_t so;
  asm volatile("s_mov_b32 %0, %1" : "=s"(so) : "s"(soff));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(dst), "v"(voff), "s"(srd), "s"(so) : "memory");
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<const uint4*>(smem + lds_off + lane * 16);
}

// cost of s_barrier: 8 waves, n barriers; variant 1: the two wave groups alternate a conditional barrier as the conv kernel does
__global__ void bar_loop(int n, int variant, unsigned long long* cyc) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool g1 = wid >= 4;
  unsigned long long t0 = clock64();
  if (variant == 0) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_barrier();
  } else if (variant == 1) {
    for (int i = 0; i < n; ++i) {
      if (g1) __builtin_amdgcn_s_barrier();
      asm volatile("s_nop 0");
      if (!g1) __builtin_amdgcn_s_barrier();
      asm volatile("s_nop 0");
    }
  } else {
    for (int i = 0; i < n; ++i) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_setprio(1); asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"); __builtin_amdgcn_s_setprio(0); }
  }
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int N = 1 << 16;
  std::vector<uint32_t> h(N);
  for (int i = 0; i < N; ++i) h[i] = i;
  uint4 *src, *out;
  hipMalloc(&src, N * 4); hipMalloc(&out, 64 * 16);
  hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct Case { uint32_t lds_off, soff; int oob_from; uint32_t nbytes; } cases[] = {
      {1024, 0, 64, N * 4}, {70000 / 16 * 16, 0, 64, N * 4}, {140000 / 16 * 16, 0, 64, N * 4}, {1024, 4096, 64, N * 4},
      {100000 / 16 * 16, 8192, 48, N * 4}, {1024, 0, 64, 512}, {1024, 1024, 64, 512}};
  for (auto c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, c.nbytes, c.lds_off, c.soff, c.oob_from, out);
    uint32_t r[256];
    hipMemcpy(r, out, 1024, hipMemcpyDeviceToHost);
    int good = 0, zero = 0, dead = 0;
    for (int l = 0; l < 64; ++l) {
      const uint32_t want = c.soff / 4 + l * 4;
      if (r[l * 4] == want && r[l * 4 + 3] == want + 3) ++good;
      else if (r[l * 4] == 0 && r[l * 4 + 3] == 0) ++zero;
      else if (r[l * 4] == 0xdeadbeefu) ++dead;
    }
    printf("lds_off %6u soff %5u oob_from %2d num_records %6u : lanes with data %2d, zeros %2d, untouched %2d  (lane 0: %u lane 63: %u)\n",
           c.lds_off, c.soff, c.oob_from, c.nbytes, good, zero, dead, r[0], r[63 * 4]);
  }
  unsigned long long* cyc;
  hipMalloc(&cyc, 256 * 8);
  for (int v = 0; v < 3; ++v) {
    const int n = 4096;
    hipLaunchKernelGGL(bar_loop, dim3(256), dim3(512), 0, 0, n, v, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(bar_loop, dim3(256), dim3(512), 0, 0, n, v, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("barrier variant %d: %.1f ns per loop iteration (%d iterations), clock64 delta %.1f per iteration\n", v, ms * 1e6 / n, n, (double)c0 / n);
  }
  return 0;
}
