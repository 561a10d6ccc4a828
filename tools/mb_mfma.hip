// MFMA issue-rate microbenchmark (development aid): cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD and for two,
// with 8 independent accumulators, as in the 256 x 256 plane tile.   hipcc --offload-arch=gfx950 -O3 tools/mb_mfma.hip -o tools/mb_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  long long h[8];
  for (int threads : {256, 512}) {
    for (int blocks : {1, 256}) {
      const int iters = 2000;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const double n = 24.0 * iters;
      printf("threads %3d blocks %3d: clock64 ticks per MFMA per wave %.1f; wall %.1f us -> %.1f ns per MFMA per wave; %.0f TFLOP/s\n", threads, blocks,
             h[0] / n, ms * 1e3, ms * 1e6 / n, blocks * (threads / 64) * n * 32768.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
