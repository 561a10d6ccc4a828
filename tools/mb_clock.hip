// tools/mb_clock.hip -- what clock do tiny dependent kernels run at?  (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_chain(float *out, long long *clk, int n, float a, float b) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1];
}
__global__ void valu_chain(float *out, long long *clk, int n, float a) {
  float x = a;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) x = fmaf(x, 1.0001f, 0.5f);
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[2] = c1 - c0; clk[3] = w1 - w0; }
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
  float *out; long long *clk, h[4];
  hipMalloc(&out, 1 << 20); hipMalloc(&clk, 64);
  int wallrate = 0; hipDeviceGetAttribute(&wallrate, hipDeviceAttributeWallClockRate, 0);
  int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, max shader clock %d kHz\n", wallrate, sclk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {160, 1024}) for (int gap = 0; gap < 2; ++gap) {
    // gap=0: 400 back-to-back launches; gap=1: a device sync between launches (idle GPU between kernels)
    hipEventRecord(e0);
    for (int it = 0; it < 400; ++it) { hipLaunchKernelGGL(mfma_chain, dim3(blocks), dim3(256), 0, 0, out, clk, 20, 1.f, 2.f); if (gap) hipDeviceSynchronize(); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipLaunchKernelGGL(valu_chain, dim3(blocks), dim3(256), 0, 0, out, clk, 1000, 1.f);
    hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    printf("blocks %4d sync-gap %d: 40 MFMA chain = %lld clock64 ticks, %lld wall ticks (%.0f ns) -> %.1f ticks/MFMA; valu 1000 fma = %lld ticks %lld wall; launch interval %.2f us\n",
           blocks, gap, h[0], h[1], h[1] * 1e6 / wallrate, h[0] / 40.0, h[2], h[3], ms * 1e3 / 400);
  }
  return 0;
}
