// tools/mb_mfma16.hip -- issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD: 5 interleaved accumulation chains of 6 (the exchange
// phase of rnn_bwd_scatter2), against the same count as one chain and as 30 independent products.  s_memtime stamps, cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(long long *out, const float *in, float *sink, int reps, int active_waves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8_t a[4], w[10];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)in[(lane * 8 + e + i) & 255];
  for (int i = 0; i < 10; ++i) for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)in[(lane * 8 + e + 7 * i) & 255];
  f32x4 acc[5];
  for (int t = 0; t < 5; ++t) acc[t] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  if (wave >= active_waves) return;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[blk * 2 + 1], w[t * 2], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[blk * 2], w[t * 2 + 1], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[blk * 2], w[t * 2], acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 30; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], w[i % 10], acc[0], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < 5; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (s == 12345.f) *sink = s;
}
int main() {
  long long *out, h[8]; float *in, *sink;
  hipMalloc(&out, 64); hipMalloc(&in, 1024); hipMalloc(&sink, 4); hipMemset(in, 0, 1024);
  const int reps = 2000;
  for (int aw : {1, 4, 8}) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, out, in, sink, reps, aw); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("5 interleaved chains of 6, %d wave(s) active in the workgroup: %.1f cycles per MFMA (wave 0)\n", aw, (double)h[0] / (30.0 * reps));
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, out, in, sink, reps, aw); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("one dependent chain of 30,      %d wave(s) active in the workgroup: %.1f cycles per MFMA (wave 0)\n", aw, (double)h[0] / (30.0 * reps));
  }
  return 0;
}
