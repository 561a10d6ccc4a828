// tools/mb_cumask.hip -- which XCDs does a CU-masked stream run on?  (development aid)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mb_cumask.bin tools/mb_cumask.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void probe(int *hist) {
  unsigned x, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { atomicAdd(&hist[x & 15], 1); atomicAdd(&hist[16 + ((hw >> 8) & 15)], 1); }
  // burn a little time so that all CUs get work
  float v = threadIdx.x;
  for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) hist[63] = 1;
}
int main() {
  int *hist; CK(hipMalloc(&hist, 256));
  const char *names[] = {"bits i%8>=4", "bits i>=128", "bits (i/32)%2==1", "bits (i/16)>=8 [same as i>=128]", "bits (i%16)>=8", "all"};
  for (int pat = 0; pat < 6; ++pat) {
    uint32_t mask[8] = {0};
    for (int i = 0; i < 256; ++i) {
      bool on = pat == 0 ? (i % 8 >= 4) : pat == 1 ? (i >= 128) : pat == 2 ? ((i / 32) % 2 == 1) : pat == 3 ? (i / 16 >= 8) : pat == 4 ? (i % 16 >= 8) : true;
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    CK(hipMemsetAsync(hist, 0, 256, st));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, st, hist);
    CK(hipStreamSynchronize(st));
    int h[64]; CK(hipMemcpy(h, hist, 256, hipMemcpyDeviceToHost));
    printf("%-34s xcc:", names[pat]);
    for (int i = 0; i < 8; ++i) printf(" %5d", h[i]);
    printf("   cu_id:");
    for (int i = 0; i < 16; ++i) printf(" %4d", h[16 + i]);
    printf("\n");
    CK(hipStreamDestroy(st));
  }
  return 0;
}
