// tools/mb_xcd.hip -- flag ping-pong latency between two workgroups: same XCD vs different XCDs, by cache-policy bits
// (development aid; decides whether an XCD-local hand-off through the shared L2 beats the device-scope one)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mb_xcd.bin tools/mb_xcd.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned ld(const __amdgpu_buffer_rsrc_t rs, unsigned off, int aux) {
  switch (aux) {
    case 0: return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
    case 1: return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 1);
    case 16: return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16);
    default: return __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 17);
  }
}
__device__ __forceinline__ void st(const __amdgpu_buffer_rsrc_t rs, unsigned off, unsigned v, int aux) {
  switch (aux) {
    case 0: __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0); break;
    case 1: __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 1); break;
    case 16: __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 16); break;
    default: __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 17); break;
  }
}

// flags[0]: ping (written by wgA), flags[64]: pong (written by wgB); separate 256-B lines
__global__ __launch_bounds__(64) void pingpong(unsigned *flags, int wgA, int wgB, int iters, int ld_aux, int st_aux, int *xcc, long long *res, int spin) {
  extern __shared__ float pad[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(x & 15);
  if ((int)blockIdx.x != wgA && (int)blockIdx.x != wgB) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 1024, 0x00020000);
  const bool isA = (int)blockIdx.x == wgA;
  long long w0 = wall_clock64(), c0 = clock64();
  int fails = 0;
  for (int i = 1; i <= iters; ++i) {
    if (isA) {
      if (threadIdx.x == 0) st(rs, 0, (unsigned)i, st_aux);
      int n = 0;
      while (ld(rs, 256, ld_aux) != (unsigned)i && ++n < spin) { asm volatile("" ::: "memory"); }
      if (n >= spin) { ++fails; break; }
    } else {
      int n = 0;
      while (ld(rs, 0, ld_aux) != (unsigned)i && ++n < spin) { asm volatile("" ::: "memory"); }
      if (n >= spin) { ++fails; break; }
      if (threadIdx.x == 0) st(rs, 256, (unsigned)i, st_aux);
    }
  }
  long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0 && isA) { res[0] = w1 - w0; res[1] = c1 - c0; res[2] = fails; }
}

int main() {
  unsigned *flags; int *xcc; long long *res;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&xcc, 4096)); CK(hipMalloc(&res, 64));
  int wallrate = 0; hipDeviceGetAttribute(&wallrate, hipDeviceAttributeWallClockRate, 0);
  const size_t lds = 96 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<int> hx(256);
  const int iters = 2000;
  struct P { int a, b; const char *name; } pairs[] = {{0, 8, "wg0-wg8 "}, {0, 16, "wg0-wg16"}, {0, 1, "wg0-wg1 "}, {0, 4, "wg0-wg4 "}, {3, 251, "wg3-wg251"}};
  struct M { int ld, st; const char *name; } modes[] = {{16, 16, "ld sc1 / st sc1"}, {17, 17, "ld sc0+sc1 / st sc0+sc1"}, {1, 1, "ld sc0 / st sc0"}, {1, 0, "ld sc0 / st plain"}, {16, 0, "ld sc1 / st plain"}, {0, 0, "plain / plain"}};
  for (auto &p : pairs)
    for (auto &m : modes) {
      CK(hipMemset(flags, 0, 4096)); CK(hipMemset(res, 0, 64));
      hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), lds, 0, flags, p.a, p.b, iters, m.ld, m.st, xcc, res, 200000);
      CK(hipDeviceSynchronize());
      long long h[3]; CK(hipMemcpy(h, res, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, 1024, hipMemcpyDeviceToHost));
      printf("%s (xcc %d,%d)  %-26s round trip %7.1f ns  %7.0f cycles%s\n", p.name, hx[p.a], hx[p.b], m.name, h[0] * 1e6 / wallrate / iters, (double)h[1] / iters,
             h[2] ? "  [TIMED OUT: never became visible]" : "");
    }
  printf("xcc of wg 0..31:");
  for (int i = 0; i < 32; ++i) printf(" %d", hx[i]);
  printf("\n");
  return 0;
}
