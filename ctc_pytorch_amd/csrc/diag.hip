// diag.hip -- diagnostics of libctcn.so that are not on the compute path.
//
// ctcn_diag_squat: a "squatter" kernel -- k workgroups per XCD that do nothing but hold their CU slots (wave slots, and optionally
// LDS) for a given time.  It is what an RCCL kernel looks like to the rest of the chip while it waits for a slow peer: resident
// workgroups on every XCD that the dispatcher cannot move.  The persistent recurrences (rnn.hip) need their whole grid co-resident
// on exact XCDs, so the data-parallel design is tested against it on ONE GPU: tests/test_gpu_kernels.py
// launches squatters on a third stream at random points of training steps and requires the hand-offs to survive
// (tools/squat_stress.py is the long version).  The wait is bounded by the clock, never by a flag somebody has to raise.
#include "common.h"

namespace {

__global__ void squat_kernel(unsigned long long ticks, int lds_floats, float *sink) {
  extern __shared__ float held[];
  const unsigned long long t0 = __builtin_readcyclecounter();          // s_memtime: shader clock, never stops
  const unsigned long long r0 = wall_clock64();                         // s_memrealtime: constant 100 MHz
  float acc = 0.0f;
  if (lds_floats > 0) held[threadIdx.x % lds_floats] = (float)threadIdx.x;
  while (wall_clock64() - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_readcyclecounter() - t0 > (1ull << 36)) break;        // ~30 s at 2.4 GHz: a second bound, in case the realtime counter misbehaves
  }
  if (lds_floats > 0) acc = held[(threadIdx.x + 1) % lds_floats];
  if (sink && acc == -1.0f) *sink = acc;                                 // keeps the LDS allocation alive
}

}  // namespace

extern "C" int ctcn_diag_squat(int wgs_per_xcd, int threads, int lds_bytes, unsigned usec, void *stream) {
  CTCN_REQUIRE(wgs_per_xcd >= 1 && wgs_per_xcd <= 64 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && lds_bytes >= 0 && lds_bytes <= 160 * 1024,
               "ctcn_diag_squat: bad arguments");
  CTCN_REQUIRE(usec <= 5000000u, "ctcn_diag_squat: at most 5 s");
  const int nx = ctcn_device_xcds() > 1 ? ctcn_device_xcds() : 8;
  if (lds_bytes > 64 * 1024)
    CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(squat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  // consecutive workgroup ids are dealt round-robin over the XCDs: nx * k workgroups = k per XCD
  hipLaunchKernelGGL(squat_kernel, dim3(nx * wgs_per_xcd), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)usec * 100ull, lds_bytes / 4, (float *)nullptr);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
