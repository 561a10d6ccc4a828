// tools/mb_store.hip -- how fast can ONE CU push 1-KB block stores (64 lanes x 16 B, the partial tiles of rnn_bwd_scatter) and dword
// publishes (rnn_fwd_tagged) into its XCD's L2?  `nw` waves of one workgroup per CU issue `per` stores each, back to back; cycles from the
// first issue until the wave's LAST store is acknowledged (s_waitcnt vmcnt(0)), and until it is ISSUED.  20 workgroups on one XCD's CUs
// (grid 8 x 20, workgroups of other XCDs exit) as in the recurrences.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: b128 plain, 1: b128 sc1, 2: dword plain (64 x 4 B scattered like the forward publish), 3: b128 loads sc1 (poll)
__global__ __launch_bounds__(1024) void k(long long *out, unsigned *buf, int per, int reps, int nw) {
  unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if ((x & 15) != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nw) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 26, 0x00020000);
  const unsigned base = (unsigned)((blockIdx.x >> 3) * 32 + wave) * (unsigned)per * 1024u;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  long long ti = 0, ta = 0;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < per; ++i) {
      const unsigned off = base + (unsigned)i * 1024u;
      if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + lane * 16, 0, 0);
      else if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + lane * 16, 0, 16);
      else if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b32(v.w, rs, off + (unsigned)(((lane & 15) * 16 + (lane >> 4)) * 4), 0, 0);
      else { const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 16, 0, 16); acc += q.x; }
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    ti += t1 - t0; ta += t2 - t0;
  }
  if (lane == 0 && (blockIdx.x >> 3) == 3) { out[wave * 2] = ti; out[wave * 2 + 1] = ta + (acc == 12345u); }
}
int main() {
  long long *out, h[32]; unsigned *buf;
  hipMalloc(&out, 256); hipMalloc(&buf, 1 << 26); hipMemset(buf, 0, 1 << 26);
  const int reps = 200;
  const char *names[4] = {"16-B stores, plain (write-back)", "16-B stores, sc1", "dword stores, plain", "16-B loads, sc1 (polls)"};
  for (int mode = 0; mode < 4; ++mode)
    for (int nw : {1, 4, 8, 12}) {
      const int per = mode == 2 ? 1 : (nw == 1 ? 20 : (nw == 4 ? 5 : (nw == 8 ? 3 : 2)));
      hipMemset(out, 0, 256);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(160), dim3(1024), 0, 0, out, buf, per, reps, nw);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(160), dim3(1024), 0, 0, out, buf, per, reps, nw);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(160), dim3(1024), 0, 0, out, buf, per, reps, nw);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(160), dim3(1024), 0, 0, out, buf, per, reps, nw);
      hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
      printf("%-34s %2d wave(s) x %2d per wave = %4.1f KB per CU, 20 CUs of one XCD: issued after %6.0f cycles, acknowledged after %6.0f  (wave 0)\n", names[mode], nw, per,
             nw * per * (mode == 2 ? 0.25 : 1.0), (double)h[0] / reps, (double)h[1] / reps);
    }
  return 0;
}
