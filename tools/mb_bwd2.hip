// tools/mb_bwd2.hip -- standalone phase accounting of the backward recurrences rnn_bwd_scatter (round 2) and rnn_bwd_scatter2 (round 3, item-wave
// gather): in-kernel clock64 stamps of item wave 0 and of the first exchange wave of slice 3 (development aid, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mb_bwd2.bin tools/mb_bwd2.hip ctc_pytorch_amd/csrc/core.hip
#define CTCN_PERSIST_STATS 1
#include "../ctc_pytorch_amd/csrc/rnn.hip"
#include <vector>

extern "C" int ctcn_gemm(int, int, int, int, int, const float *, int, const float *, int, float *, int, float, int, void *, size_t, void *) { return 0; }
extern "C" int ctcn_transpose01(const float *, float *, int, int, int, void *) { return 0; }
int ctcn_transpose01_pair(const float *, const float *, float *, float *, int, int, int, void *) { return 0; }
extern "C" int ctcn_dropout(const float *, float *, size_t, float, uint64_t, uint64_t, void *) { return 0; }
int ctcn_gemm_on_xcds(int, int, int, int, int, const float *, int, const float *, int, float *, int, float, int, void *, size_t, void *, unsigned, GemmPlanes *) { return 0; }
int ctcn_gemm_shift_b(int, int, int, const float *, int, const float *, int, float *, int, float, int, void *, size_t, void *, unsigned, int, GemmPlanes *) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 320, B = argc > 2 ? atoi(argv[2]) : 32, T = argc > 3 ? atoi(argv[3]) : 800, D = 2, G = 4;
  float *y, *gates, *aux, *wT, *dy;
  CK(hipMalloc(&y, (size_t)T * B * D * H * 4)); CK(hipMalloc(&gates, (size_t)T * B * D * G * H * 4)); CK(hipMalloc(&aux, (size_t)T * B * D * H * 4));
  CK(hipMalloc(&wT, (size_t)D * G * H * H * 4)); CK(hipMalloc(&dy, (size_t)T * B * D * H * 4));
  CK(hipMemset(y, 0, (size_t)T * B * D * H * 4)); CK(hipMemset(gates, 0, (size_t)T * B * D * G * H * 4)); CK(hipMemset(aux, 0, (size_t)T * B * D * H * 4));
  CK(hipMemset(wT, 0, (size_t)D * G * H * H * 4)); CK(hipMemset(dy, 0, (size_t)T * B * D * H * 4));
  RnnArgs a; a.cell = 0; a.T = T; a.B = B; a.H = H; a.D = D; a.G = G; a.step = 0; a.w0 = wT; a.w1 = wT + (size_t)G * H * H; a.y = y; a.gates = gates;
  a.aux = aux; a.dy = dy; a.state = nullptr;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nbt = (B + 15) / 16, nsl = (H + 15) / 16;
  const size_t hx_bytes = (size_t)2 * D * nbt * nsl * nsl * 1024, fl_bytes = (size_t)2 * D * nbt * nsl * nsl * 4 + 256;
  float *hx; unsigned *flags; int *status; long long *stats, h[32];
  CK(hipMalloc(&hx, hx_bytes)); CK(hipMalloc(&flags, fl_bytes)); CK(hipMalloc(&status, 4)); CK(hipMalloc(&stats, sizeof(h))); CK(hipMemset(stats, 0, sizeof(h)));
  const int nx = ctcn_device_xcds();
  printf("H %d B %d T %d: %d slices per group, %d groups, device XCDs %d\n", H, B, T, nsl, D * nbt, nx);
  if (nx <= 1) return 0;
  const int delays[] = {-1, 0, 8, 16, 0, 8, 16};
  for (int di = 0; di < 7; ++di) {
    const int delay = delays[di], NEWV = di <= 3 ? 4 : 8;
    PersistArgs pa = {}; pa.a = a; pa.hx = hx; pa.flags = flags; pa.status = status; pa.spin_limit = 1 << 20; pa.stats = stats;
    pa.poll_depth = 2; pa.poll_delay = delay < 0 ? 0 : delay; pa.local = 1; pa.nx = nx; pa.nbt = nbt; pa.hsu = 16; pa.nsl = nsl; pa.tagmode = 1;
    const int wpx = (D * nbt + nx - 1) / nx * nsl;
    pa.wpx = wpx; pa.tickets = flags + (fl_bytes - 256) / 4;
    dim3 gp = dim3(nx * (wpx + 4), 1, 1);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(flags, 0, fl_bytes, st)); CK(hipMemsetAsync(status, 0, 4, st)); CK(hipMemsetAsync(hx, 0, hx_bytes, st)); CK(hipMemsetAsync(stats, 0, sizeof(h), st));
      hipEventRecord(e0, st);
      if (delay < 0) {
        if (nsl <= 12) hipLaunchKernelGGL((rnn_bwd_scatter<1, 1, true, 0>), gp, dim3(1024), 0, st, pa);
        else if (nsl <= 24) hipLaunchKernelGGL((rnn_bwd_scatter<2, 1, true, 0>), gp, dim3(1024), 0, st, pa);
        // (more than 24 slices: rnn_bwd_scatter has no instantiation -- three tiles per wave spilled the W_hh fragments, 4.9 us per step at H = 512)
      } else {
        if (NEWV == 4) {
          if (nsl <= 12) hipLaunchKernelGGL((rnn_bwd_scatter2<4, 3, 0>), gp, dim3(512), 0, st, pa);
          else if (nsl <= 20) hipLaunchKernelGGL((rnn_bwd_scatter2<4, 5, 0>), gp, dim3(512), 0, st, pa);
          else if (nsl <= 24) hipLaunchKernelGGL((rnn_bwd_scatter2<4, 6, 0>), gp, dim3(512), 0, st, pa);
          else hipLaunchKernelGGL((rnn_bwd_scatter2<4, 8, 0>), gp, dim3(512), 0, st, pa);
        } else {
          if (nsl <= 16) hipLaunchKernelGGL((rnn_bwd_scatter2<8, 2, 0>), gp, dim3(768), 0, st, pa);
          else if (nsl <= 24) hipLaunchKernelGGL((rnn_bwd_scatter2<8, 3, 0>), gp, dim3(768), 0, st, pa);
          else if (nsl <= 32) hipLaunchKernelGGL((rnn_bwd_scatter2<8, 4, 0>), gp, dim3(768), 0, st, pa);
          else hipLaunchKernelGGL((rnn_bwd_scatter2<8, 5, 0>), gp, dim3(768), 0, st, pa);
        }
      }
      hipEventRecord(e1, st); CK(hipEventSynchronize(e1));
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int hs = -1; CK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost));
    if (delay < 0) {
      printf("rnn_bwd_scatter (round 2)            %6.3f us/step (status %d)\n", ms * 1e3 / T, hs);
      // the gather formulation (rnn_bwd_persist, the path of H > 384 before round 3), same buffers
      const int kq = (G * H + 255) / 256;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(flags, 0, fl_bytes, st)); CK(hipMemsetAsync(status, 0, 4, st)); CK(hipMemsetAsync(hx, 0, hx_bytes, st));
        pa.tagmode = 0;
        hipEventRecord(e0, st);
        if (kq <= 5) hipLaunchKernelGGL((rnn_bwd_persist<5, 1>), gp, dim3(1024), 0, st, pa);
        else if (kq == 6) hipLaunchKernelGGL((rnn_bwd_persist<6, 1>), gp, dim3(1024), 0, st, pa);
        else hipLaunchKernelGGL((rnn_bwd_persist<8, 1>), gp, dim3(1024), 0, st, pa);
        hipEventRecord(e1, st); CK(hipEventSynchronize(e1));
      }
      hipEventElapsedTime(&ms, e0, e1);
      CK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost));
      printf("rnn_bwd_persist (gather formulation) %6.3f us/step (status %d)\n", ms * 1e3 / T, hs);
      continue;
    }
    printf("rnn_bwd_scatter2 %d exchange waves, poll delay %2d       %6.3f us/step (status %d)\n", NEWV, delay, ms * 1e3 / T, hs);
    printf("   item wave 0   (cycles/step): delay + poll + sum %6.0f | gate math + stage %5.0f | barrier wait %5.0f | after barrier %5.0f | total %6.0f   [%.2f poll rounds per step]\n",
           (double)h[0] / T, (double)h[1] / T, (double)h[2] / T, (double)(h[3] + h[5]) / T, (double)h[6] / T, (double)h[4] / T);
    printf("   exchange wave, inside 'lds + mfma + scatter issue': barrier -> A fragments in registers %5.0f | chains + tags + block stores issued %5.0f\n", (double)h[16] / T, (double)h[18] / T);
    printf("   exchange wave (cycles/step): vmcnt wait %6.0f | lgkm %5.0f | barrier wait %5.0f | lds + mfma + scatter issue %5.0f | reserve store + dma issue %5.0f | total %6.0f\n",
           (double)h[8] / T, (double)h[9] / T, (double)h[10] / T, (double)h[11] / T, (double)h[13] / T, (double)h[14] / T);
  }
  return 0;
}
