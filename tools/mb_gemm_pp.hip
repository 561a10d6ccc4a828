// tools/mb_gemm_pp.hip -- phase accounting of the 256 x 256 ping-pong plane tile (gemm_planes_nt256pp_kernel<2>): in-kernel clock64 sums of the
// cycles waves 0 (half A) and 4 (half B) spend working before each of the four barriers of a stage and waiting inside it (development aid).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mb_gemm_pp.bin tools/mb_gemm_pp.hip ctc_pytorch_amd/csrc/core.hip
#define CTCN_GEMM_STATS 1
#include "../ctc_pytorch_amd/csrc/gemm.hip"
#include <vector>
extern "C" int ctcn_device_xcds(void) { return 8; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 76800, N = argc > 2 ? atoi(argv[2]) : 3072, Kp = argc > 3 ? atoi(argv[3]) : 1024;
  unsigned short *ah, *al, *bh, *bl; float *C; long long *stats, h[64];
  CK(hipMalloc(&ah, (size_t)M * Kp * 2)); CK(hipMalloc(&al, (size_t)M * Kp * 2)); CK(hipMalloc(&bh, (size_t)N * Kp * 2)); CK(hipMalloc(&bl, (size_t)N * Kp * 2));
  CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&stats, sizeof(h))); CK(hipMemset(stats, 0, sizeof(h)));
  {   // random finite bf16 bit patterns (sign, exponent 120..127, random mantissa): the clocks depend on the data
    std::vector<unsigned short> v((size_t)M * Kp);
    unsigned x = 12345u;
    auto fill = [&](unsigned short *d, size_t n) { for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; v[i] = (unsigned short)(((x >> 16) & 0x807f) | ((120 + ((x >> 8) & 7)) << 7)); } return hipMemcpy(d, v.data(), n * 2, hipMemcpyHostToDevice); };
    CK(fill(ah, (size_t)M * Kp)); CK(fill(al, (size_t)M * Kp)); CK(fill(bh, (size_t)N * Kp)); CK(fill(bl, (size_t)N * Kp));
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stats), &stats, sizeof(stats)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tiles_m = (M + 255) / 256;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;        // 0: plane tile <2>; 1: float32-A tile <1>; 2: float32-A tile <2>; 3: TN tile <2> (M, N = the output, Kp = the contraction)
  float *Af = nullptr;
  if (mode) {
    CK(hipMalloc(&Af, (size_t)M * Kp * 4));
    std::vector<float> v((size_t)M * Kp);
    unsigned x = 777u;
    for (auto &f : v) { x = x * 1664525u + 1013904223u; f = ((int)(x >> 8) % 2001 - 1000) * 1e-3f; }
    CK(hipMemcpy(Af, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  }
  float *Bf = nullptr, *part = nullptr; unsigned *queue = nullptr;
  if (mode == 3) {     // TN tile <2>: C (M x N) = A^T B, A: Kp x M, B: Kp x N float32 (contraction-major), split-K over the device
    CK(hipMalloc(&Bf, (size_t)N * Kp * 4)); CK(hipMemset(Bf, 0, (size_t)N * Kp * 4));
    std::vector<float> v((size_t)N * Kp);
    unsigned x = 4242u;
    for (auto &f : v) { x = x * 1664525u + 1013904223u; f = ((int)(x >> 8) % 2001 - 1000) * 1e-3f; }
    CK(hipMemcpy(Bf, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&part, (size_t)64 * M * N * 4)); CK(hipMalloc(&queue, 256));
  }
  for (int dbg : {0, 1}) {
    if (mode && dbg) break;
    const int wnt = mode == 1 ? 1 : 2;
    const int tiles_nn = (N + 128 * wnt - 1) / (128 * wnt);
    const size_t ldsb = (size_t)2 * (2 * 256 * 64 + 2 * 128 * wnt * 64);
    for (int it = 0; it < 3; ++it) {
      if (it == 2) CK(hipEventRecord(e0, st));
      if (mode == 3) {
        auto kern = gemm_tn_f32_pp_kernel<2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        const int nt = tiles_m * tiles_nn, splits = std::max(1, 256 / nt), kchunk = ((Kp + splits - 1) / splits + 31) / 32 * 32;
        CK(hipMemsetAsync(queue, 0, 128, st));
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), ldsb, st, M, N, Kp, (const float *)Af, M, (const float *)Bf, N, C, N, 0.0f, kchunk, (Kp + kchunk - 1) / kchunk, part, tiles_m, tiles_nn, 0xffu, queue);
      } else if (mode == 0) {
        auto kern = gemm_planes_nt256pp_kernel<2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_nn), dim3(512), ldsb, st, M, N, Kp, ah, al, bh, bl, C, N, 0.0f, tiles_m, tiles_nn, dbg);
      } else if (mode == 1) {
        auto kern = gemm_planes_nt256pp_af32_kernel<1>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_nn), dim3(512), ldsb, st, M, N, Kp, Kp, (const float *)Af, Kp, bh, bl, C, N, 0.0f, tiles_m, tiles_nn);
      } else {
        auto kern = gemm_planes_nt256pp_af32_kernel<2>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_nn), dim3(512), ldsb, st, M, N, Kp, Kp, (const float *)Af, Kp, bh, bl, C, N, 0.0f, tiles_m, tiles_nn);
      }
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost));
    printf("mode %d M %d N %d Kp %d dbg %d: %.1f us, %.0f TFLOP/s algorithmic, %d tiles on %d CUs\n", mode, M, N, Kp, dbg, ms * 1e3, 2.0 * M * N * Kp / (ms * 1e-3) / 1e12,
           tiles_m * tiles_nn, ctcn_device_cus());
    const char *who[4] = {"block 0 wave 0 (half A)", "block 0 wave 4 (half B)", "block 1000 wave 0 (half A)", "block 1000 wave 4 (half B)"};
    const char *site[4] = {"read (s,0)             ", "multiply 1 (+ DMA / split)", "read (s,1) (+ LDS stores)", "multiply 2 (+ loads)   "};
    for (int w = 0; w < 2; ++w) {
      const long long *o = h + w * 16;
      const double n = (double)o[9];
      if (n <= 0) continue;
      printf("  %s: %.0f cycles per stage over %d stages; whole tile %lld cycles\n", who[w], o[8] / n, (int)n, o[10]);
      for (int i = 0; i < 4; ++i) printf("      %s work %7.0f   barrier wait %7.0f\n", site[i], o[i] / n, o[4 + i] / n);
    }
  }
  return 0;
}
