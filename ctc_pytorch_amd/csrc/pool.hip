// pool.hip -- MaxPool2d(kernel = stride = (kh, kw), no padding, floor) forward / backward for the CNN front-end (gfx950).
//
// replaces: nn.MaxPool2d(pooling_size) in LayerCNN (reference timit/models/model_ctc.py:52-53,64-65; off in ctc_config.yaml:38,
// reachable through the `pooling` list of the YAML, train_ctc.py:107-108).  Windows do not overlap, so both passes are pure
// streaming kernels (HBM-bound: 4 B in + 4/(kh*kw) B out per input element, + 1 B of window offset per output):
//   fwd: one lane per OUTPUT element, lanes adjacent along w' -> each window row is a contiguous kw-float read; the winner's
//        offset inside the window (ki*kw + kj, one byte) is kept for the backward pass.  torch's scan rule: the first maximum wins,
//        a NaN replaces anything (`val > max || isnan(val)`).
//   bwd: one lane per INPUT element (gather form, no atomics, no memset): dx = dy[window] if this element won its window else 0;
//        rows / columns beyond the last full window get 0.
#include <algorithm>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned char *__restrict__ arg,
                                                            size_t nout, int Hi, int Wi, int Ho, int Wo, int kh, int kw) {
  for (size_t o = blockIdx.x * (size_t)256 + threadIdx.x; o < nout; o += (size_t)gridDim.x * 256) {
    const int wo = (int)(o % Wo);
    const size_t q = o / Wo;
    const int ho = (int)(q % Ho);
    const size_t plane = q / Ho;
    const float *src = x + (plane * Hi + (size_t)ho * kh) * Wi + (size_t)wo * kw;
    float best = src[0];
    int at = 0;
    for (int i = 0; i < kh; ++i)
      for (int j = 0; j < kw; ++j) {
        const float v = src[(size_t)i * Wi + j];
        if (v > best || v != v) { best = v; at = i * kw + j; }
      }
    y[o] = best;
    arg[o] = (unsigned char)at;
  }
}

__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const float *__restrict__ dy, const unsigned char *__restrict__ arg, float *__restrict__ dx,
                                                            size_t nin, int Hi, int Wi, int Ho, int Wo, int kh, int kw) {
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nin; e += (size_t)gridDim.x * 256) {
    const int w = (int)(e % Wi);
    const size_t q = e / Wi;
    const int h = (int)(q % Hi);
    const size_t plane = q / Hi;
    const int ho = h / kh, wo = w / kw;
    float g = 0.0f;
    if (ho < Ho && wo < Wo) {
      const size_t o = (plane * Ho + ho) * Wo + wo;
      if ((int)arg[o] == (h - ho * kh) * kw + (w - wo * kw)) g = dy[o];
    }
    dx[e] = g;
  }
}

}  // namespace

extern "C" int ctcn_maxpool2d_fwd(const float *x, float *y, unsigned char *arg, size_t planes, int Hi, int Wi, int kh, int kw, void *stream) {
  CTCN_REQUIRE(x && y && arg && planes > 0 && Hi > 0 && Wi > 0 && kh > 0 && kw > 0, "ctcn_maxpool2d_fwd: bad args");
  CTCN_REQUIRE(kh * kw <= 256, "ctcn_maxpool2d_fwd: window %dx%d has more than 256 elements", kh, kw);
  const int Ho = Hi / kh, Wo = Wi / kw;
  CTCN_REQUIRE(Ho > 0 && Wo > 0, "ctcn_maxpool2d_fwd: window %dx%d larger than the %dx%d input", kh, kw, Hi, Wi);
  const size_t nout = planes * Ho * Wo;
  const int blocks = (int)std::min((size_t)8192, ceil_div_z(nout, 256));
  hipLaunchKernelGGL(maxpool2d_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, arg, nout, Hi, Wi, Ho, Wo, kh, kw);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_maxpool2d_bwd(const float *dy, const unsigned char *arg, float *dx, size_t planes, int Hi, int Wi, int kh, int kw, void *stream) {
  CTCN_REQUIRE(dy && dx && arg && planes > 0 && Hi > 0 && Wi > 0 && kh > 0 && kw > 0 && kh * kw <= 256, "ctcn_maxpool2d_bwd: bad args");
  const int Ho = Hi / kh, Wo = Wi / kw;
  CTCN_REQUIRE(Ho > 0 && Wo > 0, "ctcn_maxpool2d_bwd: window larger than the input");
  const size_t nin = planes * Hi * Wi;
  const int blocks = (int)std::min((size_t)8192, ceil_div_z(nin, 256));
  hipLaunchKernelGGL(maxpool2d_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, arg, dx, nin, Hi, Wi, Ho, Wo, kh, kw);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
