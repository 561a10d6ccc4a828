// conv.hip -- direct Conv2d (bias, NCHW) forward / data-gradient / weight-gradient for the CNN front-end (gfx950).
//
// replaces: nn.Conv2d in LayerCNN (reference timit/models/model_ctc.py:46,61; geometry from
// timit/conf/ctc_config.yaml:33-37: 3x3, 1->32 stride (1,2), 32->32 stride (2,2), pad (1,1)) and its backward.
// Arithmetic: SURVEY Appendix A.5.  The front-end is ~0.2 % of the FLOPs of the training step
// (92 KFLOP/frame against 50 MFLOP/frame), so these kernels are written for coalesced HBM traffic and
// LDS-resident weights rather than for MFMA:
//   fwd : one lane per output position (b,t',f'), 16 output channels per pass in registers; the whole
//         filter bank (<= 60 KiB) sits in LDS and is read by broadcast; lanes are adjacent in f' so input
//         reads and output writes are coalesced rows.
//   dgrad: same shape, gather form over the (co, tap) pairs that hit an output position.
//   wgrad: positions are tiled 16 at a time into LDS (input patch + dy vector), every thread owns a fixed
//         set of filter taps and accumulates over the workgroup's chunk of positions; per-chunk partials
//         are reduced in a fixed order by a second pass (deterministic, no atomics).
#include <algorithm>

#include "common.h"

namespace {

struct ConvGeom {
  int B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw, Ho, Wo;
};

constexpr int CCH = 16;   // channels per register pass

__global__ __launch_bounds__(256) void conv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ y, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [(ci*KK+tap)][Co]
  const int KK = g.kh * g.kw, CK = g.Ci * KK;
  for (int i = threadIdx.x; i < g.Co * CK; i += 256) {
    const int co = i / CK, r = i - co * CK;
    ws[r * g.Co + co] = w[i];
  }
  __syncthreads();
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  for (size_t pos = blockIdx.x * (size_t)256 + threadIdx.x; pos < npos; pos += (size_t)gridDim.x * 256) {
    const int fo = pos % g.Wo;
    const size_t q = pos / g.Wo;
    const int to = q % g.Ho, b = q / g.Ho;
    for (int co0 = 0; co0 < g.Co; co0 += CCH) {
      float acc[CCH];
#pragma unroll
      for (int c = 0; c < CCH; ++c) acc[c] = (co0 + c < g.Co && bias) ? bias[co0 + c] : 0.0f;
      for (int ci = 0; ci < g.Ci; ++ci)
        for (int i = 0; i < g.kh; ++i) {
          const int ti = to * g.sh - g.ph + i;
          if (ti < 0 || ti >= g.Hi) continue;
          for (int j = 0; j < g.kw; ++j) {
            const int fi = fo * g.sw - g.pw + j;
            if (fi < 0 || fi >= g.Wi) continue;
            const float xv = x[(((size_t)b * g.Ci + ci) * g.Hi + ti) * g.Wi + fi];
            const float *wr = ws + (size_t)(ci * KK + i * g.kw + j) * g.Co + co0;
#pragma unroll
            for (int c = 0; c < CCH; ++c)
              if (co0 + c < g.Co) acc[c] = fmaf(xv, wr[c], acc[c]);
          }
        }
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (co0 + c < g.Co) y[(((size_t)b * g.Co + co0 + c) * g.Ho + to) * g.Wo + fo] = acc[c];
    }
  }
}

__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                         float *__restrict__ dx, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [(co*KK+tap)][Ci]
  const int KK = g.kh * g.kw;
  for (int i = threadIdx.x; i < g.Co * g.Ci * KK; i += 256) {
    const int co = i / (g.Ci * KK), r = i - co * g.Ci * KK;
    const int ci = r / KK, tap = r - ci * KK;
    ws[(size_t)(co * KK + tap) * g.Ci + ci] = w[i];
  }
  __syncthreads();
  const size_t npos = (size_t)g.B * g.Hi * g.Wi;
  for (size_t pos = blockIdx.x * (size_t)256 + threadIdx.x; pos < npos; pos += (size_t)gridDim.x * 256) {
    const int fi = pos % g.Wi;
    const size_t q = pos / g.Wi;
    const int ti = q % g.Hi, b = q / g.Hi;
    for (int ci0 = 0; ci0 < g.Ci; ci0 += CCH) {
      float acc[CCH];
#pragma unroll
      for (int c = 0; c < CCH; ++c) acc[c] = 0.0f;
      for (int i = 0; i < g.kh; ++i) {
        const int tn = ti + g.ph - i;
        if (tn < 0 || tn % g.sh != 0) continue;
        const int to = tn / g.sh;
        if (to >= g.Ho) continue;
        for (int j = 0; j < g.kw; ++j) {
          const int fn = fi + g.pw - j;
          if (fn < 0 || fn % g.sw != 0) continue;
          const int fo = fn / g.sw;
          if (fo >= g.Wo) continue;
          for (int co = 0; co < g.Co; ++co) {
            const float dv = dy[(((size_t)b * g.Co + co) * g.Ho + to) * g.Wo + fo];
            const float *wr = ws + (size_t)(co * KK + i * g.kw + j) * g.Ci + ci0;
#pragma unroll
            for (int c = 0; c < CCH; ++c)
              if (ci0 + c < g.Ci) acc[c] = fmaf(dv, wr[c], acc[c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (ci0 + c < g.Ci) dx[(((size_t)b * g.Ci + ci0 + c) * g.Hi + ti) * g.Wi + fi] = acc[c];
    }
  }
}

constexpr int WG_P = 16;     // positions staged per LDS tile
constexpr int WPT = 40;      // filter taps per thread (Co*Ci*kh*kw <= 256*40)

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         float *__restrict__ part /*[chunks][Wn + Co]*/, ConvGeom g,
                                                         int pos_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KK = g.kh * g.kw, CK = g.Ci * KK, Wn = g.Co * CK;
  float *xs = sm;                 // [WG_P][CK]
  float *ds = sm + WG_P * CK;     // [WG_P][Co]
  const int tid = threadIdx.x;
  const int nk = (Wn + 255) / 256;
  int my_co[WPT], my_r[WPT];
  float acc[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int wi = tid + 256 * k;
    my_co[k] = wi < Wn ? wi / CK : 0;
    my_r[k] = wi < Wn ? wi - my_co[k] * CK : 0;
    acc[k] = 0.0f;
  }
  float bacc = 0.0f;
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  const size_t p0 = (size_t)blockIdx.x * pos_per_chunk;
  const size_t p1 = (p0 + (size_t)pos_per_chunk < npos) ? p0 + (size_t)pos_per_chunk : npos;
  for (size_t pb = p0; pb < p1; pb += WG_P) {
    const int np = (p1 - pb) < (size_t)WG_P ? (int)(p1 - pb) : WG_P;
    __syncthreads();
    for (int i = tid; i < np * CK; i += 256) {
      const int p = i / CK, r = i - p * CK;
      const int ci = r / KK, tap = r - ci * KK;
      const int ki = tap / g.kw, kj = tap - ki * g.kw;
      const size_t pos = pb + p;
      const int fo = pos % g.Wo;
      const size_t q = pos / g.Wo;
      const int to = q % g.Ho, b = q / g.Ho;
      const int ti = to * g.sh - g.ph + ki, fi = fo * g.sw - g.pw + kj;
      xs[p * CK + r] = (ti >= 0 && ti < g.Hi && fi >= 0 && fi < g.Wi) ? x[(((size_t)b * g.Ci + ci) * g.Hi + ti) * g.Wi + fi] : 0.0f;
    }
    for (int i = tid; i < np * g.Co; i += 256) {
      const int p = i / g.Co, co = i - p * g.Co;
      const size_t pos = pb + p;
      const int fo = pos % g.Wo;
      const size_t q = pos / g.Wo;
      const int to = q % g.Ho, b = q / g.Ho;
      ds[p * g.Co + co] = dy[(((size_t)b * g.Co + co) * g.Ho + to) * g.Wo + fo];
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
#pragma unroll
      for (int k = 0; k < WPT; ++k)
        if (k < nk) acc[k] = fmaf(ds[p * g.Co + my_co[k]], xs[p * CK + my_r[k]], acc[k]);
      if (tid < g.Co) bacc += ds[p * g.Co + tid];
    }
  }
  float *out = part + (size_t)blockIdx.x * (Wn + g.Co);
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int wi = tid + 256 * k;
    if (k < nk && wi < Wn) out[wi] = acc[k];
  }
  if (tid < g.Co) out[Wn + tid] = bacc;
}

__global__ void conv_wgrad_reduce_kernel(const float *__restrict__ part, int chunks, int Wn, int Co, float *__restrict__ dw,
                                         float *__restrict__ dbias, float beta_acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Wn + Co) return;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) s += (double)part[(size_t)c * (Wn + Co) + i];
  float *dst = i < Wn ? dw + i : (dbias ? dbias + (i - Wn) : nullptr);
  if (!dst) return;
  *dst = (float)s + (beta_acc != 0.0f ? beta_acc * *dst : 0.0f);
}

int make_geom(ConvGeom &g, int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
  g = ConvGeom{B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw, 0, 0};
  if (B <= 0 || Ci <= 0 || Hi <= 0 || Wi <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return -1;
  g.Ho = (Hi + 2 * ph - kh) / sh + 1;
  g.Wo = (Wi + 2 * pw - kw) / sw + 1;
  if (g.Ho <= 0 || g.Wo <= 0) return -1;
  if ((size_t)Co * Ci * kh * kw * sizeof(float) > 60 * 1024) return -2;
  if (Co * Ci * kh * kw > 256 * WPT) return -2;
  return 0;
}
int wgrad_chunks(const ConvGeom &g) {
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  return (int)std::max((size_t)1, std::min((size_t)512, ceil_div_z(npos, 64)));
}

}  // namespace

extern "C" size_t ctcn_conv2d_ws_bytes(int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
  ConvGeom g;
  if (make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw)) return 0;
  return (size_t)wgrad_chunks(g) * (Co * Ci * kh * kw + Co) * sizeof(float);
}

extern "C" int ctcn_conv2d_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Ci, int Hi, int Wi, int Co,
                               int kh, int kw, int sh, int sw, int ph, int pw, void *stream) {
  ConvGeom g;
  const int rc = make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw);
  CTCN_REQUIRE(rc != -1 && x && w && y, "ctcn_conv2d_fwd: bad args");
  if (rc == -2) { ctcn_set_error("ctcn_conv2d_fwd: filter bank %dx%dx%dx%d too large for the LDS-resident kernel", Co, Ci, kh, kw); return CTCN_EUNSUPPORTED; }
  const size_t npos = (size_t)B * g.Ho * g.Wo;
  const int blocks = (int)std::min((size_t)2048, ceil_div_z(npos, 256));
  const size_t sm = (size_t)Co * Ci * kh * kw * sizeof(float);
  hipLaunchKernelGGL(conv_fwd_kernel, dim3(blocks), dim3(256), sm, (hipStream_t)stream, x, w, bias, y, g);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_conv2d_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw, float *dbias, int B, int Ci,
                               int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw, float beta_acc, void *ws,
                               size_t ws_bytes, void *stream) {
  ConvGeom g;
  const int rc = make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw);
  CTCN_REQUIRE(rc != -1 && x && w && dy && dw && ws, "ctcn_conv2d_bwd: bad args");
  if (rc == -2) { ctcn_set_error("ctcn_conv2d_bwd: filter bank too large for the LDS-resident kernel"); return CTCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int Wn = Co * Ci * kh * kw;
  if (dx) {
    const size_t npos = (size_t)B * Hi * Wi;
    const int blocks = (int)std::min((size_t)2048, ceil_div_z(npos, 256));
    hipLaunchKernelGGL(conv_dgrad_kernel, dim3(blocks), dim3(256), (size_t)Wn * sizeof(float), st, dy, w, dx, g);
    CTCN_LAUNCH_CHECK();
  }
  const int chunks = wgrad_chunks(g);
  if (ws_bytes < (size_t)chunks * (Wn + Co) * sizeof(float)) { ctcn_set_error("ctcn_conv2d_bwd: workspace too small"); return CTCN_EWORKSPACE; }
  const size_t npos = (size_t)B * g.Ho * g.Wo;
  const int ppc = (int)ceil_div_z(npos, chunks);
  const int nch = (int)ceil_div_z(npos, ppc);
  const size_t sm = (size_t)WG_P * (Ci * kh * kw + Co) * sizeof(float);
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(nch), dim3(256), sm, st, x, dy, (float *)ws, g, ppc);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(ceil_div(Wn + Co, 256)), dim3(256), 0, st, (const float *)ws, nch, Wn, Co, dw, dbias, beta_acc);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
