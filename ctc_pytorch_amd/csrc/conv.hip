// conv.hip -- Conv2d (bias, NCHW) forward / data-gradient / weight-gradient for the CNN front-end (gfx950).
//
// replaces: nn.Conv2d in LayerCNN (reference timit/models/model_ctc.py:46,61; geometry from
// timit/conf/ctc_config.yaml:33-37: 3x3, 1->32 stride (1,2), 32->32 stride (2,2), pad (1,1)) and its backward.
// Arithmetic: SURVEY Appendix A.5.  Two families of kernels:
//   * MFMA implicit GEMMs (the default, second half of this file): raw source window of a 128-position tile in LDS, filter
//     matrix in LDS, v_mfma_f32_16x16x4_f32; dgrad as one dense stride-1 product per stride class; wgrad as a contraction over
//     positions with per-chunk partials reduced in a fixed order (deterministic, no atomics).
//   * direct kernels (first half; option "conv_mfma" = 0, and the fallback for shapes the tile planner rejects):
//     fwd : one lane per output position (b,t',f'), 16 output channels per pass in registers; the whole
//           filter bank (<= 60 KiB) sits in LDS and is read by broadcast; lanes are adjacent in f'.
//     dgrad: same shape, gather form over the (co, tap) pairs that hit an output position.
//     wgrad: positions are tiled 16 at a time into LDS (input patch + dy vector), every thread owns a fixed
//           set of filter taps and accumulates over the workgroup's chunk of positions.
#include <algorithm>

#include "common.h"

namespace {

struct ConvGeom {
  int B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw, Ho, Wo;
};

constexpr int CCH = 16;   // channels per register pass

// (round 6) Filter banks beyond the LDS budget -- the reference's own example front-end, model_ctc.py:232-233: (32, 32, (3, 21)) = 258 KB -- run
// as several launches over (output-channel range, input-channel range) slices of the bank, each slice LDS-resident: the first slice of a
// channel range starts from the bias, the later ones from what the earlier ones left in y.  Channels in ascending order, so the fma chain of
// an output element is the one of the unsliced kernel (its partial sums pass through memory as exact float32 values).
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, float *__restrict__ y, ConvGeom g, int co_lo, int co_hi,
                                                       int ci_lo, int ci_hi) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [((ci - ci_lo)*KK+tap)][co - co_lo]
  const int KK = g.kh * g.kw, CK = g.Ci * KK, nco = co_hi - co_lo, nci = ci_hi - ci_lo, CKc = nci * KK;
  for (int i = threadIdx.x; i < nco * CKc; i += 256) {
    const int co = i / CKc, r = i - co * CKc;
    ws[r * nco + co] = w[(size_t)(co_lo + co) * CK + ci_lo * KK + r];
  }
  __syncthreads();
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  for (size_t pos = blockIdx.x * (size_t)256 + threadIdx.x; pos < npos; pos += (size_t)gridDim.x * 256) {
    const int fo = pos % g.Wo;
    const size_t q = pos / g.Wo;
    const int to = q % g.Ho, b = q / g.Ho;
    for (int co0 = 0; co0 < nco; co0 += CCH) {
      float acc[CCH];
#pragma unroll
      for (int c = 0; c < CCH; ++c) {
        acc[c] = 0.0f;
        if (co0 + c < nco) {
          if (ci_lo == 0) acc[c] = bias ? bias[co_lo + co0 + c] : 0.0f;
          else acc[c] = y[(((size_t)b * g.Co + co_lo + co0 + c) * g.Ho + to) * g.Wo + fo];
        }
      }
      for (int ci = 0; ci < nci; ++ci)
        for (int i = 0; i < g.kh; ++i) {
          const int ti = to * g.sh - g.ph + i;
          if (ti < 0 || ti >= g.Hi) continue;
          for (int j = 0; j < g.kw; ++j) {
            const int fi = fo * g.sw - g.pw + j;
            if (fi < 0 || fi >= g.Wi) continue;
            const float xv = x[(((size_t)b * g.Ci + ci_lo + ci) * g.Hi + ti) * g.Wi + fi];
            const float *wr = ws + (size_t)(ci * KK + i * g.kw + j) * nco + co0;
#pragma unroll
            for (int c = 0; c < CCH; ++c)
              if (co0 + c < nco) acc[c] = fmaf(xv, wr[c], acc[c]);
          }
        }
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (co0 + c < nco) y[(((size_t)b * g.Co + co_lo + co0 + c) * g.Ho + to) * g.Wo + fo] = acc[c];
    }
  }
}

// (slices as above: the first output-channel range of an input-channel range starts dx from zero, the later ones accumulate)
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                         float *__restrict__ dx, ConvGeom g, int co_lo, int co_hi, int ci_lo, int ci_hi) {
  extern __shared__ __attribute__((aligned(16))) float ws[];   // [((co - co_lo)*KK+tap)][ci - ci_lo]
  const int KK = g.kh * g.kw, nco = co_hi - co_lo, nci = ci_hi - ci_lo;
  for (int i = threadIdx.x; i < nco * nci * KK; i += 256) {
    const int co = i / (nci * KK), r = i - co * nci * KK;
    const int ci = r / KK, tap = r - ci * KK;
    ws[(size_t)(co * KK + tap) * nci + ci] = w[((size_t)(co_lo + co) * g.Ci + ci_lo + ci) * KK + tap];
  }
  __syncthreads();
  const size_t npos = (size_t)g.B * g.Hi * g.Wi;
  for (size_t pos = blockIdx.x * (size_t)256 + threadIdx.x; pos < npos; pos += (size_t)gridDim.x * 256) {
    const int fi = pos % g.Wi;
    const size_t q = pos / g.Wi;
    const int ti = q % g.Hi, b = q / g.Hi;
    for (int ci0 = 0; ci0 < nci; ci0 += CCH) {
      float acc[CCH];
#pragma unroll
      for (int c = 0; c < CCH; ++c) acc[c] = (co_lo != 0 && ci0 + c < nci) ? dx[(((size_t)b * g.Ci + ci_lo + ci0 + c) * g.Hi + ti) * g.Wi + fi] : 0.0f;
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
          for (int co = 0; co < nco; ++co) {
            const float dv = dy[(((size_t)b * g.Co + co_lo + co) * g.Ho + to) * g.Wo + fo];
            const float *wr = ws + (size_t)(co * KK + i * g.kw + j) * nci + ci0;
#pragma unroll
            for (int c = 0; c < CCH; ++c)
              if (ci0 + c < nci) acc[c] = fmaf(dv, wr[c], acc[c]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CCH; ++c)
        if (ci0 + c < nci) dx[(((size_t)b * g.Ci + ci_lo + ci0 + c) * g.Hi + ti) * g.Wi + fi] = acc[c];
    }
  }
}

constexpr int WG_P = 16;     // positions staged per LDS tile
constexpr int WPT = 40;      // filter taps per thread (Co*Ci*kh*kw <= 256*40)

// (slices as above: a launch owns the filter taps of one (output-channel range, input-channel range) and writes their columns of `part`;
// the bias column of an output channel comes from the launch whose input-channel range starts at 0)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         float *__restrict__ part /*[chunks][Wn + Co]*/, ConvGeom g,
                                                         int pos_per_chunk, int co_lo, int co_hi, int ci_lo, int ci_hi) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KK = g.kh * g.kw, CK = g.Ci * KK, Wn = g.Co * CK;
  const int nco = co_hi - co_lo, nci = ci_hi - ci_lo, CKc = nci * KK, Wc = nco * CKc;
  float *xs = sm;                 // [WG_P][CKc]
  float *ds = sm + WG_P * CKc;    // [WG_P][nco]
  const int tid = threadIdx.x;
  const int nk = (Wc + 255) / 256;
  int my_co[WPT], my_r[WPT];
  float acc[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int wi = tid + 256 * k;
    my_co[k] = wi < Wc ? wi / CKc : 0;
    my_r[k] = wi < Wc ? wi - my_co[k] * CKc : 0;
    acc[k] = 0.0f;
  }
  float bacc = 0.0f;
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  const size_t p0 = (size_t)blockIdx.x * pos_per_chunk;
  const size_t p1 = (p0 + (size_t)pos_per_chunk < npos) ? p0 + (size_t)pos_per_chunk : npos;
  for (size_t pb = p0; pb < p1; pb += WG_P) {
    const int np = (p1 - pb) < (size_t)WG_P ? (int)(p1 - pb) : WG_P;
    __syncthreads();
    for (int i = tid; i < np * CKc; i += 256) {
      const int p = i / CKc, r = i - p * CKc;
      const int ci = ci_lo + r / KK, tap = r % KK;
      const int ki = tap / g.kw, kj = tap - ki * g.kw;
      const size_t pos = pb + p;
      const int fo = pos % g.Wo;
      const size_t q = pos / g.Wo;
      const int to = q % g.Ho, b = q / g.Ho;
      const int ti = to * g.sh - g.ph + ki, fi = fo * g.sw - g.pw + kj;
      xs[p * CKc + r] = (ti >= 0 && ti < g.Hi && fi >= 0 && fi < g.Wi) ? x[(((size_t)b * g.Ci + ci) * g.Hi + ti) * g.Wi + fi] : 0.0f;
    }
    for (int i = tid; i < np * nco; i += 256) {
      const int p = i / nco, co = i - p * nco;
      const size_t pos = pb + p;
      const int fo = pos % g.Wo;
      const size_t q = pos / g.Wo;
      const int to = q % g.Ho, b = q / g.Ho;
      ds[p * nco + co] = dy[(((size_t)b * g.Co + co_lo + co) * g.Ho + to) * g.Wo + fo];
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
#pragma unroll
      for (int k = 0; k < WPT; ++k)
        if (k < nk) acc[k] = fmaf(ds[p * nco + my_co[k]], xs[p * CKc + my_r[k]], acc[k]);
      if (tid < nco) bacc += ds[p * nco + tid];
    }
  }
  float *out = part + (size_t)blockIdx.x * (Wn + g.Co);
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int wi = tid + 256 * k;
    if (k < nk && wi < Wc) out[(size_t)(co_lo + my_co[k]) * CK + ci_lo * KK + my_r[k]] = acc[k];
  }
  if (ci_lo == 0 && tid < nco) out[Wn + co_lo + tid] = bacc;
}

// ================================================================================================
// MFMA implicit-GEMM kernels (the default; the direct kernels above remain for filter banks whose LDS images do not fit).
//   forward : y[pos, co]  = sum_{ci,tap} x[src(pos, tap), ci] * w[co, ci, tap]          pos = (b, t', f')
//   dgrad   : dx[pos, ci] = sum_{co,tap} dy[src(pos, tap), co] * w[co, ci, tap]         pos = (b, t, f), one launch per stride class
//             (a, b) = ((t + ph) mod sh, (f + pw) mod sw): inside a class only the taps ki = a (mod sh), kj = b (mod sw) reach an
//             output position and they do so at unit stride, so the class is a dense stride-1 convolution with ~KK / (sh*sw) taps
//   wgrad   : dw[co, (ci,tap)] = sum_pos dy[pos, co] * x[src(pos, tap), ci]              contraction over positions, per-chunk partials
// A workgroup owns a tile of R x C positions (R*C <= 128) of one image.  It stages the RAW source window of the tile -- Cs channels
// x RH rows x RW columns, zero-filled outside the image -- in LDS with row-contiguous loads (no im2col copy: every source element is
// read from HBM/L2 once per tile and the 9x tap reuse happens in LDS), and each of its 8 waves multiplies 16 positions by all
// output channels with v_mfma_f32_16x16x4_f32, reading the A operand at raw[koff[k] + posoff[position]].  The arithmetic is exact
// float32 (fma chain in ascending k): both matmul "precisions" of the library give the same convolution.  Outputs go through an
// LDS transpose so that stores are rows of consecutive positions per channel.
// ================================================================================================
constexpr int TP = 128;                 // positions per tile (8 waves x 16)
constexpr int CONV_THREADS = 512;
constexpr int MAX_TAPS = 25;

struct ConvTile {
  int Cs, Hs, Ws;                       // source tensor: channels, rows, columns
  int N, Hd, Wd;                        // destination tensor
  int Hc, Wc;                           // positions of this class per image
  int a, b, osh, osw;                   // destination coordinates of class position (hc, wc): (a + hc*osh, b + wc*osw)
  int ssh, ssw, dh0, dw0;               // source coordinates of tap (dh, dw) at (hc, wc): (hc*ssh + dh0 + dh, wc*ssw + dw0 + dw)
  int R, C, RH, RW;                     // tile = R x C positions; raw window = RH x RW per channel
  int ntap, K, K4;                      // taps of the class, K = Cs*ntap, K4 = K rounded up to the MFMA k-step
  int wsn, wsc;                         // weight address: w[n*wsn + c*wsc + tap index]
  int tiles_h, tiles_w, ntiles;
  int tap[MAX_TAPS];                    // (dh << 16) | (dw << 8) | (ki*kw + kj), dh / dw relative to (dh0, dw0)
};

__host__ __device__ inline int conv_ws_stride(int nt) { return nt == 1 ? 16 : (nt <= 3 ? 48 : 80); }   // k-rows of an MFMA step in disjoint banks

__device__ __forceinline__ void conv_fill_koff(int *koff, const ConvTile &m) {
  for (int k = threadIdx.x; k < m.K4; k += CONV_THREADS) {
    int v = 0;
    if (k < m.K) {
      const int c = k / m.ntap, t = m.tap[k - c * m.ntap];
      v = (c * m.RH + (t >> 16)) * m.RW + ((t >> 8) & 255);
    }
    koff[k] = v;
  }
}

#ifndef CONV_LOAD_DEPTH
#define CONV_LOAD_DEPTH 8
#endif
// raw[c][rh][rw] = src[b, c, h0 + rh, w0 + rw] (0 outside the tensor); a group of RWP lanes walks one row.  Loads are issued
// eight rows at a time from clamped (always valid) addresses and masked afterwards, so that eight HBM/L2 latencies overlap.
__device__ __forceinline__ void conv_load_raw(float *raw, const float *__restrict__ src, const ConvTile &m, int b, int h0, int w0, int rwp_log) {
  constexpr int U = CONV_LOAD_DEPTH;
  const int tid = threadIdx.x, rw = tid & ((1 << rwp_log) - 1), grp = tid >> rwp_log, ngrp = CONV_THREADS >> rwp_log;
  if (rw >= m.RW) return;
  const int rows = m.Cs * m.RH, ws = w0 + rw;
  const bool col_ok = ws >= 0 && ws < m.Ws;
  const int wsc = min(max(ws, 0), m.Ws - 1);
  const float *base = src + (size_t)b * m.Cs * m.Hs * m.Ws + wsc;
  int c = grp / m.RH, rh = grp - c * m.RH;
  const int dc = ngrp / m.RH, drh = ngrp - dc * m.RH;
  for (int row0 = grp; row0 < rows; row0 += ngrp * U) {
    float v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int hs = h0 + rh, cc = min(c, m.Cs - 1), hc = min(max(hs, 0), m.Hs - 1);
      ok[u] = col_ok && hs >= 0 && hs < m.Hs && row0 + u * ngrp < rows;
      v[u] = base[((size_t)cc * m.Hs + hc) * m.Ws];
      c += dc; rh += drh;
      if (rh >= m.RH) { rh -= m.RH; ++c; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (row0 + u * ngrp < rows) raw[(row0 + u * ngrp) * m.RW + rw] = ok[u] ? v[u] : 0.0f;
  }
}

__device__ __forceinline__ void conv_tile_origin(const ConvTile &m, int tile, int &b, int &hc0, int &wc0) {
  const int per_img = m.tiles_h * m.tiles_w;
  b = tile / per_img;
  const int rem = tile - b * per_img, th = rem / m.tiles_w;
  hc0 = th * m.R;
  wc0 = (rem - th * m.tiles_w) * m.C;
}

template <int NT>
__global__ __launch_bounds__(CONV_THREADS) void conv_mfma_kernel(const float *__restrict__ src, const float *__restrict__ w, const float *__restrict__ bias,
                                                                 float *__restrict__ out, ConvTile m, int rwp_log, int region_floats, int dbg) {
  // dbg (option "conv_dbg", development: tools/conv_phase_probe.py): 1 = no window load, 2 = no MFMA loop, 4 = no output phase (results invalid)
  extern __shared__ __attribute__((aligned(16))) float csm[];   // region[max(raw, transpose)] | wmat[K4][WS] | koff[K4]
  constexpr int N16 = 16 * NT, OS = TP + 1;
  const int WS = conv_ws_stride(NT);
  float *raw = csm, *wmat = csm + region_floats;
  int *koff = reinterpret_cast<int *>(wmat + (size_t)m.K4 * WS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  conv_fill_koff(koff, m);
  for (int i = tid; i < m.K4 * N16; i += CONV_THREADS) {
    const int k = i / N16, n = i - k * N16;
    float v = 0.0f;
    if (k < m.K && n < m.N) { const int c = k / m.ntap; v = w[(size_t)n * m.wsn + (size_t)c * m.wsc + (m.tap[k - c * m.ntap] & 255)]; }
    wmat[k * WS + n] = v;
  }
  const int r = lane & 15, q = lane >> 4;
  const int p = wave * 16 + r, ptr_ = p / m.C, ptc = p - ptr_ * m.C;
  const int posoff = ptr_ < m.R ? ptr_ * m.ssh * m.RW + ptc * m.ssw : 0;
  const int RC = m.R * m.C;
  // output phase (round 5): wave v stores channels v, v + 8, ...; a lane owns positions pp = lane, lane + 64 of EVERY tile, so their
  // (row, column) inside the tile and their offset inside a destination plane are computed once per launch -- the phase used to divide
  // twice per stored element (i -> (n, pp) -> (tr, tc)) and was 38 of the 62 us of cfg3's first layer (tools/conv_phase_probe.py)
  int otr[2], otc[2], odo[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int pp = lane + 64 * jj;
    otr[jj] = pp / m.C; otc[jj] = pp - otr[jj] * m.C;
    odo[jj] = (m.a + otr[jj] * m.osh) * m.Wd + m.b + otc[jj] * m.osw;
    if (pp >= RC) otr[jj] = 1 << 20;                                   // never inside the image
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < m.ntiles; tile += gridDim.x) {
    int b, hc0, wc0;
    conv_tile_origin(m, tile, b, hc0, wc0);
    // (eight row loads in flight per thread; 16 / 32 measured 98 / 97 us against 101 at cfg3's second layer -- and three instantiations of the
    // loader in one kernel cost the first layer its occupancy: 39 -> 61 us)
    if (m.K > 0 && !(dbg & 1)) conv_load_raw(raw, src, m, b, hc0 * m.ssh + m.dh0, wc0 * m.ssw + m.dw0, rwp_log);
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // four k-steps per trip: the four table reads, then the four window reads, then the filter reads are independent of each other, so their
    // LDS latencies overlap (one step at a time the chain koff -> raw -> MFMA was ~150 cycles per step with two waves per SIMD to hide it:
    // 45 of the 123 us of cfg3's second layer).  K4 is a multiple of 4; the tail runs step by step.  Same products in the same order.
    const int kend = (dbg & 2) ? 4 : m.K4;
    int k0 = 0;
    for (; k0 + 16 <= kend; k0 += 16) {
      int ko[4];
      float av[4], bv[4][NT];
#pragma unroll
      for (int u = 0; u < 4; ++u) ko[u] = koff[k0 + 4 * u + q];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = raw[ko[u] + posoff];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[u][nt] = wmat[(k0 + 4 * u + q) * WS + nt * 16 + r];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = (k0 + 4 * u + q) < m.K ? av[u] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[u][nt], acc[nt], 0, 0, 0);
      }
    }
    for (; k0 < kend; k0 += 4) {
      const int kk = k0 + q;
      float av = raw[koff[kk] + posoff];
      av = kk < m.K ? av : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wmat[kk * WS + nt * 16 + r], acc[nt], 0, 0, 0);
    }
    __syncthreads();
    if (dbg & 4) { if (acc[0][0] == 12345.678f) out[0] = 1.0f; continue; }
    // C layout: column r (= channel within the 16-tile), rows 4q .. 4q+3 (= positions of this wave's 16) -> ot[n][p]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) raw[(nt * 16 + r) * OS + wave * 16 + 4 * q + e] = acc[nt][e];
    __syncthreads();
    {
      const size_t plane = (size_t)m.Hd * m.Wd;
      float *ob = out + (size_t)b * m.N * plane + (size_t)hc0 * m.osh * m.Wd + (size_t)wc0 * m.osw;
      bool ov[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) ov[jj] = hc0 + otr[jj] < m.Hc && wc0 + otc[jj] < m.Wc;
      for (int n = wave; n < m.N; n += CONV_THREADS / 64) {
        const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          if (ov[jj]) ob[(size_t)n * plane + odo[jj]] = raw[n * OS + lane + 64 * jj] + bn;
      }
    }
    __syncthreads();
  }
}

// wgrad: per chunk of tiles, part[co][k] = sum_pos dy[pos][co] * x[src(pos, k)], and the bias partial as the extra column k = K
// (a patch value of 1).  A = dyT[pos][co] (LDS, zero for positions outside the image), B = raw[koff[k] + posoff[pos]]; wave v owns the
// 16-wide k tiles v, v + 8, ... (JR of them) for all NTC channel tiles.
template <int NTC, int JR>
__global__ __launch_bounds__(CONV_THREADS) void conv_wgrad_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part,
                                                                       ConvTile m, int Co, int rwp_log, int raw_floats, int tiles_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float csm[];   // raw | dyT[TP][16*NTC + 1] | koff[K4] | posoff[TP]
  constexpr int C16 = 16 * NTC, DS = C16 + 1;
  float *raw = csm, *dyT = csm + raw_floats;
  int *koff = reinterpret_cast<int *>(dyT + (size_t)TP * DS);
  int *posoff = koff + m.K4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int K = m.K, Wn = Co * K;
  conv_fill_koff(koff, m);
  if (tid < TP) {
    const int tr = tid / m.C, tc = tid - tr * m.C;
    posoff[tid] = tr < m.R ? tr * m.ssh * m.RW + tc * m.ssw : 0;
  }
  __syncthreads();
  int kbase[JR];
  bool kone[JR];
  f32x4 acc[JR][NTC];
#pragma unroll
  for (int j = 0; j < JR; ++j) {
    const int k = (wave + 8 * j) * 16 + r;
    kbase[j] = k < K ? koff[k] : 0;
    kone[j] = k == K;
#pragma unroll
    for (int ct = 0; ct < NTC; ++ct) acc[j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nkt = (K + 1 + 15) / 16;                         // k tiles in use (with the bias column)
  const int t_begin = blockIdx.x * tiles_per_chunk, t_end = min(m.ntiles, t_begin + tiles_per_chunk);
  const int dp = tid & (TP - 1), dq = tid >> 7;              // dy staging role: position dp, channels dq, dq + 4, ...
  const int dtr = dp / m.C, dtc = dp - dtr * m.C;
  for (int tile = t_begin; tile < t_end; ++tile) {
    int b, hc0, wc0;
    conv_tile_origin(m, tile, b, hc0, wc0);
    conv_load_raw(raw, x, m, b, hc0 * m.ssh + m.dh0, wc0 * m.ssw + m.dw0, rwp_log);
    {
      const int hc = hc0 + dtr, wc = wc0 + dtc;
      const bool pv = dtr < m.R && hc < m.Hc && wc < m.Wc;
      const size_t base = ((size_t)b * Co * m.Hc + min(hc, m.Hc - 1)) * m.Wc + min(wc, m.Wc - 1);
      float v[C16 / 4];
#pragma unroll
      for (int u = 0; u < C16 / 4; ++u) v[u] = dy[base + (size_t)min(dq + 4 * u, Co - 1) * m.Hc * m.Wc];
#pragma unroll
      for (int u = 0; u < C16 / 4; ++u) dyT[dp * DS + dq + 4 * u] = (pv && dq + 4 * u < Co) ? v[u] : 0.0f;
    }
    __syncthreads();
    if (wave < nkt) {
      // two position steps per trip (round 5): the table reads, then the dy and window reads of both steps are independent of each other, so
      // their LDS latencies overlap (the chain posoff -> raw -> MFMA one step at a time, as in conv_mfma_kernel).  TP is a multiple of 8; the
      // products enter every accumulator in the same ascending position order as before.
      for (int s = 0; s < TP; s += 8) {
        int po[2];
        float av[2][NTC], bv[2][JR];
#pragma unroll
        for (int u = 0; u < 2; ++u) po[u] = posoff[s + 4 * u + q];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int ct = 0; ct < NTC; ++ct) av[u][ct] = dyT[(s + 4 * u + q) * DS + ct * 16 + r];
#pragma unroll
          for (int j = 0; j < JR; ++j) bv[u][j] = raw[kbase[j] + po[u]];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < JR; ++j) {
            if (wave + 8 * j < nkt) {
              const float b1 = kone[j] ? 1.0f : bv[u][j];
#pragma unroll
              for (int ct = 0; ct < NTC; ++ct) acc[j][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][ct], b1, acc[j][ct], 0, 0, 0);
            }
          }
      }
    }
    __syncthreads();
  }
  float *outp = part + (size_t)blockIdx.x * (Wn + Co);
#pragma unroll
  for (int j = 0; j < JR; ++j) {
    const int k = (wave + 8 * j) * 16 + r;                   // C layout: column r = k, rows 4q + e = co
#pragma unroll
    for (int ct = 0; ct < NTC; ++ct)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = ct * 16 + 4 * q + e;
        if (co < Co && k < K) outp[(size_t)co * K + k] = acc[j][ct][e];
        else if (co < Co && k == K) outp[Wn + co] = acc[j][ct][e];
      }
  }
}

// fixed-order reduction of the per-chunk partials: 64 elements x 16 chunk groups per workgroup, doubles, LDS tree
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float *__restrict__ part, int chunks, int Wn, int Co, float *__restrict__ dw,
                                                                 float *__restrict__ dbias, float beta_acc) {
  __shared__ double red[16][65];
  const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e, n = Wn + Co;
  double s = 0.0;
  if (i < n)
    for (int c = cg; c < chunks; c += 16) s += (double)part[(size_t)c * n + i];
  red[cg][e] = s;
  __syncthreads();
  if (cg == 0 && i < n) {
    for (int c = 1; c < 16; ++c) s += red[c][e];
    float *dst = i < Wn ? dw + i : (dbias ? dbias + (i - Wn) : nullptr);
    if (dst) *dst = (float)s + (beta_acc != 0.0f ? beta_acc * *dst : 0.0f);
  }
}

int make_geom(ConvGeom &g, int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
  g = ConvGeom{B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw, 0, 0};
  if (B <= 0 || Ci <= 0 || Hi <= 0 || Wi <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return -1;
  g.Ho = (Hi + 2 * ph - kh) / sh + 1;
  g.Wo = (Wi + 2 * pw - kw) / sw + 1;
  if (g.Ho <= 0 || g.Wo <= 0) return -1;
  return 0;
}

// (co, ci) slice sizes of the direct kernels for a bank that is not LDS-resident as a whole (make_geom: -2).  fwd / dgrad: the slice of the
// bank itself within 60 KB; wgrad: WG_P staged patches + dy vectors within 60 KB and at most 256 * WPT taps per launch.  false: a single
// input channel's taps do not fit (kh * kw beyond ~900) -- no kernel here takes that.
struct ConvSlices { int nco, nci, wnco, wnci; };
bool conv_slices(const ConvGeom &g, ConvSlices &sl) {
  const int KK = g.kh * g.kw;
  const size_t budget = 60 * 1024 / sizeof(float), bank = (size_t)g.Co * g.Ci * KK;
  sl.nco = g.Co; sl.nci = g.Ci;
  if (bank > budget) {
    if ((size_t)KK > budget) return false;
    sl.nco = (int)std::min((size_t)g.Co, budget / KK);
    if (sl.nco > CCH) sl.nco = sl.nco / CCH * CCH;                  // whole register passes
    sl.nci = (int)std::max((size_t)1, std::min((size_t)g.Ci, budget / ((size_t)sl.nco * KK)));
  }
  sl.wnco = g.Co; sl.wnci = g.Ci;
  if (bank > (size_t)256 * WPT || (size_t)WG_P * ((size_t)g.Ci * KK + g.Co) > budget) {
    if ((size_t)WG_P * (KK + 1) > budget || KK > 256 * WPT) return false;
    sl.wnci = (int)std::max((size_t)1, std::min((size_t)g.Ci, (budget / WG_P - 1) / KK));
    sl.wnco = std::max(1, std::min(g.Co, 256 * WPT / (sl.wnci * KK)));
    while ((size_t)WG_P * ((size_t)sl.wnci * KK + sl.wnco) > budget && sl.wnco > 1) --sl.wnco;
  }
  return true;
}

// ---- MFMA path: tile plans ---------------------------------------------------------------------------------------------------------
// (round 4, the forward / dgrad kernel at cfg3's layer 2 is 116-140 us against ~16 us of HBM and MFMA floor; three things measured and not kept:
// (1) eight k-steps per trip with every LDS read issued before the first MFMA: 141 -> 136 us, within noise -- the k loop is not the chain;
// (2) the next tile's window prefetched into registers before the MFMA loop and committed to LDS after the stores: 139 -> 142 us forward,
// 283 -> 352 us backward (243 VGPRs, and most workgroups own one or two tiles); (3) smaller LDS images for 2-3 workgroups per CU, below.)
constexpr size_t CONV_LDS_MAX = 158 * 1024;   // (round 4: capping the image at 79 / 52 KB so that two / three workgroups share a CU -- the kernels use 61 VGPRs --
                                               // measured no gain at cfg3 (8.02 / 8.02 / 8.31 ms per step) and a loss at the shipped-YAML shape (4.36 / 4.73 / 5.19):
                                               // the smaller tiles re-read more halo and waste MFMA rows)
inline int round4(int k) { return (k + 3) & ~3; }
inline int log2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct ConvPlan { ConvTile m; int rwp_log; int region_floats; size_t lds; bool ok; };

// mode 0: forward / wgrad position space (output positions, all taps);  mode 1: dgrad class `cls` = a*sw + b
// extra_floats(m): LDS floats next to the raw window (filter matrix, tables) as a function of the finished map
template <class Extra>
ConvPlan plan_tiles(const ConvGeom &g, int mode, int cls, int region_min_floats, Extra extra_floats) {
  ConvPlan P{};
  ConvTile &m = P.m;
  P.ok = false;
  if (g.kh * g.kw > MAX_TAPS || g.kh > 255 || g.kw > 255) return P;
  int dh[MAX_TAPS], dw[MAX_TAPS], ki[MAX_TAPS], kj[MAX_TAPS], nki = 0, nkj = 0;
  if (mode == 0) {
    m.Cs = g.Ci; m.Hs = g.Hi; m.Ws = g.Wi; m.N = g.Co; m.Hd = g.Ho; m.Wd = g.Wo; m.Hc = g.Ho; m.Wc = g.Wo;
    m.a = m.b = 0; m.osh = m.osw = 1; m.ssh = g.sh; m.ssw = g.sw;
    for (int i = 0; i < g.kh; ++i) { ki[nki] = i; dh[nki++] = i - g.ph; }
    for (int j = 0; j < g.kw; ++j) { kj[nkj] = j; dw[nkj++] = j - g.pw; }
    m.wsn = g.Ci * g.kh * g.kw; m.wsc = g.kh * g.kw;
  } else {
    m.Cs = g.Co; m.Hs = g.Ho; m.Ws = g.Wo; m.N = g.Ci; m.Hd = g.Hi; m.Wd = g.Wi;
    // class (ra, rb) = ((t + ph) mod sh, (f + pw) mod sw); its first position is a = (ra - ph) mod sh
    const int ra = cls / g.sw, rb = cls % g.sw;
    m.a = ((ra - g.ph) % g.sh + g.sh) % g.sh; m.b = ((rb - g.pw) % g.sw + g.sw) % g.sw;
    m.osh = g.sh; m.osw = g.sw; m.ssh = m.ssw = 1;
    m.Hc = m.a < g.Hi ? (g.Hi - m.a + g.sh - 1) / g.sh : 0;
    m.Wc = m.b < g.Wi ? (g.Wi - m.b + g.sw - 1) / g.sw : 0;
    for (int i = 0; i < g.kh; ++i) if ((m.a + g.ph - i) % g.sh == 0) { ki[nki] = i; dh[nki++] = (m.a + g.ph - i) / g.sh; }
    for (int j = 0; j < g.kw; ++j) if ((m.b + g.pw - j) % g.sw == 0) { kj[nkj] = j; dw[nkj++] = (m.b + g.pw - j) / g.sw; }
    m.wsn = g.kh * g.kw; m.wsc = g.Ci * g.kh * g.kw;
  }
  int dhs = 1, dws = 1;
  m.dh0 = m.dw0 = 0;
  if (nki && nkj) {
    m.dh0 = *std::min_element(dh, dh + nki); m.dw0 = *std::min_element(dw, dw + nkj);
    dhs = *std::max_element(dh, dh + nki) - m.dh0 + 1; dws = *std::max_element(dw, dw + nkj) - m.dw0 + 1;
  }
  m.ntap = nki * nkj;
  for (int i = 0; i < nki; ++i)
    for (int j = 0; j < nkj; ++j) m.tap[i * nkj + j] = ((dh[i] - m.dh0) << 16) | ((dw[j] - m.dw0) << 8) | (ki[i] * g.kw + kj[j]);
  if (dhs > 255 || dws > 255) return P;
  m.K = m.Cs * m.ntap; m.K4 = round4(m.K);
  if (m.Hc <= 0 || m.Wc <= 0) { m.ntiles = 0; P.ok = true; return P; }
  m.C = std::min(m.Wc, TP);
  for (m.R = std::max(1, std::min(TP / m.C, m.Hc)); m.R >= 1; --m.R) {
    m.RH = (m.R - 1) * m.ssh + dhs; m.RW = (m.C - 1) * m.ssw + dws;
    if (m.RW > CONV_THREADS) return P;
    P.region_floats = std::max(region_min_floats, m.K > 0 ? m.Cs * m.RH * m.RW : 0);
    P.lds = ((size_t)P.region_floats + extra_floats(m)) * sizeof(float);
    if (P.lds <= CONV_LDS_MAX) break;
  }
  if (m.R < 1) return P;
  P.rwp_log = log2_ceil(m.RW);
  m.tiles_h = ceil_div(m.Hc, m.R); m.tiles_w = ceil_div(m.Wc, m.C);
  const size_t nt = (size_t)g.B * m.tiles_h * m.tiles_w;
  if (nt >= ((size_t)1 << 30)) return P;
  m.ntiles = (int)nt;
  P.ok = true;
  return P;
}

ConvPlan plan_gemm(const ConvGeom &g, int mode, int cls) {
  const int N = mode == 0 ? g.Co : g.Ci, NT = ceil_div(N, 16);
  if (N > 64 || ctcn_get_option("conv_mfma") == 0) { ConvPlan P{}; P.ok = false; return P; }
  return plan_tiles(g, mode, cls, 16 * NT * (TP + 1), [NT](const ConvTile &m) { return (size_t)m.K4 * (conv_ws_stride(NT) + 1); });
}

constexpr int WGRAD_JR_MAX = 8;
ConvPlan plan_wgrad(const ConvGeom &g) {
  const int NTC = ceil_div(g.Co, 16);
  ConvPlan P{};
  P.ok = false;
  if (g.Co > 64 || ctcn_get_option("conv_mfma") == 0) return P;
  P = plan_tiles(g, 0, 0, 0, [NTC](const ConvTile &m) { return (size_t)TP * (16 * NTC + 1) + round4(m.K + 1) + TP; });
  if (P.ok) {
    P.m.K4 = round4(P.m.K + 1);                       // + the bias column
    if (ceil_div(ceil_div(P.m.K + 1, 16), 8) > WGRAD_JR_MAX) P.ok = false;
  }
  return P;
}

template <int NT>
int launch_gemm_nt(const float *src, const float *w, const float *bias, float *out, const ConvPlan &P, hipStream_t st) {
  auto kern = conv_mfma_kernel<NT>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds));
  const int blocks = std::min(P.m.ntiles, 2 * std::max(ctcn_device_cus(), 64));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(CONV_THREADS), P.lds, st, src, w, bias, out, P.m, P.rwp_log, P.region_floats, ctcn_get_option("conv_dbg"));
  return CTCN_OK;
}
int launch_gemm(const float *src, const float *w, const float *bias, float *out, const ConvPlan &P, hipStream_t st) {
  if (P.m.ntiles == 0) return CTCN_OK;
  switch (ceil_div(P.m.N, 16)) {
    case 1: return launch_gemm_nt<1>(src, w, bias, out, P, st);
    case 2: return launch_gemm_nt<2>(src, w, bias, out, P, st);
    case 3: return launch_gemm_nt<3>(src, w, bias, out, P, st);
    default: return launch_gemm_nt<4>(src, w, bias, out, P, st);
  }
}

template <int NTC, int JR>
int launch_wgrad_t(const float *x, const float *dy, float *part, const ConvPlan &P, int Co, int nch, int tpc, hipStream_t st) {
  auto kern = conv_wgrad_mfma_kernel<NTC, JR>;
  CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds));
  hipLaunchKernelGGL(kern, dim3(nch), dim3(CONV_THREADS), P.lds, st, x, dy, part, P.m, Co, P.rwp_log, P.region_floats, tpc);
  return CTCN_OK;
}
template <int NTC>
int launch_wgrad_ntc(const float *x, const float *dy, float *part, const ConvPlan &P, int Co, int nch, int tpc, hipStream_t st) {
  const int jr = ceil_div(ceil_div(P.m.K + 1, 16), 8);
  if (jr <= 1) return launch_wgrad_t<NTC, 1>(x, dy, part, P, Co, nch, tpc, st);
  if (jr <= 3) return launch_wgrad_t<NTC, 3>(x, dy, part, P, Co, nch, tpc, st);
  return launch_wgrad_t<NTC, WGRAD_JR_MAX>(x, dy, part, P, Co, nch, tpc, st);
}
int launch_wgrad(const float *x, const float *dy, float *part, const ConvPlan &P, int Co, int nch, int tpc, hipStream_t st) {
  switch (ceil_div(Co, 16)) {
    case 1: return launch_wgrad_ntc<1>(x, dy, part, P, Co, nch, tpc, st);
    case 2: return launch_wgrad_ntc<2>(x, dy, part, P, Co, nch, tpc, st);
    case 3: return launch_wgrad_ntc<3>(x, dy, part, P, Co, nch, tpc, st);
    default: return launch_wgrad_ntc<4>(x, dy, part, P, Co, nch, tpc, st);
  }
}

int wgrad_chunks(const ConvGeom &g) {
  const size_t npos = (size_t)g.B * g.Ho * g.Wo;
  return (int)std::max((size_t)1, std::min((size_t)512, ceil_div_z(npos, 64)));
}

}  // namespace

extern "C" size_t ctcn_conv2d_ws_bytes(int B, int Ci, int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
  ConvGeom g;
  if (make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw) == -1) return 0;
  return (size_t)wgrad_chunks(g) * (Co * Ci * kh * kw + Co) * sizeof(float);
}

extern "C" int ctcn_conv2d_fwd(const float *x, const float *w, const float *bias, float *y, int B, int Ci, int Hi, int Wi, int Co,
                               int kh, int kw, int sh, int sw, int ph, int pw, void *stream) {
  ConvGeom g;
  const int rc = make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw);
  CTCN_REQUIRE(rc != -1 && x && w && y, "ctcn_conv2d_fwd: bad args");
  const ConvPlan P = plan_gemm(g, 0, 0);
  if (P.ok) {
    const int lrc = launch_gemm(x, w, bias, y, P, (hipStream_t)stream);
    if (lrc) return lrc;
    CTCN_LAUNCH_CHECK();
    return CTCN_OK;
  }
  const size_t npos = (size_t)B * g.Ho * g.Wo;
  const int blocks = (int)std::min((size_t)2048, ceil_div_z(npos, 256));
  ConvSlices sl{Co, Ci, Co, Ci};
  if (!conv_slices(g, sl)) { ctcn_set_error("ctcn_conv2d_fwd: %dx%d taps per channel pair do not fit the LDS-resident kernels", kh, kw); return CTCN_EUNSUPPORTED; }
  // (one launch for a bank within 60 KB; otherwise slices of it, input channels innermost and ascending: model_ctc.py:232-233's (32, 32, (3, 21)))
  for (int co = 0; co < Co; co += sl.nco)
    for (int ci = 0; ci < Ci; ci += sl.nci) {
      const int co1 = std::min(Co, co + sl.nco), ci1 = std::min(Ci, ci + sl.nci);
      const size_t sm = (size_t)(co1 - co) * (ci1 - ci) * kh * kw * sizeof(float);
      hipLaunchKernelGGL(conv_fwd_kernel, dim3(blocks), dim3(256), sm, (hipStream_t)stream, x, w, bias, y, g, co, co1, ci, ci1);
    }
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_conv2d_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw, float *dbias, int B, int Ci,
                               int Hi, int Wi, int Co, int kh, int kw, int sh, int sw, int ph, int pw, float beta_acc, void *ws,
                               size_t ws_bytes, void *stream) {
  ConvGeom g;
  const int rc = make_geom(g, B, Ci, Hi, Wi, Co, kh, kw, sh, sw, ph, pw);
  CTCN_REQUIRE(rc != -1 && x && w && dy && dw && ws, "ctcn_conv2d_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int Wn = Co * Ci * kh * kw;
  // dgrad: one dense stride-1 product per stride class
  const int ncls = sh * sw;
  bool dgrad_mfma = dx != nullptr && ncls <= 16;
  ConvPlan DP[16];
  for (int c = 0; dgrad_mfma && c < ncls; ++c) { DP[c] = plan_gemm(g, 1, c); dgrad_mfma = DP[c].ok; }
  const ConvPlan WP = plan_wgrad(g);
  ConvSlices sl{Co, Ci, Co, Ci};
  if (!(WP.ok && (dgrad_mfma || !dx)) && !conv_slices(g, sl)) {
    ctcn_set_error("ctcn_conv2d_bwd: %dx%d taps per channel pair do not fit the LDS-resident kernels", kh, kw);
    return CTCN_EUNSUPPORTED;
  }
  if (dx && dgrad_mfma) {
    for (int c = 0; c < ncls; ++c) {
      const int lrc = launch_gemm(dy, w, nullptr, dx, DP[c], st);
      if (lrc) return lrc;
      CTCN_LAUNCH_CHECK();
    }
  } else if (dx) {
    const size_t npos = (size_t)B * Hi * Wi;
    const int blocks = (int)std::min((size_t)2048, ceil_div_z(npos, 256));
    for (int ci = 0; ci < Ci; ci += sl.nci)                       // output-channel slices innermost: the first one starts dx from zero
      for (int co = 0; co < Co; co += sl.nco) {
        const int co1 = std::min(Co, co + sl.nco), ci1 = std::min(Ci, ci + sl.nci);
        hipLaunchKernelGGL(conv_dgrad_kernel, dim3(blocks), dim3(256), (size_t)(co1 - co) * (ci1 - ci) * kh * kw * sizeof(float), st, dy, w, dx, g, co, co1, ci, ci1);
      }
    CTCN_LAUNCH_CHECK();
  }
  const int chunks = wgrad_chunks(g);
  if (ws_bytes < (size_t)chunks * (Wn + Co) * sizeof(float)) { ctcn_set_error("ctcn_conv2d_bwd: workspace too small"); return CTCN_EWORKSPACE; }
  int nch;
  if (WP.ok) {
    const int tpc = ceil_div(WP.m.ntiles, std::min(WP.m.ntiles, chunks));
    nch = ceil_div(WP.m.ntiles, tpc);
    const int lrc = launch_wgrad(x, dy, (float *)ws, WP, Co, nch, tpc, st);
    if (lrc) return lrc;
  } else {
    const size_t npos = (size_t)B * g.Ho * g.Wo;
    const int ppc = (int)ceil_div_z(npos, chunks);
    nch = (int)ceil_div_z(npos, ppc);
    for (int co = 0; co < Co; co += sl.wnco)
      for (int ci = 0; ci < Ci; ci += sl.wnci) {
        const int co1 = std::min(Co, co + sl.wnco), ci1 = std::min(Ci, ci + sl.wnci);
        const size_t sm = (size_t)WG_P * ((ci1 - ci) * kh * kw + (co1 - co)) * sizeof(float);
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(nch), dim3(256), sm, st, x, dy, (float *)ws, g, ppc, co, co1, ci, ci1);
      }
  }
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(ceil_div(Wn + Co, 64)), dim3(1024), 0, st, (const float *)ws, nch, Wn, Co, dw, dbias, beta_acc);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
