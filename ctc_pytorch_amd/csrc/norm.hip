// norm.hip -- BatchNorm forward (train / eval) and backward for gfx950, HBM-bound.
//
// replaces: nn.BatchNorm1d applied over all T*B rows in BatchRNN.forward (reference
// timit/models/model_ctc.py:29-32; padded frames included), the fc BatchNorm1d (:136,165-166) and
// nn.BatchNorm2d + activation(inplace) in LayerCNN.forward (:47,63-64).  Arithmetic: SURVEY Appendix A.3.
//
// x is viewed as (outer, C, inner); statistics per channel over outer*inner elements.
//   inner == 1: rows x C matrix.  Column reductions: a workgroup covers 64 adjacent channels (lanes ->
//               consecutive channels => coalesced 256-B row segments) x a chunk of rows.
//   inner  > 1: NCHW.  A workgroup covers one channel x a chunk of the outer dim; lanes run along the
//               contiguous inner (T*F) dim.
// Partial sums are float64 and written per chunk; a finalize pass adds them in a fixed order
// (deterministic, no atomics) and produces mean / rstd / running-stat updates.  The apply and dx passes
// are single streaming passes (16 B per lane when the shape allows).
#include <algorithm>

#include "common.h"

namespace {

struct Pair { double a, b; };

// value functors --------------------------------------------------------------------------------
struct StatVal {   // (x, x*x)
  const float *x;
  __device__ __forceinline__ Pair operator()(size_t idx, int) const {
    const double v = x[idx];
    return {v, v * v};
  }
  // four consecutive elements (idx % 4 == 0, 16-B aligned base): one 16-B load per operand; added in element order
  struct Quad { f32x4 x; };
  __device__ __forceinline__ Quad load4(size_t idx) const { return {*reinterpret_cast<const f32x4 *>(x + idx)}; }
  __device__ __forceinline__ void add4(const Quad &q, int, double &a, double &b) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const double v = q.x[e]; a += v; b += v * v; }
  }
  __host__ __device__ __forceinline__ bool aligned16() const { return ((uintptr_t)x & 15) == 0; }
  // (colreduce_rows4_kernel) the four elements are four neighbouring COLUMNS of one row: an accumulator pair each
  struct ColK {};
  static constexpr int ROWS_IN_FLIGHT = 8;
  __device__ __forceinline__ ColK colk(int) const { return {}; }
  __device__ __forceinline__ void cols4(const Quad &q, const ColK &, double (&a)[4], double (&b)[4]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const double v = q.x[e]; a[e] += v; b[e] += v * v; }
  }
};
// Fused dropout behind BatchNorm (+ ReLU) (round 5: LayerCNN's Conv2d -> BatchNorm2d -> ReLU -> Dropout, model_ctc.py:61-67): `dy` is the
// gradient of the DROPPED output; the keep mask is regenerated from the Philox counters of the forward pass (ctcn_dropout's: word i & 3 of
// group offset + (i >> 2)), and the ReLU mask is recomputed from x -- the forward pass stored nothing but the dropped tensor.
struct DropSpec { float p, scale; uint64_t seed, offset; const float *gamma, *beta; };     // p == 0: plain BatchNorm backward (relu mask from y)
__device__ __forceinline__ float bn_value(float x, float m, float rs, float g, float b) { return fmaf((x - m) * rs, g, b); }   // ONE rounding sequence: forward and recomputation
__device__ __forceinline__ bool drop_keep(uint32_t word, float p) { return (word >> 8) * (1.0f / 16777216.0f) >= p; }
// dx = gamma * rstd * (dy' - sum(dy') / N - xhat * sum(dy' xhat) / N), ONE rounding sequence for every dx kernel (the fused and the plain ones agree bit for bit)
__device__ __forceinline__ float bn_dx_value(float g, float xh, float ga, float rs, float s0, float s1, float inv_n) {
  return ga * rs * fmaf(-xh, s1 * inv_n, fmaf(-s0, inv_n, g));
}

struct BwdVal {    // (dy', dy' * xhat), dy' = dy masked by relu (and, with a DropSpec, first by the dropout's keep mask, scaled)
  const float *x, *y, *dy, *mean, *rstd;
  int relu;
  DropSpec d;
  __device__ __forceinline__ Pair operator()(size_t idx, int c) const {
    float g = dy[idx];
    if (d.p > 0.0f) {
      uint32_t r[4];
      philox4(d.seed, d.offset + (idx >> 2), r);
      g = drop_keep(r[idx & 3], d.p) ? g * d.scale : 0.0f;
      if (relu && !(bn_value(x[idx], mean[c], rstd[c], d.gamma[c], d.beta[c]) > 0.0f)) g = 0.0f;
    } else if (relu && !(y[idx] > 0.0f)) g = 0.0f;
    const float xh = (x[idx] - mean[c]) * rstd[c];
    return {(double)g, (double)g * (double)xh};
  }
  struct Quad { f32x4 x, y, dy; };
  __device__ __forceinline__ Quad load4(size_t idx) const {
    Quad q;
    q.x = *reinterpret_cast<const f32x4 *>(x + idx);
    q.dy = *reinterpret_cast<const f32x4 *>(dy + idx);
    if (d.p > 0.0f) {                   // dy' of the four elements; the relu test is recomputed per channel in add4
      uint32_t r[4];
      philox4(d.seed, d.offset + (idx >> 2), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) q.dy[e] = drop_keep(r[e], d.p) ? q.dy[e] * d.scale : 0.0f;
      q.y = q.x;
    } else q.y = relu ? *reinterpret_cast<const f32x4 *>(y + idx) : q.x;
    return q;
  }
  __device__ __forceinline__ void add4(const Quad &q, int c, double &a, double &b) const {
    const float m = mean[c], rs = rstd[c];
    const bool rec = d.p > 0.0f;
    const float ga = rec ? d.gamma[c] : 0.0f, be = rec ? d.beta[c] : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = q.dy[e];
      const float yv = rec ? bn_value(q.x[e], m, rs, ga, be) : q.y[e];
      if (relu && !(yv > 0.0f)) g = 0.0f;
      const float xh = (q.x[e] - m) * rs;
      a += (double)g; b += (double)g * (double)xh;
    }
  }
  __host__ __device__ __forceinline__ bool aligned16() const {
    return (((uintptr_t)x | (uintptr_t)dy | ((relu && !(d.p > 0.0f)) ? (uintptr_t)y : 0)) & 15) == 0;
  }
  // (colreduce_rows4_kernel) four neighbouring columns of one row: their per-column constants are loaded once per thread; the value of
  // every element is formed by the expressions of operator() / add4 above
  struct ColK { float m[4], rs[4], ga[4], be[4]; };
  static constexpr int ROWS_IN_FLIGHT = 4;
  __device__ __forceinline__ ColK colk(int c0) const {
    ColK k;
    const bool rec = d.p > 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { k.m[e] = mean[c0 + e]; k.rs[e] = rstd[c0 + e]; k.ga[e] = rec ? d.gamma[c0 + e] : 0.0f; k.be[e] = rec ? d.beta[c0 + e] : 0.0f; }
    return k;
  }
  __device__ __forceinline__ void cols4(const Quad &q, const ColK &k, double (&a)[4], double (&b)[4]) const {
    const bool rec = d.p > 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = q.dy[e];
      const float yv = rec ? bn_value(q.x[e], k.m[e], k.rs[e], k.ga[e], k.be[e]) : q.y[e];
      if (relu && !(yv > 0.0f)) g = 0.0f;
      const float xh = (q.x[e] - k.m[e]) * k.rs[e];
      a[e] += (double)g; b[e] += (double)g * (double)xh;
    }
  }
};
static const DropSpec k_no_drop = {0.0f, 1.0f, 0, 0, nullptr, nullptr};

template <class F>
__global__ __launch_bounds__(256) void colreduce_rows_kernel(F f, int rows, int C, int rows_per_chunk, double *__restrict__ part) {
  __shared__ double sa[4][64], sb[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  double a = 0.0, b = 0.0;
  if (c < C) {
    // eight rows' values are fetched before they are added (eight loads in flight per lane instead of what the compiler pipelines by
    // itself: 22 -> 15 us for 25 600 x 640); the additions keep their order
    int r = r0 + ry;
    for (; r + 28 < r1; r += 32) {
      Pair p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = f((size_t)(r + 4 * u) * C + c, c);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += p[u].a; b += p[u].b; }
    }
    for (; r < r1; r += 4) {
      const Pair p = f((size_t)r * C + c, c);
      a += p.a; b += p.b;
    }
  }
  sa[ry][cx] = a; sb[ry][cx] = b;
  __syncthreads();
  if (ry == 0 && c < C) {
    a = sa[0][cx] + sa[1][cx] + sa[2][cx] + sa[3][cx];
    b = sb[0][cx] + sb[1][cx] + sb[2][cx] + sb[3][cx];
    part[((size_t)blockIdx.y * C + c) * 2 + 0] = a;
    part[((size_t)blockIdx.y * C + c) * 2 + 1] = b;
  }
}

// The same sums with 16-B loads (round 5; C % 4 == 0, 16-B aligned operands): a lane takes four neighbouring columns of a row, sixteen lanes a
// 256-B run of it, the workgroup's sixteen row phases x ROWS_IN_FLIGHT rows are requested before the first is added -- 8-32 KB in flight per
// wave instead of 2.  tools/bn_rows_probe.py (whole BatchNorm calls): backward 412 -> 338 us at cfg4's 76 800 x 1 024, 136 -> 111 at cfg2's
// 25 600 x 640; forward 73 -> 63 at cfg2, unchanged at cfg4 (315 MB in 90 us: the rate a pure read reduction gets from this chip either way).
// Steps: cfg2 13.33 -> 13.25 ms, cfg4 53.2 -> 52.8 (A/B in one session, option "bn_rows4"), losses bit-identical.
// Same chunks (grid, rows_per_chunk) and the same per-element values as colreduce_rows_kernel; a column's rows are grouped into sixteen
// phases instead of four before the (fixed-order) float64 sums, so the two kernels agree to float64 rounding, not bit for bit.
template <class F>
__global__ __launch_bounds__(256) void colreduce_rows4_kernel(F f, int rows, int C, int rows_per_chunk, double *__restrict__ part) {
  __shared__ double sa[16][65], sb[16][65];
  const int l16 = threadIdx.x & 15, rp = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + 4 * l16;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  if (c0 < C && r0 < r1) {                         // (C % 4 == 0: a quad is inside or outside as a whole)
    const typename F::ColK k = f.colk(c0);
    constexpr int U = F::ROWS_IN_FLIGHT;
    for (int r = r0 + rp; r < r1; r += 16 * U) {
      typename F::Quad q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = f.load4((size_t)min(r + 16 * u, r1 - 1) * C + c0);     // (clamped address; the row test is on the addition)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r + 16 * u < r1) f.cols4(q[u], k, a, b);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { sa[rp][4 * l16 + e] = a[e]; sb[rp][4 * l16 + e] = b[e]; }
  __syncthreads();
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < C) {
    double ta = sa[0][threadIdx.x], tb = sb[0][threadIdx.x];
#pragma unroll
    for (int p = 1; p < 16; ++p) { ta += sa[p][threadIdx.x]; tb += sb[p][threadIdx.x]; }
    part[((size_t)blockIdx.y * C + c) * 2 + 0] = ta;
    part[((size_t)blockIdx.y * C + c) * 2 + 1] = tb;
  }
}

// NCHW: chunk k of a channel = plane o = k / ich (outer index), inner range [ic * ilen, (ic + 1) * ilen) with ic = k % ich.  Round 3 (the
// shipped-YAML shape, 8 x 32 x 400 x 122: one workgroup per (channel, image) walked 195 KB with one dword load in flight per lane: 100 /
// 163 us for the backward sums against a ~30 us HBM floor): the planes are cut along the inner dimension as well (`ich` pieces of <= 64 KB),
// lanes take 16-B pieces, four of them in flight per operand; the partials stay float64, one per chunk, added by the finalize pass in chunk
// order (deterministic).
template <class F>
__global__ __launch_bounds__(256) void colreduce_nchw_kernel(F f, int outer, int C, int inner, int opc, int ich, int ilen, int vec, double *__restrict__ part) {
  __shared__ double sa[4], sb[4];
  const int c = blockIdx.x;
  // chunk -> planes [o0, o1) (opc of them; 1 when the planes are cut into ich pieces) and the inner range [i0, i1)
  const int oc = blockIdx.y / ich, ic = blockIdx.y - oc * ich;
  const int o0 = oc * opc, o1 = min(outer, o0 + opc);
  const int i0 = ic * ilen, i1 = min(inner, i0 + ilen);
  double a = 0.0, b = 0.0;
  for (int o = o0; o < o1; ++o) {
    const size_t base = ((size_t)o * C + c) * inner;
    if (vec) {                      // inner % 4 == 0, ilen % 1024 == 0, 16-B aligned operands
      int i = i0 + threadIdx.x * 4;
      for (; i + 3 * 1024 < i1; i += 4096) {
        typename F::Quad q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = f.load4(base + i + 1024 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) f.add4(q[u], c, a, b);
      }
      for (; i < i1; i += 1024) f.add4(f.load4(base + i), c, a, b);
    } else {
      for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const Pair p = f(base + i, c);
        a += p.a; b += p.b;
      }
    }
  }
  a = wave_sum_d(a); b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)blockIdx.y * C + c) * 2 + 0] = sa[0] + sa[1] + sa[2] + sa[3];
    part[((size_t)blockIdx.y * C + c) * 2 + 1] = sb[0] + sb[1] + sb[2] + sb[3];
  }
}

__global__ void bn_finalize_stats_kernel(const double *__restrict__ part, int nchunks, int C, double count, float eps,
                                         float momentum, float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                         float *__restrict__ rm, float *__restrict__ rv, long long *__restrict__ batches) {
  // one wave per channel: lanes take the chunk partials k = lane, lane+64, ... in order, then a fixed shuffle tree
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (batches && blockIdx.x == 0 && threadIdx.x == 0) *batches += 1;          // nn.BatchNorm's num_batches_tracked: no launch of its own
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < nchunks; k += 64) { s += part[((size_t)k * C + c) * 2]; ss += part[((size_t)k * C + c) * 2 + 1]; }
  s = wave_sum_d(s); ss = wave_sum_d(ss);
  if (lane != 0) return;
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  mean_out[c] = (float)mean;
  rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * mean);
  if (rv) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * unbiased);
  }
}

// sums[c*2 + {0,1}] = sum over the chunk partials (wave per channel, fixed order): the per-rank half of a synchronised BN
__global__ void bn_sum_chunks_kernel(const double *__restrict__ part, int nchunks, int C, double *__restrict__ sums) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < nchunks; k += 64) { s += part[((size_t)k * C + c) * 2]; ss += part[((size_t)k * C + c) * 2 + 1]; }
  s = wave_sum_d(s); ss = wave_sum_d(ss);
  if (lane == 0) { sums[c * 2] = s; sums[c * 2 + 1] = ss; }
}

__global__ void bn_finalize_bwd_kernel(const double *__restrict__ part, int nchunks, int C, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, float *__restrict__ sums /*[2*C]: sum dy, sum dy*xhat*/,
                                       float beta_acc) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < nchunks; k += 64) { s += part[((size_t)k * C + c) * 2]; ss += part[((size_t)k * C + c) * 2 + 1]; }
  s = wave_sum_d(s); ss = wave_sum_d(ss);
  if (lane != 0) return;
  sums[c] = (float)s; sums[C + c] = (float)ss;
  if (dbeta) dbeta[c] = (float)s + (beta_acc != 0.0f ? beta_acc * dbeta[c] : 0.0f);
  if (dgamma) dgamma[c] = (float)ss + (beta_acc != 0.0f ? beta_acc * dgamma[c] : 0.0f);
}

// y = (x - mean)*rstd*gamma + beta  [relu]
__global__ void bn_apply_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ gamma,
                                const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ rstd_or_var,
                                float eps, int var_is_variance, size_t total, int C, int inner, int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / inner) % C);
    const float rs = var_is_variance ? 1.0f / sqrtf(rstd_or_var[c] + eps) : rstd_or_var[c];
    float v = bn_value(x[i], mean[c], rs, gamma[c], beta[c]);
    if (relu) v = fmaxf(v, 0.0f);
    y[i] = v;
  }
}

// rows of C channels (inner == 1), C % 4 == 0, 16-B aligned tensors: four channels per lane, one 32-bit modulo per four elements (the
// general kernel pays a 64-bit division and modulo per element: 29.9 -> ~22 us for 25 600 x 640, the time of the dropout pass);
// the same expression per element
__global__ void bn_apply_rows4_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ rstd_or_var,
                                      float eps, int var_is_variance, size_t total4, int C4, int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (unsigned)C4) * 4;
    const float4 xv = reinterpret_cast<const float4 *>(x)[i];
    const float4 m = *reinterpret_cast<const float4 *>(mean + c), r = *reinterpret_cast<const float4 *>(rstd_or_var + c);
    const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ms[4] = {m.x, m.y, m.z, m.w}, rv[4] = {r.x, r.y, r.z, r.w}, gv[4] = {g.x, g.y, g.z, g.w},
                bv[4] = {b.x, b.y, b.z, b.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float rs = var_is_variance ? 1.0f / sqrtf(rv[k] + eps) : rv[k];
      float v = bn_value(xs[k], ms[k], rs, gv[k], bv[k]);
      if (relu) v = fmaxf(v, 0.0f);
      o[k] = v;
    }
    reinterpret_cast<float4 *>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
// BatchNorm (+ ReLU) + inverted dropout in one pass: y = keep ? relu(bn(x)) * scale : 0 -- the values BatchNorm -> ctcn_dropout produce, bit for
// bit (same expression, same Philox words).  Four consecutive elements per thread (one Philox group); the channel per element (inner % 4 may be
// anything).
__global__ void bn_apply_drop_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ beta,
                                     const float *__restrict__ mean, const float *__restrict__ rstd, size_t total, int C, int inner, int relu, DropSpec d) {
  const size_t ngroups = (total + 3) / 4;
  const bool vec = ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
  for (size_t gi = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (size_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(d.seed, d.offset + gi, r);
    const size_t i0 = gi * 4;
    float xv[4], o[4];
    const bool full = i0 + 3 < total && vec;
    if (full) { const float4 v = *reinterpret_cast<const float4 *>(x + i0); xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; }
    else for (int e = 0; e < 4; ++e) xv[e] = i0 + e < total ? x[i0 + e] : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (int)(((i0 + e) / inner) % C);
      float v = bn_value(xv[e], mean[c], rstd[c], gamma[c], beta[c]);
      if (relu) v = fmaxf(v, 0.0f);
      o[e] = drop_keep(r[e], d.p) ? v * d.scale : 0.0f;
    }
    if (full) *reinterpret_cast<float4 *>(y + i0) = make_float4(o[0], o[1], o[2], o[3]);
    else for (int e = 0; e < 4 && i0 + e < total; ++e) y[i0 + e] = o[e];
  }
}
// dx of the same fused block: dy is the gradient of the dropped output
__global__ void bn_dx_drop_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gamma, const float *__restrict__ beta,
                                  const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ sums, float *__restrict__ dx,
                                  size_t total, int C, int inner, float inv_n, int relu, DropSpec d) {
  const size_t ngroups = (total + 3) / 4;
  const bool vec = ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0);
  for (size_t gi = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (size_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(d.seed, d.offset + gi, r);
    const size_t i0 = gi * 4;
    float xv[4], gv[4], o[4];
    const bool full = i0 + 3 < total && vec;
    if (full) {
      const float4 v = *reinterpret_cast<const float4 *>(x + i0), g4 = *reinterpret_cast<const float4 *>(dy + i0);
      xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    } else for (int e = 0; e < 4; ++e) { xv[e] = i0 + e < total ? x[i0 + e] : 0.0f; gv[e] = i0 + e < total ? dy[i0 + e] : 0.0f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = (int)(((i0 + e) / inner) % C);
      const float m = mean[c], rs = rstd[c], ga = gamma[c];
      float g = drop_keep(r[e], d.p) ? gv[e] * d.scale : 0.0f;
      if (relu && !(bn_value(xv[e], m, rs, ga, beta[c]) > 0.0f)) g = 0.0f;
      const float xh = (xv[e] - m) * rs;
      o[e] = bn_dx_value(g, xh, ga, rs, sums[c], sums[C + c], inv_n);
    }
    if (full) *reinterpret_cast<float4 *>(dx + i0) = make_float4(o[0], o[1], o[2], o[3]);
    else for (int e = 0; e < 4 && i0 + e < total; ++e) dx[i0 + e] = o[e];
  }
}

static void launch_bn_apply(hipStream_t st, const float *x, float *y, const float *gamma, const float *beta, const float *mean, const float *rstd_or_var,
                            float eps, int var_is_variance, size_t total, int C, int inner, int relu) {
  const bool al = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean | (uintptr_t)rstd_or_var) & 15) == 0);
  if (inner == 1 && C % 4 == 0 && al) {
    const size_t total4 = total / 4;
    const int blocks = (int)std::min((size_t)4096, ceil_div_z(total4, 256));
    hipLaunchKernelGGL(bn_apply_rows4_kernel, dim3(blocks), dim3(256), 0, st, x, y, gamma, beta, mean, rstd_or_var, eps, var_is_variance, total4, C / 4, relu);
  } else {
    const int blocks = (int)std::min((size_t)4096, ceil_div_z(total, 256));
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, x, y, gamma, beta, mean, rstd_or_var, eps, var_is_variance, total, C, inner, relu);
  }
}

// dx = gamma*rstd*(dy' - sum(dy')/N - xhat*sum(dy' xhat)/N)
__global__ void bn_dx_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                             const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
                             const float *__restrict__ sums, float *__restrict__ dx, size_t total, int C, int inner, float inv_n,
                             int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / inner) % C);
    float g = dy[i];
    if (relu && !(y[i] > 0.0f)) g = 0.0f;
    const float xh = (x[i] - mean[c]) * rstd[c];
    dx[i] = bn_dx_value(g, xh, gamma[c], rstd[c], sums[c], sums[C + c], inv_n);
  }
}

// the rows-of-channels form of bn_dx_kernel (see bn_apply_rows4_kernel): four channels per lane, the same expression per element
__global__ void bn_dx_rows4_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                                   const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
                                   const float *__restrict__ sums, float *__restrict__ dx, size_t total4, int C4, float inv_n, int relu) {
  const int C = C4 * 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (unsigned)C4) * 4;
    const float4 xv = reinterpret_cast<const float4 *>(x)[i], gv4 = reinterpret_cast<const float4 *>(dy)[i];
    float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (relu) yv = reinterpret_cast<const float4 *>(y)[i];
    const float4 m = *reinterpret_cast<const float4 *>(mean + c), r = *reinterpret_cast<const float4 *>(rstd + c);
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), s0 = *reinterpret_cast<const float4 *>(sums + c),
                 s1 = *reinterpret_cast<const float4 *>(sums + C + c);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
    const float ms[4] = {m.x, m.y, m.z, m.w}, rs[4] = {r.x, r.y, r.z, r.w}, gm[4] = {ga.x, ga.y, ga.z, ga.w};
    const float a0[4] = {s0.x, s0.y, s0.z, s0.w}, a1[4] = {s1.x, s1.y, s1.z, s1.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float g = gs[k];
      if (relu && !(ys[k] > 0.0f)) g = 0.0f;
      const float xh = (xs[k] - ms[k]) * rs[k];
      o[k] = bn_dx_value(g, xh, gm[k], rs[k], a0[k], a1[k], inv_n);
    }
    reinterpret_cast<float4 *>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
static void launch_bn_dx(hipStream_t st, const float *x, const float *y, const float *dy, const float *gamma, const float *mean, const float *rstd,
                         const float *sums, float *dx, size_t total, int C, int inner, float inv_n, int relu) {
  const bool al = ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)sums |
                     (relu ? (uintptr_t)y : 0)) & 15) == 0);
  if (inner == 1 && C % 4 == 0 && al) {
    const size_t total4 = total / 4;
    hipLaunchKernelGGL(bn_dx_rows4_kernel, dim3((int)std::min((size_t)4096, ceil_div_z(total4, 256))), dim3(256), 0, st, x, y, dy, gamma, mean, rstd, sums, dx,
                       total4, C / 4, inv_n, relu);
  } else {
    hipLaunchKernelGGL(bn_dx_kernel, dim3((int)std::min((size_t)4096, ceil_div_z(total, 256))), dim3(256), 0, st, x, y, dy, gamma, mean, rstd, sums, dx, total, C,
                       inner, inv_n, relu);
  }
}

int chunks_rows(int rows, int C) {
  const int colblocks = ceil_div(C, 64);
  int n = std::max(1, 1024 / colblocks);
  n = std::min(n, ceil_div(rows, 64));
  return std::max(n, 1);
}
// NCHW chunking: at most max(1, 4096 / C) chunks per channel.  Few planes (a CNN feature map: outer = batch): every plane is cut into `ich`
// pieces of <= 64 KB (whole 4-KB wave rounds); many planes: `opc` whole planes per chunk.
struct NchwChunks { int opc, ich, ilen, n; };
NchwChunks chunks_nchw(int outer, int C, int inner) {
  const int nmax = std::max(1, 4096 / std::max(C, 1));
  NchwChunks k;
  if (outer >= nmax) {
    k.opc = ceil_div(outer, nmax); k.ich = 1; k.ilen = ceil_div(std::max(inner, 1), 1024) * 1024;
    k.n = ceil_div(outer, k.opc);
  } else {
    int ich = std::min(ceil_div(inner, 16384), nmax / outer);
    ich = std::max(ich, 1);
    k.ilen = ceil_div(ceil_div(inner, ich), 1024) * 1024;
    k.ich = ceil_div(inner, k.ilen); k.opc = 1;
    k.n = outer * k.ich;
  }
  return k;
}

template <class F>
int launch_reduce(F f, int outer, int C, int inner, double *part, int *nchunks_out, hipStream_t st) {
  if (inner == 1) {
    const int n = chunks_rows(outer, C);
    const int rpc = ceil_div(outer, n);
    const int nn = ceil_div(outer, rpc);
    if (C % 4 == 0 && f.aligned16() && ctcn_get_option("bn_rows4") != 0)
      hipLaunchKernelGGL((colreduce_rows4_kernel<F>), dim3(ceil_div(C, 64), nn), dim3(256), 0, st, f, outer, C, rpc, part);
    else
      hipLaunchKernelGGL((colreduce_rows_kernel<F>), dim3(ceil_div(C, 64), nn), dim3(256), 0, st, f, outer, C, rpc, part);
    *nchunks_out = nn;
  } else {
    const NchwChunks k = chunks_nchw(outer, C, inner);
    const int nn = k.n;
    const int vec = inner % 4 == 0 && f.aligned16() ? 1 : 0;
    hipLaunchKernelGGL((colreduce_nchw_kernel<F>), dim3(C, nn), dim3(256), 0, st, f, outer, C, inner, k.opc, k.ich, k.ilen, vec, part);
    *nchunks_out = nn;
  }
  return 0;
}

}  // namespace

extern "C" size_t ctcn_bn_ws_bytes(int outer, int C, int inner) {
  if (outer <= 0 || C <= 0 || inner <= 0) return 0;                  // (as the other *_ws_bytes queries answer bad dims; the chunk planners divide by them)
  const int n = inner == 1 ? chunks_rows(outer, C) : chunks_nchw(outer, C, inner).n;
  return align_up((size_t)(n + 1) * C * 2 * sizeof(double), 256) + (size_t)2 * C * sizeof(float);
}

extern "C" int ctcn_bn_fwd_train(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                                 float *running_var, float *save_mean, float *save_rstd, int outer, int C, int inner,
                                 float eps, float momentum, int relu, void *ws, size_t ws_bytes, void *stream, long long *num_batches_tracked) {
  CTCN_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && ws, "ctcn_bn_fwd_train: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_fwd_train: bad dims");
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_fwd_train: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)ws;
  int nchunks = 0;
  launch_reduce(StatVal{x}, outer, C, inner, part, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  const double count = (double)outer * inner;
  hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nchunks, C, count, eps, momentum,
                     save_mean, save_rstd, running_mean, running_var, num_batches_tracked);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  launch_bn_apply(st, x, y, gamma, beta, save_mean, save_rstd, eps, 0, total, C, inner, relu);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

// BatchNorm (training statistics) + ReLU + dropout in one apply pass (LayerCNN: conv -> BN -> ReLU -> Dropout).  y_drop receives
// ctcn_dropout(relu(bn(x)), p, seed, offset) bit for bit; the un-dropped activation is not stored (ctcn_bn_bwd_dropout recomputes what it needs).
extern "C" int ctcn_bn_fwd_train_dropout(const float *x, float *y_drop, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                         float *save_mean, float *save_rstd, int outer, int C, int inner, float eps, float momentum, int relu, void *ws,
                                         size_t ws_bytes, void *stream, long long *num_batches_tracked, float p, uint64_t seed, uint64_t offset) {
  CTCN_REQUIRE(x && y_drop && gamma && beta && save_mean && save_rstd && ws, "ctcn_bn_fwd_train_dropout: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_fwd_train_dropout: bad dims");
  CTCN_REQUIRE(p > 0.0f && p < 1.0f, "ctcn_bn_fwd_train_dropout: p=%f outside (0,1)", (double)p);
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_fwd_train_dropout: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)ws;
  int nchunks = 0;
  launch_reduce(StatVal{x}, outer, C, inner, part, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nchunks, C, (double)outer * inner, eps, momentum,
                     save_mean, save_rstd, running_mean, running_var, num_batches_tracked);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  const DropSpec d = {p, 1.0f / (1.0f - p), seed, offset, gamma, beta};
  hipLaunchKernelGGL(bn_apply_drop_kernel, dim3((int)std::min((size_t)4096, ceil_div_z((total + 3) / 4, 256))), dim3(256), 0, st, x, y_drop, gamma, beta,
                     save_mean, save_rstd, total, C, inner, relu, d);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_bn_bwd_dropout(const float *x, const float *dy_drop, const float *gamma, const float *beta, const float *save_mean,
                                   const float *save_rstd, float *dx, float *dgamma, float *dbeta, int outer, int C, int inner, int relu, float beta_acc,
                                   void *ws, size_t ws_bytes, void *stream, float p, uint64_t seed, uint64_t offset) {
  CTCN_REQUIRE(x && dy_drop && gamma && beta && save_mean && save_rstd && dx && ws, "ctcn_bn_bwd_dropout: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_bwd_dropout: bad dims");
  CTCN_REQUIRE(p > 0.0f && p < 1.0f, "ctcn_bn_bwd_dropout: p=%f outside (0,1)", (double)p);
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_bwd_dropout: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)ws;
  const int nmax = inner == 1 ? chunks_rows(outer, C) : chunks_nchw(outer, C, inner).n;
  float *sums = (float *)((char *)ws + align_up((size_t)(nmax + 1) * C * 2 * sizeof(double), 256));
  const DropSpec d = {p, 1.0f / (1.0f - p), seed, offset, gamma, beta};
  int nchunks = 0;
  launch_reduce(BwdVal{x, nullptr, dy_drop, save_mean, save_rstd, relu, d}, outer, C, inner, part, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nchunks, C, dgamma, dbeta, sums, beta_acc);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  hipLaunchKernelGGL(bn_dx_drop_kernel, dim3((int)std::min((size_t)4096, ceil_div_z((total + 3) / 4, 256))), dim3(256), 0, st, x, dy_drop, gamma, beta,
                     save_mean, save_rstd, sums, dx, total, C, inner, (float)(1.0 / ((double)outer * inner)), relu, d);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

// ---- synchronised BatchNorm (data parallel): statistics over the GLOBAL batch --------------------------------------
// Each rank computes its per-channel sums (fp64, [C][2]), the host all-reduces them (2*C doubles), and the finish call
// normalises with the global count.  With one rank the result equals ctcn_bn_fwd_train / ctcn_bn_bwd exactly.
extern "C" int ctcn_bn_fwd_sums(const float *x, double *sums, int outer, int C, int inner, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && sums && ws, "ctcn_bn_fwd_sums: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_fwd_sums: bad dims");
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_fwd_sums: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  int nchunks = 0;
  launch_reduce(StatVal{x}, outer, C, inner, (double *)ws, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double *)ws, nchunks, C, sums);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_bn_fwd_finish(const float *x, float *y, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                  float *save_mean, float *save_rstd, const double *sums, double count_total, int outer, int C, int inner,
                                  float eps, float momentum, int relu, void *stream, long long *num_batches_tracked) {
  CTCN_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && sums, "ctcn_bn_fwd_finish: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0 && count_total >= (double)outer * inner, "ctcn_bn_fwd_finish: bad dims / count");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, sums, 1, C, count_total, eps, momentum, save_mean,
                     save_rstd, running_mean, running_var, num_batches_tracked);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  launch_bn_apply(st, x, y, gamma, beta, save_mean, save_rstd, eps, 0, total, C, inner, relu);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_bn_bwd_sums(const float *x, const float *y, const float *dy, const float *save_mean, const float *save_rstd, double *sums,
                                int outer, int C, int inner, int relu, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && dy && save_mean && save_rstd && sums && ws, "ctcn_bn_bwd_sums: null pointer");
  CTCN_REQUIRE(!relu || y, "ctcn_bn_bwd_sums: y required for the fused relu mask");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_bwd_sums: bad dims");
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_bwd_sums: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  int nchunks = 0;
  launch_reduce(BwdVal{x, y, dy, save_mean, save_rstd, relu, k_no_drop}, outer, C, inner, (double *)ws, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double *)ws, nchunks, C, sums);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

// local_sums feed dgamma / dbeta (the gradient all-reduce adds the ranks up later); global_sums and count_total feed dx
extern "C" int ctcn_bn_bwd_finish(const float *x, const float *y, const float *dy, const float *gamma, const float *save_mean,
                                  const float *save_rstd, float *dx, float *dgamma, float *dbeta, const double *local_sums,
                                  const double *global_sums, double count_total, int outer, int C, int inner, int relu, float beta_acc,
                                  void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && dy && gamma && save_mean && save_rstd && dx && local_sums && global_sums && ws, "ctcn_bn_bwd_finish: null pointer");
  CTCN_REQUIRE(!relu || y, "ctcn_bn_bwd_finish: y required for the fused relu mask");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0 && count_total >= (double)outer * inner, "ctcn_bn_bwd_finish: bad dims / count");
  if (ws_bytes < (size_t)4 * C * sizeof(float)) { ctcn_set_error("ctcn_bn_bwd_finish: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float *scratch = (float *)ws, *sums = scratch + 2 * C;
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, local_sums, 1, C, dgamma, dbeta, scratch, beta_acc);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, global_sums, 1, C, (float *)nullptr, (float *)nullptr, sums, 0.0f);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  launch_bn_dx(st, x, y, dy, gamma, save_mean, save_rstd, sums, dx, total, C, inner, (float)(1.0 / count_total), relu);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_bn_fwd_eval(const float *x, float *y, const float *gamma, const float *beta, const float *running_mean,
                                const float *running_var, int outer, int C, int inner, float eps, int relu, void *stream) {
  CTCN_REQUIRE(x && y && gamma && beta && running_mean && running_var, "ctcn_bn_fwd_eval: null pointer");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_fwd_eval: bad dims");
  const size_t total = (size_t)outer * C * inner;
  launch_bn_apply((hipStream_t)stream, x, y, gamma, beta, running_mean, running_var, eps, 1, total, C, inner, relu);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_bn_bwd(const float *x, const float *y, const float *dy, const float *gamma, const float *save_mean,
                           const float *save_rstd, float *dx, float *dgamma, float *dbeta, int outer, int C, int inner,
                           int relu, float beta_acc, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && dy && gamma && save_mean && save_rstd && dx && ws, "ctcn_bn_bwd: null pointer");
  CTCN_REQUIRE(!relu || y, "ctcn_bn_bwd: y required for the fused relu mask");
  CTCN_REQUIRE(outer > 0 && C > 0 && inner > 0, "ctcn_bn_bwd: bad dims");
  if (ws_bytes < ctcn_bn_ws_bytes(outer, C, inner)) { ctcn_set_error("ctcn_bn_bwd: workspace too small"); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)ws;
  const int nmax = inner == 1 ? chunks_rows(outer, C) : chunks_nchw(outer, C, inner).n;
  float *sums = (float *)((char *)ws + align_up((size_t)(nmax + 1) * C * 2 * sizeof(double), 256));
  int nchunks = 0;
  launch_reduce(BwdVal{x, y, dy, save_mean, save_rstd, relu, k_no_drop}, outer, C, inner, part, &nchunks, st);
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nchunks, C, dgamma, dbeta, sums, beta_acc);
  CTCN_LAUNCH_CHECK();
  const size_t total = (size_t)outer * C * inner;
  launch_bn_dx(st, x, y, dy, gamma, save_mean, save_rstd, sums, dx, total, C, inner, (float)(1.0 / ((double)outer * inner)), relu);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
