// ctc.hip -- log_softmax (+arg-max), CTC alpha/beta lattice and gradient for gfx950.
//
// replaces: nn.LogSoftmax(dim=-1) (reference timit/models/model_ctc.py:140,168,181), torch.max(out,-1)
// (timit/steps/train_ctc.py:51, timit/utils/ctcDecoder.py:163) and nn.CTCLoss(reduction='sum') forward +
// autograd backward (train_ctc.py:144,47-48,63).  Arithmetic: SURVEY Appendix A.6-A.8; blank = 0,
// zero_infinity = False (an infeasible utterance gives nll = +inf and NaN gradient rows, as torch).
//
// Kernels (all HBM/latency-bound, no MFMA):
//   log_softmax fwd : one wave per row, lanes over classes, max / sum-exp by wave shuffles; the arg-max
//                     (lowest index on ties) is taken on the produced log-probs in the same pass.
//   ctc_alpha/beta  : one workgroup per utterance; the S = 2L+1 lattice states live across the threads,
//                     alpha_{t-1} / beta_{t+1} in a double-buffered LDS row (neighbour states s-1, s-2 are
//                     LDS reads), one barrier per frame; the log-prob gather lp[t, ext(s)] of the NEXT frame
//                     is issued before the current frame's recursion so its latency is hidden.
//                     alpha is stored (T,B,S) for the backward; beta is folded into it (alpha+beta).
//   ctc_grad        : fully parallel over (t,b): one wave per frame; blank occupancy by a wave-wide
//                     log-sum-exp over the even states, label occupancy by a per-class scan of the label
//                     string (deterministic: fixed order, no atomics).
#include <algorithm>

#include "common.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int CTC_NS = 16;  // lattice states per thread -> S <= 4096 (L <= 2047; 3 S floats of LDS = 48 KB)

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

template <bool WRITE_LP>
__global__ __launch_bounds__(256) void log_softmax_kernel(const float *__restrict__ in, float *__restrict__ lp,
                                                          int32_t *__restrict__ amax, int rows, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *x = in + (size_t)row * V;
  float m = -INFINITY;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float lse = 0.0f;
  if (WRITE_LP) {
    float s = 0.0f;
    for (int c = lane; c < V; c += 64) s += expf(x[c] - m);
    s = wave_sum(s);
    lse = m + logf(s);
  }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = WRITE_LP ? x[c] - lse : x[c];
    if (WRITE_LP) lp[(size_t)row * V + c] = v;
    if (better(v, c, bv, bi)) { bv = v; bi = c; }
  }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) amax[row] = bi;
  }
}

__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float *__restrict__ lp, const float *__restrict__ dlp,
                                                              float *__restrict__ dx, int rows, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *l = lp + (size_t)row * V, *g = dlp + (size_t)row * V;
  float s = 0.0f;
  for (int c = lane; c < V; c += 64) s += g[c];
  s = wave_sum(s);
  for (int c = lane; c < V; c += 64) dx[(size_t)row * V + c] = g[c] - expf(l[c]) * s;
}

// log(e^x0 + e^x1 + e^x2) on the hardware exp2 / log2 (1 ulp each): the lattice recursion is a chain of T dependent
// steps per utterance, so the ~100 instructions of three expf + logf per step were most of its run time.  The sum is in
// [1, 3], so the log2 result carries an absolute error below 2e-7 per step.
__device__ __forceinline__ float lse3(float x0, float x1, float x2) {
  const float m = fmaxf(x0, fmaxf(x1, x2));
  if (m == -INFINITY) return -INFINITY;
  const float L2E = 1.4426950408889634f;
  const float sum = __builtin_amdgcn_exp2f((x0 - m) * L2E) + __builtin_amdgcn_exp2f((x1 - m) * L2E) + __builtin_amdgcn_exp2f((x2 - m) * L2E);
  return m + __builtin_amdgcn_logf(sum) * 0.6931471805599453f;
}

// direction: +1 alpha (t = 0..Tb-1, neighbours s-1, s-2), -1 beta (t = Tb-1..0, neighbours s+1, s+2)
// NS = lattice states per thread (host picks the smallest of 1/2/4/8 that covers S = 2L+1 with 256 threads).
// The recursion is a chain of Tb dependent steps whose per-step work is three LDS reads, one lse3 and one LDS write, so
// everything else is kept off that chain: the gathered log-probs (and, in the beta pass, the alpha values that
// alpha + beta overwrites in place) are fetched PF frames ahead into registers, the barrier between steps waits for LDS
// only, and the alpha / alpha+beta stores are never waited for.
// ADD (beta pass only): accumulate into `alpha` in place (alpha + beta, the one-buffer reserve of ctcn_ctc_fwd / _bwd);
// otherwise the pass writes its own lattice, which lets alpha and beta run side by side in one launch (ctcn_ctc_fwd_both).
template <int DIR, int NS, bool ADD>
__device__ __forceinline__ void ctc_lattice_body(float *smem, const float *__restrict__ lp, const int64_t *__restrict__ targets,
                                                 const int64_t *__restrict__ in_len, const int64_t *__restrict__ tgt_len,
                                                 float *__restrict__ alpha, float *__restrict__ nll, int T, int B, int V, int Lmax) {
  constexpr int PF = 4;
  constexpr bool ACC = DIR < 0 && ADD;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Smax = 2 * Lmax + 1;
  const int Tb = (int)in_len[b], L = (int)tgt_len[b];
  // lengths a caller must not pass (torch.nn.CTCLoss raises): nothing of the lattice is defined -- NaN loss, NaN gradient rows
  if (in_len[b] < 0 || in_len[b] > T || tgt_len[b] < 0 || tgt_len[b] > Lmax) {
    if (DIR > 0 && tid == 0) nll[b] = __uint_as_float(0x7fc00000u);
    return;
  }
  const int S = 2 * L + 1;
  float *buf0 = smem, *buf1 = smem + Smax;
  int *ext = reinterpret_cast<int *>(smem + 2 * Smax);
  for (int s = tid; s < S; s += CTC_THREADS) ext[s] = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : 0;
  __syncthreads();
  if (Tb <= 0) {
    if (DIR > 0 && tid == 0) nll[b] = L == 0 ? 0.0f : INFINITY;
    return;
  }
  // per-thread static state info
  int my_ext[NS];
  bool my_skip[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = tid + k * CTC_THREADS;
    my_ext[k] = 0; my_skip[k] = false;
    if (s < S) {
      my_ext[k] = ext[s];
      if (DIR > 0) my_skip[k] = s >= 2 && ext[s] != 0 && ext[s] != ext[s - 2];
      else my_skip[k] = s + 2 < S && ext[s] != 0 && ext[s] != ext[s + 2];
    }
  }
  const int t_first = DIR > 0 ? 0 : Tb - 1;
  // frame t_first
  {
    const float *lpt = lp + ((size_t)t_first * B + b) * V;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = tid + k * CTC_THREADS;
      if (s < S) {
        float v = -INFINITY;
        if (DIR > 0) { if (s <= 1) v = lpt[my_ext[k]]; }
        else { if (s >= S - 2) v = lpt[my_ext[k]]; }
        buf0[s] = v;
        float *ap = alpha + ((size_t)t_first * B + b) * Smax + s;
        if (ACC) *ap += v; else *ap = v;
      }
    }
  }
  __syncthreads();
  float *prev = buf0, *cur = buf1;
  float nq[PF][NS], na[PF][NS];      // prefetched frames n0 .. n0+PF-1 of the NEXT chunk: log-prob gather / old alpha
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int t = t_first + DIR * min(1 + i, Tb - 1);
    const float *lpt = lp + ((size_t)t * B + b) * V;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int s = tid + k * CTC_THREADS;
      nq[i][k] = s < S ? lpt[my_ext[k]] : 0.0f;
      na[i][k] = (ACC && s < S) ? alpha[((size_t)t * B + b) * Smax + s] : 0.0f;
    }
  }
  for (int n0 = 1; n0 < Tb; n0 += PF) {
    float cq[PF][NS], ca[PF][NS];
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      for (int k = 0; k < NS; ++k) { cq[i][k] = nq[i][k]; ca[i][k] = na[i][k]; }
    if (n0 + PF < Tb) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int t = t_first + DIR * min(n0 + PF + i, Tb - 1);
        const float *lpt = lp + ((size_t)t * B + b) * V;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int s = tid + k * CTC_THREADS;
          nq[i][k] = s < S ? lpt[my_ext[k]] : 0.0f;
          if (ACC) na[i][k] = s < S ? alpha[((size_t)t * B + b) * Smax + s] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int n = n0 + i;
      if (n < Tb) {
        const int t = t_first + DIR * n;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int s = tid + k * CTC_THREADS;
          if (s < S) {
            const float x0 = prev[s];
            float x1, x2;
            if (DIR > 0) { x1 = s >= 1 ? prev[s - 1] : -INFINITY; x2 = my_skip[k] ? prev[s - 2] : -INFINITY; }
            else { x1 = s + 1 < S ? prev[s + 1] : -INFINITY; x2 = my_skip[k] ? prev[s + 2] : -INFINITY; }
            const float v = lse3(x0, x1, x2) + cq[i][k];
            cur[s] = v;
            alpha[((size_t)t * B + b) * Smax + s] = ACC ? ca[i][k] + v : v;
          }
        }
        lds_barrier();       // LDS only: the alpha stores and the prefetches stay in flight across timesteps
        float *tmp = prev; prev = cur; cur = tmp;
      }
    }
  }
  if (DIR > 0 && tid == 0) {
    const float l1 = prev[S - 1], l2 = S > 1 ? prev[S - 2] : -INFINITY;
    const float m = fmaxf(l1, l2);
    nll[b] = m == -INFINITY ? INFINITY : -(m + logf(expf(l1 - m) + expf(l2 - m)));
  }
}

template <int DIR, int NS>
__global__ __launch_bounds__(CTC_THREADS) void ctc_lattice_kernel(const float *__restrict__ lp, const int64_t *__restrict__ targets,
                                                                  const int64_t *__restrict__ in_len,
                                                                  const int64_t *__restrict__ tgt_len, float *__restrict__ alpha,
                                                                  float *__restrict__ nll, int T, int B, int V, int Lmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  ctc_lattice_body<DIR, NS, true>(smem, lp, targets, in_len, tgt_len, alpha, nll, T, B, V, Lmax);
}

// (Measured and dropped: one wave per (utterance, pass) with the lattice in registers -- lane l holding states 2l and 2l + 1,
// no LDS, no barrier -- is bit-identical but not faster at cfg2: neighbour states through ds_bpermute 230 us (round 1); through DPP
// wave shifts (one per alpha step), with hand-counted vmcnt for the gathers / lattice stores, pointer-stepped addressing and a
// two-term lse for the blank states: 195 us, the same as the 193-195 us of this kernel (round 2).  With lse3 replaced by a max the
// wave version takes 101 us: a step is ~280 cycles of quarter-rate transcendentals (5 exp2 + 2 log2 per lane holding two states)
// + ~300 cycles of everything else, and neither is the exchange.)
// alpha (blockIdx.y = 0) and beta (blockIdx.y = 1) of every utterance in one launch: the two passes are independent chains of
// T dependent steps, so running them side by side halves the latency of the loss (2B workgroups instead of B twice).
template <int NS>
__global__ __launch_bounds__(CTC_THREADS) void ctc_lattices_kernel(const float *__restrict__ lp, const int64_t *__restrict__ targets,
                                                                   const int64_t *__restrict__ in_len,
                                                                   const int64_t *__restrict__ tgt_len, float *__restrict__ alpha,
                                                                   float *__restrict__ beta, float *__restrict__ nll, int T, int B, int V,
                                                                   int Lmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.y == 0) ctc_lattice_body<1, NS, false>(smem, lp, targets, in_len, tgt_len, alpha, nll, T, B, V, Lmax);
  else ctc_lattice_body<-1, NS, false>(smem, lp, targets, in_len, tgt_len, beta, nll, T, B, V, Lmax);
}

// online log-sum-exp accumulator
struct Lse {
  float m, s;
  __device__ __forceinline__ void add(float x) {
    if (x == -INFINITY) return;
    if (x > m) { s = s * expf(m - x) + 1.0f; m = x; }
    else s += expf(x - m);
  }
};

// SEP: alpha and beta are separate lattices (`ab` = alpha, `bt` = beta) and are added here -- the same single f32 add the
// in-place beta pass performs, so both reserves give bit-identical gradients.
// One workgroup = GRAD_TCH frames of ONE utterance (grid (ceil(T / GRAD_TCH), B), a wave per frame in turn).  The gradient of class c
// at a frame is exp(lp) - exp(logsumexp over the label positions j with target[j] == c of alpha+beta - ...): instead of every class
// lane scanning all L labels at every frame (O(V L) per frame: 81 us at cfg2), the workgroup sorts the label positions by class once
// (counting sort in LDS, positions of a class in increasing order) and a class lane walks its own positions only -- the same terms
// added in the same order, so the result is bit-identical to the scan.
constexpr int GRAD_TCH = 16;
template <bool SEP>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float *__restrict__ lp, const int64_t *__restrict__ targets,
                                                       const int64_t *__restrict__ in_len, const int64_t *__restrict__ tgt_len,
                                                       const float *__restrict__ ab, const float *__restrict__ bt,
                                                       const float *__restrict__ nll,
                                                       const float *__restrict__ gscale, float *__restrict__ grad, int T, int B, int V,
                                                       int Lmax) {
  extern __shared__ int gsm[];                 // start[V + 1] | pos[Lmax]
  int *start = gsm, *pos = gsm + V + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * GRAD_TCH, t1 = min(T, t0 + GRAD_TCH);
  const bool bad = in_len[b] < 0 || in_len[b] > T || tgt_len[b] < 0 || tgt_len[b] > Lmax;          // see ctc_lattice_body
  const int Tb = bad ? 0 : (int)in_len[b], L = bad ? 0 : (int)tgt_len[b];
  const int64_t *tg = targets + (size_t)b * Lmax;
  if (!bad && t0 < Tb) {
    // class c (thread c) counts its label positions, a serial prefix sum over the V classes, then every class writes its positions
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
      int n = 0;
      if (c > 0)
        for (int j = 0; j < L; ++j) n += (int)tg[j] == c;
      start[c + 1] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      start[0] = 0;
      for (int c = 0; c < V; ++c) start[c + 1] += start[c];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
      int k = start[c];
      if (c > 0)
        for (int j = 0; j < L; ++j)
          if ((int)tg[j] == c) pos[k++] = j;
    }
    __syncthreads();
  }
  const int Smax = 2 * Lmax + 1;
  const float gs = gscale[0];
  for (int t = t0 + wave; t < t1; t += 4) {
    const size_t pair = (size_t)t * B + b;
    float *g = grad + pair * V;
    if (bad) {
      for (int c = lane; c < V; c += 64) g[c] = __uint_as_float(0x7fc00000u);
      continue;
    }
    if (t >= Tb) {
      for (int c = lane; c < V; c += 64) g[c] = 0.0f;
      continue;
    }
    const float *abr = ab + pair * Smax;
    const float *btr = SEP ? bt + pair * Smax : nullptr;
    auto AB = [&](int s_) -> float { return SEP ? abr[s_] + btr[s_] : abr[s_]; };
    const float *lpr = lp + pair * V;
    const float n = nll[b];
    // blank: even states 0,2,..,2L
    Lse bl{-INFINITY, 0.0f};
    for (int j = lane; j <= L; j += 64) bl.add(AB(2 * j));
    const float M = wave_max(bl.m);
    float ssum = bl.m == -INFINITY ? 0.0f : bl.s * expf(bl.m - M);
    ssum = wave_sum(ssum);
    const float lcab0 = M == -INFINITY ? -INFINITY : M + logf(ssum);
    for (int c = lane; c < V; c += 64) {
      float lcab;
      if (c == 0) lcab = lcab0;
      else {
        Lse a{-INFINITY, 0.0f};
        for (int k = start[c]; k < start[c + 1]; ++k) a.add(AB(2 * pos[k] + 1));
        lcab = a.m == -INFINITY ? -INFINITY : a.m + logf(a.s);
      }
      const float l = lpr[c];
      g[c] = (expf(l) - expf(lcab + n - l)) * gs;
    }
  }
}

__global__ void greedy_collapse_kernel(const int32_t *__restrict__ idx, size_t st_t, size_t st_b, const int32_t *__restrict__ lens,
                                       int32_t *__restrict__ out_ids, int32_t *__restrict__ out_len, int T, int B, int blank) {
  // one wave per utterance: 64 frames per iteration, ballot + popcount compaction (order preserving)
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = min(max(lens[b], 0), T);
  int base = 0;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const int t = t0 + lane;
    int k = blank;
    bool keep = false;
    if (t < n) {
      k = idx[t * st_t + b * st_b];
      keep = k != blank && (t == 0 || k != idx[(t - 1) * st_t + b * st_b]);
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) out_ids[(size_t)b * T + base + __popcll(mask & ((1ull << lane) - 1ull))] = k;
    base += __popcll(mask);
  }
  if (lane == 0) out_len[b] = base;
}

// Levenshtein distance between the collapsed prediction a[b,:a_len[b]] (int32) and the label b[b,:b_len[b]] (int64):
// one lane per utterance, DP row in LDS (row stride ldrow), sequential over the O(La*Lb) cells of its utterance.
__global__ void edit_distance_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ a_len, const int64_t *__restrict__ bl,
                                     const int64_t *__restrict__ b_len, int32_t *__restrict__ out, int B, int lda, int ldb, int ldrow) {
  extern __shared__ int rows[];
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= B) return;
  int *row = rows + (size_t)threadIdx.x * ldrow;
  const int la = min(max(a_len[u], 0), lda), lb = (int)min(max(b_len[u], (int64_t)0), (int64_t)min(ldrow - 2, ldb));   // never beyond the row / the buffers
  const int32_t *pa = a + (size_t)u * lda;
  const int64_t *pb = bl + (size_t)u * ldb;
  for (int j = 0; j <= lb; ++j) row[j] = j;
  for (int i = 1; i <= la; ++i) {
    int diag = row[0];
    row[0] = i;
    const int x = pa[i - 1];
    for (int j = 1; j <= lb; ++j) {
      const int up = row[j];
      const int v = min(min(up + 1, row[j - 1] + 1), diag + (x != (int)pb[j - 1]));
      diag = up;
      row[j] = v;
    }
  }
  out[u] = row[lb];
}

// The same distance on one wavefront per utterance, along anti-diagonals: lane L owns the NC label columns L*NC+1 .. (L+1)*NC and
// at step d fills row i = d - L of them, so the only cross-lane traffic per step is one DPP shift (the right-most cell of the
// left neighbour, which that lane finished one step earlier; its value of two steps ago is the diagonal).  la + 63 steps of ~5
// dependent instructions instead of la*lb sequential LDS round trips: 1 230 -> ~25 us for 32 x (800 x 50) (tools/epoch_probe.py).
template <int NC>
__global__ __launch_bounds__(64) void edit_distance_wave_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ a_len, const int64_t *__restrict__ bl,
                                                                const int64_t *__restrict__ b_len, int32_t *__restrict__ out, int lda, int ldb, int max_b_len) {
  extern __shared__ int pred[];
  const int u = blockIdx.x, L = threadIdx.x;
  const int la = min(max(a_len[u], 0), lda), lb = (int)min(max(b_len[u], (int64_t)0), (int64_t)min(max_b_len, ldb));
  const int32_t *pa = a + (size_t)u * lda;
  const int64_t *pb = bl + (size_t)u * ldb;
  for (int i = L; i < la; i += 64) pred[i] = pa[i];
  int lab[NC], up[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = L * NC + c + 1;
    lab[c] = j <= lb ? (int)pb[j - 1] : -1;
    up[c] = j;                                          // row 0
  }
  int vlast = (L + 1) * NC, left_prev = L * NC;
  __syncthreads();
  int x = (L == 0 && la > 0) ? pred[0] : 0;             // prediction symbol of the row this lane fills at step 1
  for (int d = 1; d <= la + 63; ++d) {
    const int i = d - L;
    const bool active = i >= 1 && i <= la;
    const int xn = (i >= 0 && i < la) ? pred[i] : 0;    // next step's symbol (row i + 1), fetched off the dependent chain
    const int recv = __builtin_amdgcn_update_dpp(0, vlast, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const int left = L == 0 ? i : recv, diag = L == 0 ? i - 1 : left_prev;
    int nv[NC];
    nv[0] = min(min(up[0] + 1, left + 1), diag + (x != lab[0] ? 1 : 0));
#pragma unroll
    for (int c = 1; c < NC; ++c) nv[c] = min(min(up[c] + 1, nv[c - 1] + 1), up[c - 1] + (x != lab[c] ? 1 : 0));
    if (active) {
#pragma unroll
      for (int c = 0; c < NC; ++c) up[c] = nv[c];
      vlast = nv[NC - 1];
      left_prev = recv;
    }
    x = xn;
  }
  if (lb == 0) {
    if (L == 0) out[u] = la;
  } else if (L == (lb - 1) / NC) {
    const int cc = (lb - 1) % NC;
    int r = up[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) r = c == cc ? up[c] : r;
    out[u] = r;
  }
}

}  // namespace

extern "C" int ctcn_edit_distance(const int32_t *a, const int32_t *a_len, const int64_t *b, const int64_t *b_len, int32_t *out,
                                  int B, int lda, int ldb, int max_b_len, void *stream) {
  CTCN_REQUIRE(a && a_len && b_len && out && (b || ldb == 0) && B > 0 && max_b_len >= 0, "ctcn_edit_distance: bad args");
  if (max_b_len <= 512 && (size_t)lda * sizeof(int) <= 60 * 1024 && ctcn_get_option("edit_wave") != 0) {
    const size_t sm = (size_t)std::max(lda, 1) * sizeof(int);
    hipStream_t st = (hipStream_t)stream;
    if (max_b_len <= 64) hipLaunchKernelGGL(edit_distance_wave_kernel<1>, dim3(B), dim3(64), sm, st, a, a_len, b, b_len, out, lda, ldb, max_b_len);
    else if (max_b_len <= 128) hipLaunchKernelGGL(edit_distance_wave_kernel<2>, dim3(B), dim3(64), sm, st, a, a_len, b, b_len, out, lda, ldb, max_b_len);
    else if (max_b_len <= 256) hipLaunchKernelGGL(edit_distance_wave_kernel<4>, dim3(B), dim3(64), sm, st, a, a_len, b, b_len, out, lda, ldb, max_b_len);
    else hipLaunchKernelGGL(edit_distance_wave_kernel<8>, dim3(B), dim3(64), sm, st, a, a_len, b, b_len, out, lda, ldb, max_b_len);
    CTCN_LAUNCH_CHECK();
    return CTCN_OK;
  }
  const int ldrow = max_b_len + 2;
  int threads = 64;
  while (threads > 1 && (size_t)threads * ldrow * sizeof(int) > 48 * 1024) threads >>= 1;
  if ((size_t)threads * ldrow * sizeof(int) > 48 * 1024) { ctcn_set_error("ctcn_edit_distance: label length %d too long", max_b_len); return CTCN_EUNSUPPORTED; }
  hipLaunchKernelGGL(edit_distance_kernel, dim3(ceil_div(B, threads)), dim3(threads), (size_t)threads * ldrow * sizeof(int),
                     (hipStream_t)stream, a, a_len, b, b_len, out, B, lda, ldb, ldrow);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_log_softmax_fwd(const float *logits, float *lp, int32_t *argmax, int rows, int V, void *stream) {
  CTCN_REQUIRE(logits && lp && rows > 0 && V > 0, "ctcn_log_softmax_fwd: bad args");
  hipLaunchKernelGGL((log_softmax_kernel<true>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, lp, argmax, rows, V);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
extern "C" int ctcn_argmax(const float *lp, int32_t *argmax, int rows, int V, void *stream) {
  CTCN_REQUIRE(lp && argmax && rows > 0 && V > 0, "ctcn_argmax: bad args");
  hipLaunchKernelGGL((log_softmax_kernel<false>), dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, lp, (float *)nullptr, argmax, rows, V);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
extern "C" int ctcn_log_softmax_bwd(const float *lp, const float *dlp, float *dlogits, int rows, int V, void *stream) {
  CTCN_REQUIRE(lp && dlp && dlogits && rows > 0 && V > 0, "ctcn_log_softmax_bwd: bad args");
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, lp, dlp, dlogits, rows, V);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_ctc_fwd(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len, float *alpha,
                            float *nll, int T, int B, int V, int Lmax, void *stream) {
  CTCN_REQUIRE(lp && in_len && tgt_len && alpha && nll && (targets || Lmax == 0), "ctcn_ctc_fwd: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 0 && Lmax >= 0, "ctcn_ctc_fwd: bad dims");
  if (2 * Lmax + 1 > CTC_THREADS * CTC_NS) { ctcn_set_error("ctcn_ctc_fwd: label length %d > %d unsupported", Lmax, (CTC_THREADS * CTC_NS - 1) / 2); return CTCN_EUNSUPPORTED; }
  const size_t sm = (size_t)(3 * (2 * Lmax + 1)) * sizeof(float);
  const int ns = ceil_div(2 * Lmax + 1, CTC_THREADS);
#define CTC_LAUNCH(NS) hipLaunchKernelGGL((ctc_lattice_kernel<1, NS>), dim3(B), dim3(CTC_THREADS), sm, (hipStream_t)stream, lp, targets, in_len, tgt_len, alpha, nll, T, B, V, Lmax)
  if (ns <= 1) CTC_LAUNCH(1); else if (ns <= 2) CTC_LAUNCH(2); else if (ns <= 4) CTC_LAUNCH(4); else if (ns <= 8) CTC_LAUNCH(8); else CTC_LAUNCH(16);
#undef CTC_LAUNCH
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_ctc_bwd(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len, float *alpha,
                            const float *nll, const float *gscale, float *grad_lp, int T, int B, int V, int Lmax, void *stream) {
  CTCN_REQUIRE(lp && in_len && tgt_len && alpha && nll && gscale && grad_lp && (targets || Lmax == 0), "ctcn_ctc_bwd: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 0 && Lmax >= 0, "ctcn_ctc_bwd: bad dims");
  if (2 * Lmax + 1 > CTC_THREADS * CTC_NS) { ctcn_set_error("ctcn_ctc_bwd: label length %d unsupported", Lmax); return CTCN_EUNSUPPORTED; }
  const size_t sm = (size_t)(3 * (2 * Lmax + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const int ns = ceil_div(2 * Lmax + 1, CTC_THREADS);
#define CTC_LAUNCH(NS) hipLaunchKernelGGL((ctc_lattice_kernel<-1, NS>), dim3(B), dim3(CTC_THREADS), sm, st, lp, targets, in_len, tgt_len, alpha, (float *)nullptr, T, B, V, Lmax)
  if (ns <= 1) CTC_LAUNCH(1); else if (ns <= 2) CTC_LAUNCH(2); else if (ns <= 4) CTC_LAUNCH(4); else if (ns <= 8) CTC_LAUNCH(8); else CTC_LAUNCH(16);
#undef CTC_LAUNCH
  CTCN_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_grad_kernel<false>, dim3(ceil_div(T, GRAD_TCH), B), dim3(256), (size_t)(V + 1 + Lmax) * sizeof(int), st, lp, targets, in_len, tgt_len,
                     alpha, (const float *)nullptr, nll, gscale, grad_lp, T, B, V, Lmax);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_ctc_fwd_both(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len, float *alpha,
                                 float *beta, float *nll, int T, int B, int V, int Lmax, void *stream) {
  CTCN_REQUIRE(lp && in_len && tgt_len && alpha && beta && nll && (targets || Lmax == 0), "ctcn_ctc_fwd_both: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 0 && Lmax >= 0, "ctcn_ctc_fwd_both: bad dims");
  if (2 * Lmax + 1 > CTC_THREADS * CTC_NS) { ctcn_set_error("ctcn_ctc_fwd_both: label length %d > %d unsupported", Lmax, (CTC_THREADS * CTC_NS - 1) / 2); return CTCN_EUNSUPPORTED; }
  const size_t sm = (size_t)(3 * (2 * Lmax + 1)) * sizeof(float);
  const int ns = ceil_div(2 * Lmax + 1, CTC_THREADS);
#define CTC_LAUNCH(NS) hipLaunchKernelGGL((ctc_lattices_kernel<NS>), dim3(B, 2), dim3(CTC_THREADS), sm, (hipStream_t)stream, lp, targets, in_len, tgt_len, alpha, beta, nll, T, B, V, Lmax)
  if (ns <= 1) CTC_LAUNCH(1); else if (ns <= 2) CTC_LAUNCH(2); else if (ns <= 4) CTC_LAUNCH(4); else if (ns <= 8) CTC_LAUNCH(8); else CTC_LAUNCH(16);
#undef CTC_LAUNCH
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_ctc_grad(const float *lp, const int64_t *targets, const int64_t *in_len, const int64_t *tgt_len, const float *alpha,
                             const float *beta, const float *nll, const float *gscale, float *grad_lp, int T, int B, int V, int Lmax,
                             void *stream) {
  CTCN_REQUIRE(lp && in_len && tgt_len && alpha && beta && nll && gscale && grad_lp && (targets || Lmax == 0), "ctcn_ctc_grad: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 0 && Lmax >= 0, "ctcn_ctc_grad: bad dims");
  hipLaunchKernelGGL(ctc_grad_kernel<true>, dim3(ceil_div(T, GRAD_TCH), B), dim3(256), (size_t)(V + 1 + Lmax) * sizeof(int), (hipStream_t)stream, lp, targets,
                     in_len, tgt_len, alpha, beta, nll, gscale, grad_lp, T, B, V, Lmax);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

// (loss, sum of the edit distances, sum of the label lengths, status word) of a training step as four doubles: what
// steps/train_ctc.run_epoch copies to the host one step behind -- one launch instead of nine small torch kernels
__global__ void step_stats_kernel(const float *__restrict__ loss, const int32_t *__restrict__ dist, const int64_t *__restrict__ tgt_len, int B,
                                  const int32_t *__restrict__ status, double *__restrict__ out) {
  const int lane = threadIdx.x;
  long long d = 0, n = 0;
  for (int b = lane; b < B; b += 64) { d += dist[b]; n += tgt_len[b]; }
  for (int o = 32; o > 0; o >>= 1) { d += __shfl_down(d, o, 64); n += __shfl_down(n, o, 64); }
  if (lane == 0) {
    out[0] = (double)loss[0];
    out[1] = (double)d;
    out[2] = (double)n;
    out[3] = status ? (double)status[0] : 0.0;
  }
}
extern "C" int ctcn_step_stats(const float *loss, const int32_t *dist, const int64_t *tgt_len, int B, const int32_t *status, double *out4,
                               void *stream) {
  CTCN_REQUIRE(loss && dist && tgt_len && out4 && B > 0, "ctcn_step_stats: bad args");
  hipLaunchKernelGGL(step_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, dist, tgt_len, B, status, out4);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_greedy_collapse(const int32_t *idx, size_t stride_t, size_t stride_b, const int32_t *lens, int32_t *out_ids,
                                    int32_t *out_len, int T, int B, int blank, void *stream) {
  CTCN_REQUIRE(idx && lens && out_ids && out_len && T > 0 && B > 0, "ctcn_greedy_collapse: bad args");
  hipLaunchKernelGGL(greedy_collapse_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, idx, stride_t, stride_b, lens, out_ids, out_len, T, B, blank);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
