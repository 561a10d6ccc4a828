// decode.hip -- CTC prefix beam search with a bigram LM, one workgroup per utterance (gfx950).
//
// replaces: BeamDecoder.decode -> ctcBeamSearch.decode (reference timit/utils/ctcDecoder.py:181-192,
// timit/utils/BeamSearch.py:73-153 with log_add_prob :43-50, calcExtPr :52-66, BeamState.sort :29-33,
// BeamState.norm :23-27) and LanguageModel.get_bi_prob (timit/utils/NgramLM.py:65-78, pre-tabulated on the
// host into lm[(V+1)*(V+1)]).  Semantics reproduced exactly:
//   * scores are IEEE double in the ln domain with the LOG_ZERO = -99999999.0 sentinel rules of log_add_prob;
//   * the frame-skip test (1 - p_blank < 0.1) and the repeat rule (p_blank[t-1] < 0.9) are float32 compares;
//   * the python dict of labellings is modelled by a prefix trie (labelling == node id, children found
//     through an open-addressing table in the workspace), so equal labellings reached along different
//     paths merge exactly as dict keys do; the merged entry takes the insertion position of its first touch
//     and accumulates its contributions in the reference's visiting order;
//   * BHat = first W entries of a stable descending sort == W rounds of arg-max with (score desc,
//     insertion index asc) ordering over the <= W*V candidate entries of the step;
//   * the reference's two failure modes are reported, not hidden: status 1 = an empty labelling reaches the
//     final LM step (python IndexError at BeamSearch.py:135), status 2 = log of a zero probability
//     (python ValueError).
// Parallelism: utterances across workgroups (replicas, no collective), candidates (beam x class) across the
// 256 lanes of the workgroup; every per-step quantity lives in LDS (beam state, log-probs, candidate scores).
#include <algorithm>

#include "common.h"

namespace {

constexpr int BEAM_WMAX = 256;
constexpr double LOG_ZERO = -99999999.0;
constexpr unsigned long long HT_EMPTY = ~0ull;

__device__ __forceinline__ double log_add_prob(double log_x, double log_y) {   // BeamSearch.py:43-50
  if (log_x <= LOG_ZERO) return log_y;
  if (log_y <= LOG_ZERO) return log_x;
  if ((log_y - log_x) > 0.0) { const double t = log_x; log_x = log_y; log_y = t; }
  return log_x + log(1 + exp(log_y - log_x));
}

struct Fields { double nb, b, t; };
__device__ __forceinline__ void apply_stay(Fields &e, double s_nb, double s_b) {   // BeamSearch.py:108-113
  e.nb = log_add_prob(e.nb, s_nb);
  e.b = log_add_prob(e.b, s_b);
  const double tot = log_add_prob(s_b, s_nb);
  e.t = log_add_prob(e.t, tot);
}
__device__ __forceinline__ void apply_ext(Fields &e, double pr) {                  // BeamSearch.py:122-125
  e.nb = log_add_prob(e.nb, pr);
  e.t = log_add_prob(e.t, pr);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

struct BeamState {   // one copy of the beam (BHat) in LDS
  int node[BEAM_WMAX], len[BEAM_WMAX], last[BEAM_WMAX], par[BEAM_WMAX];
  double pB[BEAM_WMAX], pNB[BEAM_WMAX], pT[BEAM_WMAX];
};

struct BeamArgs {
  const float *x; int input_is_prob; const int32_t *lens; const double *lm; double alpha; int W, blank;
  int32_t *out_ids, *out_len; double *out_score; int32_t *status; int T, B, V;
  unsigned long long *ht_keys; int *ht_ids; int *node_par; int *node_sym; double *cand_global;
  int ht_size, max_nodes, cand_in_lds;
};

__device__ __forceinline__ bool cand_better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void beam_kernel(BeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];   // lg[V] | cand[W*V] (if it fits)
  __shared__ BeamState S[2];
  __shared__ double sNB[BEAM_WMAX], sB[BEAM_WMAX], sT[BEAM_WMAX];
  __shared__ int mfrom[BEAM_WMAX], sel[BEAM_WMAX];
  __shared__ double selv[BEAM_WMAX];
  __shared__ double red_v[4];
  __shared__ int red_i[4];
  __shared__ int s_flag, s_nodes, s_best;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, V = a.V, W = a.W, B = a.B, T = a.T, blank = a.blank;
  double *lg = dsm;
  double *cand = a.cand_in_lds ? dsm + V : a.cand_global + (size_t)b * W * V;
  unsigned long long *keys = a.ht_keys + (size_t)b * a.ht_size;
  int *ids = a.ht_ids + (size_t)b * a.ht_size;
  int *npar = a.node_par + (size_t)b * a.max_nodes;
  int *nsym = a.node_sym + (size_t)b * a.max_nodes;
  const int htmask = a.ht_size - 1;

  int cur = 0, nb = 1, status = 0;
  if (tid == 0) {
    S[0].node[0] = 0; S[0].len[0] = 0; S[0].last[0] = -1; S[0].par[0] = -1;
    S[0].pB[0] = 0.0; S[0].pNB[0] = LOG_ZERO; S[0].pT[0] = 0.0;   // BeamSearch.py:83-87
    s_nodes = 1; s_flag = 0;
    npar[0] = -1; nsym[0] = -1;
  }
  __syncthreads();
  const int nframes = min(max(a.lens[b], 0), T);
  for (int t = 0; t < nframes; ++t) {
    const float *row = a.x + ((size_t)t * B + b) * V;
    const float pblank = a.input_is_prob ? row[blank] : expf(row[blank]);
    if ((1.0f - pblank) < 0.1f) continue;                         // BeamSearch.py:93-94 (float32 compare)
    BeamState &L = S[cur];
    BeamState &N = S[cur ^ 1];
    // 1. ln of the frame's probabilities (math.log of the float32 value widened to double)
    for (int k = tid; k < V; k += 256) {
      const float p = a.input_is_prob ? row[k] : expf(row[k]);
      if (!(p > 0.0f)) s_flag = 2;
      lg[k] = log((double)p);
    }
    bool rep_ok = false;
    if (t > 0) {
      const float *prow = a.x + ((size_t)(t - 1) * B + b) * V;
      const float pprev = a.input_is_prob ? prow[blank] : expf(prow[blank]);
      rep_ok = pprev < 0.9f;                                       // BeamSearch.py:63 (float32 compare)
    }
    // 2. which beam (if any) is the parent labelling of beam i' -> its extension by last(i') merges with i'
    if (tid < nb) {
      int m = -1;
      if (L.len[tid] > 0) {
        const int pnode = L.par[tid];
        for (int i2 = 0; i2 < nb; ++i2) if (L.node[i2] == pnode) m = i2;
      }
      mfrom[tid] = m;
    }
    __syncthreads();
    if (s_flag == 2) { status = 2; break; }
    // 3a. extension scores (calcExtPr), candidate slot c = i*V + 1 + kk  (kk enumerates k != blank in order)
    const int ncand = nb * V;
    for (int c = tid; c < ncand; c += 256) {
      const int i = c / V, kk = c - i * V;
      if (kk == 0) continue;
      const int k = (kk - 1 < blank) ? kk - 1 : kk;
      const int c1 = L.len[i] > 0 ? L.last[i] : V;
      const double bigram = a.lm[(size_t)c1 * (V + 1) + k] * a.alpha;
      const double base = (L.len[i] > 0 && L.last[i] == k && rep_ok) ? L.pB[i] : L.pT[i];
      cand[c] = lg[k] + bigram + base;
    }
    __syncthreads();
    // 3b. stay entries, merged with the matching extension in the reference's visiting order
    if (tid < nb) {
      const int ip = tid;
      double s_nb = LOG_ZERO;
      if (L.len[ip] > 0) s_nb = L.pNB[ip] + lg[L.last[ip]];         // BeamSearch.py:102-103
      const double s_b = L.pT[ip] + lg[blank];                      // :106
      Fields e{LOG_ZERO, LOG_ZERO, LOG_ZERO};
      const int i = mfrom[ip];
      if (i >= 0) {
        const int k = L.last[ip];
        const int kk = (k < blank) ? k + 1 : k;
        const int ce = i * V + kk;
        const double pr = cand[ce];
        if (i < ip) { apply_ext(e, pr); apply_stay(e, s_nb, s_b); cand[ce] = e.t; cand[ip * V] = -INFINITY; }
        else        { apply_stay(e, s_nb, s_b); apply_ext(e, pr); cand[ip * V] = e.t; cand[ce] = -INFINITY; }
      } else {
        apply_stay(e, s_nb, s_b);
        cand[ip * V] = e.t;
      }
      sNB[ip] = e.nb; sB[ip] = e.b; sT[ip] = e.t;
    }
    __syncthreads();
    // 4. BHat = top-W by (prTotal desc, insertion index asc)
    int m = 0;
    for (int r = 0; r < W; ++r) {
      double bv = -INFINITY; int bi = 0x7fffffff;
      for (int c = tid; c < ncand; c += 256) {
        const double v = cand[c];
        if (v != -INFINITY && cand_better(v, c, bv, bi)) { bv = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || cand_better(ov, oi, bv, bi))) { bv = ov; bi = oi; }
      }
      if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
      __syncthreads();
      if (tid == 0) {
        double v = red_v[0]; int ix = red_i[0];
        for (int w = 1; w < 4; ++w)
          if (red_i[w] != 0x7fffffff && (ix == 0x7fffffff || cand_better(red_v[w], red_i[w], v, ix))) { v = red_v[w]; ix = red_i[w]; }
        s_best = ix;
        if (ix != 0x7fffffff) { sel[r] = ix; selv[r] = v; cand[ix] = -INFINITY; }
      }
      __syncthreads();
      if (s_best == 0x7fffffff) break;
      ++m;
    }
    // 5. materialise the new beam
    if (tid < m) {
      const int c = sel[tid];
      const int i = c / V, kk = c - i * V;
      if (kk == 0) {
        N.node[tid] = L.node[i]; N.len[tid] = L.len[i]; N.last[tid] = L.last[i]; N.par[tid] = L.par[i];
        N.pNB[tid] = sNB[i]; N.pB[tid] = sB[i]; N.pT[tid] = sT[i];
      } else {
        const int k = (kk - 1 < blank) ? kk - 1 : kk;
        int ip = -1;
        for (int j = 0; j < nb; ++j) if (mfrom[j] == i && L.last[j] == k) ip = j;
        if (ip >= 0) {   // this slot holds the merged entry of existing labelling ip (first touched as an extension)
          N.node[tid] = L.node[ip]; N.len[tid] = L.len[ip]; N.last[tid] = L.last[ip]; N.par[tid] = L.par[ip];
          N.pNB[tid] = sNB[ip]; N.pB[tid] = sB[ip]; N.pT[tid] = sT[ip];
        } else {
          // trie child lookup / insert: key = (parent node, symbol)
          const int parent = L.node[i];
          const unsigned long long key = ((unsigned long long)(unsigned)parent << 32) | (unsigned)k;
          unsigned h = (unsigned)mix64(key) & htmask;
          int id = -1;
          for (int probe = 0; probe <= htmask; ++probe) {
            const unsigned long long prev = atomicCAS(&keys[h], HT_EMPTY, key);
            if (prev == HT_EMPTY) {
              id = atomicAdd(&s_nodes, 1);
              if (id < a.max_nodes) { npar[id] = parent; nsym[id] = k; }
              ids[h] = id;
              break;
            }
            if (prev == key) { id = ids[h]; break; }
            h = (h + 1) & htmask;
          }
          N.node[tid] = id; N.len[tid] = L.len[i] + 1; N.last[tid] = k; N.par[tid] = parent;
          const double pr = selv[tid];
          N.pNB[tid] = pr; N.pB[tid] = LOG_ZERO; N.pT[tid] = pr;
        }
      }
    }
    __syncthreads();
    nb = m;
    cur ^= 1;
  }
  __syncthreads();
  // final LM step, length normalisation and best labelling (BeamSearch.py:130-151)
  BeamState &L = S[cur];
  if (status == 0 && tid == 0) {
    int st = 0;
    for (int r = 0; r < nb; ++r) if (L.len[r] == 0) st = 1;          // classes[y[-1]] on () -> IndexError
    if (s_nodes > a.max_nodes) st = 3;
    int best = -1; double bestv = 0.0;
    if (st == 0) {
      for (int r = 0; r < nb; ++r) {
        const double pr = L.pT[r] + a.lm[(size_t)L.last[r] * (V + 1) + V] * a.alpha;
        const double tot = log_add_prob(LOG_ZERO, pr);
        const int ln = L.len[r];
        const double nv = tot * (1.0 / (ln ? ln : 1));
        if (best < 0 || nv > bestv) { best = r; bestv = nv; }
      }
      const int ln = L.len[best];
      a.out_len[b] = ln; a.out_score[b] = bestv;
      int n = L.node[best];
      for (int i = ln - 1; i >= 0; --i) { a.out_ids[(size_t)b * T + i] = nsym[n]; n = npar[n]; }
    } else { a.out_len[b] = 0; a.out_score[b] = 0.0; }
    a.status[b] = st;
  } else if (tid == 0) {
    a.out_len[b] = 0; a.out_score[b] = 0.0; a.status[b] = status;
  }
}

struct BeamLayout { size_t keys, ids, npar, nsym, cand, total; int ht_size, max_nodes; };
BeamLayout beam_layout(int T, int B, int V, int W) {
  BeamLayout l;
  l.max_nodes = W * T + 2;
  int ht = 1024;
  while (ht < 2 * l.max_nodes) ht <<= 1;
  l.ht_size = ht;
  size_t off = 0;
  l.keys = off; off += align_up((size_t)B * ht * sizeof(unsigned long long), 256);
  l.ids = off;  off += align_up((size_t)B * ht * sizeof(int), 256);
  l.npar = off; off += align_up((size_t)B * l.max_nodes * sizeof(int), 256);
  l.nsym = off; off += align_up((size_t)B * l.max_nodes * sizeof(int), 256);
  l.cand = off; off += align_up((size_t)B * W * V * sizeof(double), 256);
  l.total = off;
  return l;
}

}  // namespace

extern "C" size_t ctcn_beam_ws_bytes(int T, int B, int V, int W) {
  if (T <= 0 || B <= 0 || V <= 0 || W <= 0) return 0;
  return beam_layout(T, B, V, W).total;
}

extern "C" int ctcn_beam_decode(const float *x, int input_is_prob, const int32_t *lens, const double *lm, double alpha, int W,
                                int blank, int32_t *out_ids, int32_t *out_len, double *out_score, int32_t *status, int T, int B,
                                int V, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && lens && lm && out_ids && out_len && out_score && status && ws, "ctcn_beam_decode: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 1 && blank >= 0 && blank < V, "ctcn_beam_decode: bad dims");
  if (W < 1 || W > BEAM_WMAX) { ctcn_set_error("ctcn_beam_decode: beam width %d outside [1,%d]", W, BEAM_WMAX); return CTCN_EUNSUPPORTED; }
  const BeamLayout l = beam_layout(T, B, V, W);
  if (ws_bytes < l.total) { ctcn_set_error("ctcn_beam_decode: workspace too small (%zu < %zu)", ws_bytes, l.total); return CTCN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  char *base = (char *)ws;
  CTCN_HIP(hipMemsetAsync(base + l.keys, 0xFF, (size_t)B * l.ht_size * sizeof(unsigned long long), st));
  BeamArgs a;
  a.x = x; a.input_is_prob = input_is_prob; a.lens = lens; a.lm = lm; a.alpha = alpha; a.W = W; a.blank = blank;
  a.out_ids = out_ids; a.out_len = out_len; a.out_score = out_score; a.status = status; a.T = T; a.B = B; a.V = V;
  a.ht_keys = (unsigned long long *)(base + l.keys); a.ht_ids = (int *)(base + l.ids);
  a.node_par = (int *)(base + l.npar); a.node_sym = (int *)(base + l.nsym); a.cand_global = (double *)(base + l.cand);
  a.ht_size = l.ht_size; a.max_nodes = l.max_nodes;
  const size_t cand_bytes = (size_t)W * V * sizeof(double);
  a.cand_in_lds = cand_bytes <= 32 * 1024 ? 1 : 0;
  const size_t sm = (size_t)V * sizeof(double) + (a.cand_in_lds ? cand_bytes : 0);
  hipLaunchKernelGGL(beam_kernel, dim3(B), dim3(256), sm, st, a);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}
