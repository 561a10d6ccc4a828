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

// Widest beam of the generic kernel (round 6: 1 024, was 256 -- the reference takes any beam_width, ctcDecoder.py:170 / BeamSearch.py:96).  The
// beam state lives in DYNAMIC LDS sized for the call's width (120 B per beam slot + 12 B per selection survivor: 42 KB at W <= 256, 146 KB
// at W = 1 024, which gfx950's 160 KB per workgroup still hold); beyond that the state would have to live in global memory.
constexpr int BEAM_WMAX = 1024;
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

struct BeamState {   // one copy of the beam (BHat) in LDS: arrays of `wcap` entries carved out of the kernel's dynamic shared memory
  int *node, *len, *last, *par;
  double *pB, *pNB, *pT;
};

struct BeamArgs {
  const float *x; int input_is_prob; const int32_t *lens; const double *lm; double alpha; int W, blank;
  int32_t *out_ids, *out_len; double *out_score; int32_t *status; int T, B, V;
  int nbest; int32_t *out_count;        // ctcn_beam_decode_nbest: the `nbest` best labellings per utterance (outputs [B][nbest]...), their number in out_count
  unsigned long long *ht_keys; int *ht_ids; int *node_par; int *node_sym; double *cand_global;
  int *node_slot, *cand_owner;   // beam_kernel: beam slot of a trie node / beam slot whose labelling merges with a candidate (lookups that replace scans of the beam)
  int ht_size, max_nodes, cand_in_lds;
  int wcap, smax;       // beam_kernel: capacity of the beam-state arrays (W rounded up to 64) and of the selection's survivor list
  int bitonic;          // beam_kernel: rank up to 256 survivors by a bitonic sort (option beam_bitonic, default 1) instead of counting pairs
  int lm_in_lds;        // beam_kernel: the (V+1)^2 ln-prob table is copied into dynamic LDS behind the state arrays (when the launch's budget holds it)
#ifdef CTCN_BEAM_STATS
  long long *stats;     // development instrumentation (tools/mb_beam.py generic): cycles per phase of workgroup 0, thread 0
#endif
};
#ifdef CTCN_BEAM_STATS
#define GSTAMP(i) do { if (b == 0 && tid == 0) { const long long now_ = clock64(); gst[i] += now_ - glast; glast = now_; } } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ bool cand_better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }
// order-preserving map double -> uint64 for the selection's bound (v > w <=> dkey(v) > dkey(w); -0.0 folded onto +0.0, which compare equal; no NaN
// among the scores); never 0 for a finite value
__device__ __forceinline__ unsigned long long dkey(double v) {
  const unsigned long long bts = (unsigned long long)__double_as_longlong(v == 0.0 ? 0.0 : v);
  return (bts >> 63) ? ~bts : (bts | 0x8000000000000000ull);
}

// Sum / 64-bit maximum over each row of 16 lanes by DPP (no LDS crossbar: a ds_bpermute butterfly is a chain of ~150-cycle hops under load);
// afterwards every lane of a row holds the row's result
__device__ __forceinline__ int row16_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);    // row_mirror
  return v;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_u64_step(unsigned long long x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(x & 0xffffffffull), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), CTRL, 0xf, 0xf, true);
  const unsigned long long o = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
  return o > x ? o : x;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long x) {
  x = dpp_max_u64_step<0xB1>(x);
  x = dpp_max_u64_step<0x4E>(x);
  x = dpp_max_u64_step<0x141>(x);
  x = dpp_max_u64_step<0x140>(x);
  return x;
}
__device__ __forceinline__ int wave_sum_rows(int v) {               // wave-wide sum, uniform result
  v = row16_sum(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// Lane i <- lane i ^ J of a 32-bit value without the LDS crossbar: DPP quad permutes / row rotation inside a row of 16 lanes, the gfx950
// v_permlane16_swap / v_permlane32_swap across rows and halves (swap(x, x): result 0 holds the lower partner's rows, result 1 the upper's).
// J = 4 has no single pattern: both row shifts by 4 are taken and `pick_shl` (probed once by the caller: which of the two reads lane i ^ 4)
// selects.
template <int J>
__device__ __forceinline__ unsigned xor_lane(unsigned x, int lane, bool pick_shl) {
  if (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
  if (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
  if (J == 4) {
    const unsigned a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0xf, true);    // row_shl:4
    const unsigned c = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    return pick_shl ? a : c;
  }
  if (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);      // row_ror:8
  if (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); return (lane & 16) ? r[0] : r[1]; }
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return (lane & 32) ? r[0] : r[1];
}
// (key a, index ia) sorts before (key b, index ib): key descending, index ascending.  (A 96-bit subtract-with-borrow chain through
// __builtin_subc was measured: slower than what hipcc makes of the two 64-bit compares.)
__device__ __forceinline__ bool elem_before(unsigned ahi, unsigned alo, int ia, unsigned bhi, unsigned blo, int ib) {
  const unsigned long long ka = ((unsigned long long)ahi << 32) | alo, kb = ((unsigned long long)bhi << 32) | blo;
  return ka > kb || (ka == kb && ia < ib);
}
// one compare-exchange step of the bitonic network on (64-bit key descending, index ascending) elements, partner lane ^ J; `keep_better`:
// this lane keeps the better element of the pair.  Equal elements (only the padding) may swap: they are identical.
template <int J>
__device__ __forceinline__ void bitonic_step(unsigned &khi, unsigned &klo, int &ix, bool keep_better, int lane, bool pick_shl) {
  const unsigned ohi = xor_lane<J>(khi, lane, pick_shl), olo = xor_lane<J>(klo, lane, pick_shl);
  const int oi = (int)xor_lane<J>((unsigned)ix, lane, pick_shl);
  const bool take = elem_before(ohi, olo, oi, khi, klo, ix) == keep_better;
  khi = take ? ohi : khi; klo = take ? olo : klo; ix = take ? oi : ix;
}

// NT threads per utterance: 256 for small tables, 1 024 beyond W = 64 or 3 500 candidates per frame (round 5: every phase of a frame is a loop over nb * V candidates or over the beam, and the
// kernel ran one wave per SIMD -- nothing hid an LDS or L2 latency)
template <int NT>
__global__ __launch_bounds__(NT) void beam_kernel(BeamArgs a) {
  constexpr int NWV = NT / 64;
  extern __shared__ __attribute__((aligned(16))) double dsm[];   // lg[V] | cand[W*V] (if it fits) | beam state, selection lists (below)
  __shared__ double red_v[NWV];
  __shared__ int red_i[NWV];
  __shared__ unsigned long long wthr[NWV];                     // selection: every wave's bound
  __shared__ int wcnt[NWV][NWV];                               //            ... and how many of wave w's maxima reach wave j's
  __shared__ int s_flag, s_nodes, s_best;
  __shared__ int s_scnt, s_ovf;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: per-wave decisions are scalar branches)
  const int b = blockIdx.x, V = a.V, W = a.W, B = a.B, T = a.T, blank = a.blank;
  double *lg = dsm;
  double *cand = a.cand_in_lds ? dsm + V : a.cand_global + (size_t)b * W * V;
  // dynamic LDS behind lg / cand: doubles first (2 x {pB, pNB, pT}, sNB, sB, sT, selv: 10 x wcap; sv_v: smax), then ints (2 x {node, len,
  // last, par}, mfrom, sel: 10 x wcap; sv_i: smax) -- the host sizes the launch with the same formula (beam_state_lds)
  const int wcap = a.wcap, SEL_SMAX = a.smax;
  double *const dbase = dsm + V + (a.cand_in_lds ? W * V : 0);
  int *const ibase = reinterpret_cast<int *>(dbase + 10 * wcap + SEL_SMAX);
  auto beam_state = [&](int k) {
    BeamState st;
    st.pB = dbase + (3 * k) * wcap; st.pNB = st.pB + wcap; st.pT = st.pNB + wcap;
    st.node = ibase + (4 * k) * wcap; st.len = st.node + wcap; st.last = st.len + wcap; st.par = st.last + wcap;
    return st;
  };
  double *const sNB = dbase + 6 * wcap, *const sB = sNB + wcap, *const sT = sB + wcap, *const selv = sT + wcap;
  double *const sv_v = selv + wcap;                            // selection: survivors of the pruning bound (value | candidate index)
  int *const mfrom = ibase + 8 * wcap, *const sel = mfrom + wcap, *const sv_i = sel + wcap;
  // (round 6) the LM table in LDS when the launch has room for it: one LDS read instead of a global gather per candidate and frame; the raw
  // ln-probs are copied, the product with alpha is formed per use exactly as before
  const double *lmt = a.lm;
  if (a.lm_in_lds) {
    double *lml = reinterpret_cast<double *>(reinterpret_cast<char *>(sv_i + SEL_SMAX) + ((8 - ((size_t)(10 * wcap + SEL_SMAX) * 4) % 8) % 8));
    for (int i = tid; i < (V + 1) * (V + 1); i += NT) lml[i] = a.lm[i];
    lmt = lml;
  }
  unsigned long long *keys = a.ht_keys + (size_t)b * a.ht_size;
  int *ids = a.ht_ids + (size_t)b * a.ht_size;
  int *npar = a.node_par + (size_t)b * a.max_nodes;
  int *nsym = a.node_sym + (size_t)b * a.max_nodes;
  // Lookups instead of scans of the beam (last session of round 6; both were ~200 LDS reads per thread and frame at the reference's W = 200):
  //   nslot[node] = beam slot that holds trie node `node` -- written for every entry of a new beam, read by step 2 of the next frame;
  //   cown[c]     = beam slot whose labelling merges with extension candidate c -- written by step 3b, read by step 5 of the same frame.
  // Neither table is initialised or stamped: a value is USED only if the slot it names passes the very test the scan applied (L.node[s] ==
  // pnode; mfrom[s] == i && L.last[s] == k), which at most one slot of a beam can pass -- and for a node / candidate that does have such a
  // slot the table was written this frame.  Anything else (stale, never written) fails the test exactly as the scan found nothing.
  // c / V for candidate indices c < W * V: one multiply-high by ceil(2^32 / V) (exact while c * V < 2^32: W * V * V = 2^20 at the reference's width and vocabulary) -- the runtime
  // division hipcc emits is ~25 instructions, and step 3a pays it per candidate: the kernel's parallel phases are instruction-issue bound
  const unsigned vmagic = 0xffffffffu / (unsigned)V + 1u;
  const bool vmagic_ok = (unsigned long long)W * V * V < (1ull << 32);         // (a vocabulary of thousands: the plain division)
  auto div_v = [&](int c) { return vmagic_ok ? (int)__umulhi((unsigned)c, vmagic) : c / V; };
  int *nslot = a.node_slot + (size_t)b * a.max_nodes;
  int *cown = a.cand_owner + (size_t)b * W * V;
  const int htmask = a.ht_size - 1;

  int cur = 0, nb = 1, status = 0;
#ifdef CTCN_BEAM_STATS
  long long gst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, glast = clock64(), gframes = 0;
#endif
  if (tid == 0) {
    const BeamState S0 = beam_state(0);
    S0.node[0] = 0; S0.len[0] = 0; S0.last[0] = -1; S0.par[0] = -1;
    S0.pB[0] = 0.0; S0.pNB[0] = LOG_ZERO; S0.pT[0] = 0.0;   // BeamSearch.py:83-87
    s_nodes = 1; s_flag = 0;
    npar[0] = -1; nsym[0] = -1;
    nslot[0] = 0;
  }
  __syncthreads();
  const int nframes = min(max(a.lens[b], 0), T);
  // p(blank) of 64 frames per load (lane j: frame chunk0 + j): the skip test of a frame that is not processed costs a register read, not a
  // round trip to memory (peaky posteriors skip two frames of three); the previous frame's value (repeat rule) is carried along
  float pb_lane = 0.0f, pb_before = 0.0f;
  int chunk0 = -64;
  for (int t = 0; t < nframes; ++t) {
    const float *row = a.x + ((size_t)t * B + b) * V;
    if (t >= chunk0 + 64) {
      chunk0 = t;
      const int tl = min(t + lane, nframes - 1);
      const float *r = a.x + ((size_t)tl * B + b) * V;
      pb_lane = a.input_is_prob ? r[blank] : expf(r[blank]);
    }
    const float pblank = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pb_lane), __builtin_amdgcn_readfirstlane(t - chunk0)));
    const float pprev = pb_before;
    pb_before = pblank;
    if ((1.0f - pblank) < 0.1f) continue;                         // BeamSearch.py:93-94 (float32 compare)
    const BeamState L = beam_state(cur), N = beam_state(cur ^ 1);
    GSTAMP(0);
#ifdef CTCN_BEAM_STATS
    ++gframes;
#endif
    // 1. ln of the frame's probabilities (math.log of the float32 value widened to double)
    for (int k = tid; k < V; k += NT) {
      const float p = a.input_is_prob ? row[k] : expf(row[k]);
      if (!(p > 0.0f)) s_flag = 2;
      lg[k] = log((double)p);
    }
    const bool rep_ok = t > 0 && pprev < 0.9f;                     // BeamSearch.py:63 (float32 compare)
    // 2. which beam (if any) is the parent labelling of beam i' -> its extension by last(i') merges with i'
    if (tid < nb) {
      int m = -1;
      if (L.len[tid] > 0) {
        const int pnode = L.par[tid];
        if ((unsigned)pnode < (unsigned)a.max_nodes) {
          const int sl = nslot[pnode];
          if ((unsigned)sl < (unsigned)nb && L.node[sl] == pnode) m = sl;
        }
      }
      mfrom[tid] = m;
    }
    __syncthreads();
    if (s_flag == 2) { status = 2; break; }
    GSTAMP(1);
    // 3a. extension scores (calcExtPr), candidate slot c = i*V + 1 + kk  (kk enumerates k != blank in order)
    const int ncand = nb * V;
    // (four candidates per trip with 256 threads -- two with 1 024 --: the slot reads, then the LM gathers -- global memory -- of all four are issued before the first is used; the same
    // expression per candidate)
    constexpr int EU = 4;                 // (candidates per trip: a trip is one round trip to the LM table in L2; W = 200 at 1 024 threads: 4 trips, was 7 with two per trip)
    for (int c0 = tid; c0 < ncand; c0 += EU * NT) {
      int ci[EU], ck[EU], cl[EU], cln[EU];
      double lmv[EU];
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = min(c0 + u * NT, ncand - 1);
        ci[u] = div_v(c);
        const int kk = c - ci[u] * V;
        ck[u] = kk == 0 ? -1 : ((kk - 1 < blank) ? kk - 1 : kk);
        cln[u] = L.len[ci[u]]; cl[u] = L.last[ci[u]];
      }
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c1 = cln[u] > 0 ? cl[u] : V;
        lmv[u] = lmt[(size_t)c1 * (V + 1) + max(ck[u], 0)];
      }
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int c = c0 + u * NT;
        if (c < ncand && ck[u] >= 0) {
          const int i = ci[u], k = ck[u];
          const double bigram = lmv[u] * a.alpha;
          const double base = (cln[u] > 0 && cl[u] == k && rep_ok) ? L.pB[i] : L.pT[i];
          cand[c] = lg[k] + bigram + base;
        }
      }
    }
    __syncthreads();
    GSTAMP(2);
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
        cown[ce] = ip;
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
    GSTAMP(3);
    // 4. BHat = top-W by (prTotal desc, insertion index asc).
    // Round 5 (wide beams: the reference's class default is W = 200, ctcDecoder.py:170): NOT W block-wide arg-max rounds any more -- each a scan
    // of all nb * V candidates (99 KB of global memory at W = 200, V = 62), a shuffle tree and two barriers -- but the selection scheme of the
    // fast kernel in its plainest form:
    //  (a) a lower bound of the W-th best candidate: every thread keeps the best of its candidates (two, of disjoint halves of its share, with
    //      256 threads), every WAVE finds the k-th largest of its maxima, k = ceil(W / waves), by a bitwise binary search on the order-preserving
    //      64-bit keys (ballot + popcount: no LDS, no barrier), and the bound is the smallest of the waves' values -- at least waves * k >= W
    //      distinct candidates reach it, so the true top W all do (exact pruning; a tighter bound only means fewer survivors);
    //  (b) the survivors (about W, more when values tie at the bound) are compacted into LDS;
    //  (c) each is ranked by counting the survivors that beat it under the same explicit order cand_better() the rounds used.
    // Same set, same order.  The rounds remain as the fallback (a wave with fewer than k candidates, or more than SEL_SMAX survivors: exact ties
    // by the hundred).
    int m = 0;
    bool ranked = false;
    {
      // (the candidate table is in global memory from W * V > 4 096 on: eight loads in flight per thread, or every scan is a chain of L2 round trips)
      auto scan = [&](auto &&visit) {
        constexpr int SU = NT >= 1024 ? 4 : 8;          // (1 024 threads: 128 registers each, and a fourth of the candidates per thread)
        int c = tid;
        for (; c + (SU - 1) * NT < ncand; c += SU * NT) {
          double v[SU];
#pragma unroll
          for (int u = 0; u < SU; ++u) v[u] = cand[c + NT * u];
#pragma unroll
          for (int u = 0; u < SU; ++u) visit(v[u], c + NT * u);
        }
        for (; c < ncand; c += NT) visit(cand[c], c);
      };
      constexpr int NH = NT >= 1024 ? 1 : 2;                   // maxima per thread
      unsigned long long tk[NH];
      int nval = 0;
      {
        // (last session of round 6: the maxima in the double domain -- one v_max_f64 per candidate; fmax drops a NaN as "v > -inf" does -- and
        // ONE conversion to the order-preserving key at the end: the scans are instruction-issue bound, ~25 -> ~8 instructions per candidate)
        double mx[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) mx[h] = -INFINITY;
        scan([&](double v, int c) {
          nval += v > -INFINITY ? 1 : 0;
          if (NH == 2 && ((c / NT) & 1)) mx[NH - 1] = fmax(mx[NH - 1], v);
          else mx[0] = fmax(mx[0], v);
        });
#pragma unroll
        for (int h = 0; h < NH; ++h) tk[h] = mx[h] > -INFINITY ? dkey(mx[h]) : 0ull;      // (no candidate: below every key of a finite value)
      }
      GSTAMP(8);
      nval = wave_sum_rows(nval);
      if (lane == 0) red_i[wave] = nval;
      {
        const int kth = (W + NWV - 1) / NWV;
        unsigned long long p = 0ull;
        // Bitwise search, two bits per step (the three thresholds of a step are independent ballots), bits 63 .. 36 only: the result is a LOWER
        // bound of the wave's k-th largest maximum either way, 2^-16 relative below it at most -- the survivor count did not move (225 | 207 per
        // frame at W = 200) while the search is the most instruction-heavy part of the bound phase, which is issue-bound (sixteen waves at once:
        // the first barrier waits ~3 000 cycles for the slowest wave's scan + search): 14 steps instead of 18, 44 in round 5.
        for (int bit = 62; bit >= 36; bit -= 2) {
          const unsigned long long t1 = p | (1ull << bit), t2 = p | (2ull << bit), t3 = p | (3ull << bit);
          int c1 = __popcll(__ballot(tk[0] >= t1)), c2 = __popcll(__ballot(tk[0] >= t2)), c3 = __popcll(__ballot(tk[0] >= t3));
          if (NH == 2) {
            c1 += __popcll(__ballot(tk[NH - 1] >= t1)); c2 += __popcll(__ballot(tk[NH - 1] >= t2)); c3 += __popcll(__ballot(tk[NH - 1] >= t3));
          }
          p = c3 >= kth ? t3 : (c2 >= kth ? t2 : (c1 >= kth ? t1 : p));
        }
        if (lane == 0) wthr[wave] = p;
      }
      if (tid == 0) { s_scnt = 0; s_ovf = 0; }
      GSTAMP(12);
      __syncthreads();
      GSTAMP(13);
      // the waves' values are NWV candidate bounds; the smallest is always valid, a larger one is valid whenever W of ALL the maxima still reach
      // it: every wave counts its own maxima against every candidate (NWV ballots), the largest candidate with a block-wide count >= W wins --
      // the survivors drop from ~1.9 W to ~1.2 W at W = 200, and a wave that has no k candidates (narrow beam) no longer voids the bound.
      // (round 6: lane j keeps candidate j in registers and v_readlane broadcasts it -- no chain of dependent LDS reads; every wave forms the
      // block-wide counts itself from the waves' rows, so two barriers instead of three)
      const unsigned long long tj = wthr[lane < NWV ? lane : 0];
      const int tj_lo = (int)(unsigned)(tj & 0xffffffffull), tj_hi = (int)(unsigned)(tj >> 32);
      int mycnt = 0;
      for (int j = 0; j < NWV; ++j) {
        const unsigned long long t = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(tj_hi, j) << 32) | (unsigned)__builtin_amdgcn_readlane(tj_lo, j);
        int cnt = __popcll(__ballot(tk[0] >= t));
        if (NH == 2) cnt += __popcll(__ballot(tk[NH - 1] >= t));
        mycnt = lane == j ? (t != 0ull ? cnt : 0) : mycnt;
      }
      if (lane < NWV) wcnt[wave][lane] = mycnt;
      GSTAMP(14);
      __syncthreads();
      int reach = 0, total = 0;
      {
        const int jl = lane < NWV ? lane : 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) reach += wcnt[w][jl];
        total = lane < NWV ? red_i[jl] : 0;
      }
      unsigned long long theta = (lane < NWV && reach >= W) ? tj : 0ull;
      static_assert(NWV <= 16, "the waves' bounds sit in the first row of 16 lanes");
      total = __builtin_amdgcn_readfirstlane(row16_sum(total));
      theta = row16_max_u64(theta);
      theta = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(theta >> 32)) << 32) |
              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(theta & 0xffffffffull));
      GSTAMP(9);
      // (no candidate bound that W maxima reach -- the first frames, when the beam is still narrow --: then everything valid is ranked, if it fits)
      const bool prune = total > W && theta != 0ull;
      if (prune || total <= SEL_SMAX) {
        // one scan: a thread's first two survivors wait in registers (it rarely has more than one), a second scan only for a thread with more
        // The bound as a double: dkey is an order-preserving bijection of the finite values, so dkey(v) >= theta <=> v >= thd (one
        // v_cmp_ge_f64; a NaN fails both; theta >= 2^52 > dkey(-inf) whenever a finite maximum produced it, so "v > -inf" is implied).  Without
        // a bound every finite candidate is ranked: v >= -DBL_MAX.  Only the INDICES of a thread's first two survivors wait in registers; their
        // values are read again when they are stored.
        double thd = -1.7976931348623157e308;
        if (prune) {
          const unsigned long long tb = (theta >> 63) ? (theta & 0x7fffffffffffffffull) : ~theta;
          thd = __longlong_as_double((long long)tb);
        }
        int cnt = 0, kc0 = 0, kc1 = 0;
        scan([&](double v, int c) {
          if (v >= thd) {
            kc0 = cnt == 0 ? c : kc0;
            kc1 = cnt == 1 ? c : kc1;
            ++cnt;
          }
        });
        // one atomic per WAVE (a thread rarely keeps more than two survivors: then its position follows from two ballots; a wave with such a
        // thread takes the per-thread atomics) -- ~225 same-address LDS atomics per frame were a serial chain in front of the barrier
        int pos = 0;
        {
          const unsigned long long b1 = __ballot(cnt >= 1), b2 = __ballot(cnt >= 2), b3 = __ballot(cnt >= 3);
          if (b3 == 0ull) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const int wtot = __popcll(b1) + __popcll(b2);
            int wbase = 0;
            if (lane == 0 && wtot > 0) wbase = atomicAdd(&s_scnt, wtot);
            pos = __builtin_amdgcn_readfirstlane(wbase) + __popcll(b1 & below) + __popcll(b2 & below);
          } else {
            pos = cnt ? atomicAdd(&s_scnt, cnt) : 0;
          }
        }
        if (cnt && pos + cnt > SEL_SMAX) s_ovf = 1;
        else if (cnt <= 2) {
          if (cnt > 0) { sv_v[pos] = cand[kc0]; sv_i[pos] = kc0; }
          if (cnt > 1) { sv_v[pos + 1] = cand[kc1]; sv_i[pos + 1] = kc1; }
        } else scan([&](double v, int c) { if (v >= thd) { sv_v[pos] = v; sv_i[pos] = c; ++pos; } });
        __syncthreads();
        GSTAMP(10);
#ifdef CTCN_BEAM_STATS
        if (b == 0 && tid == 0) { gst[6] += s_scnt; gst[7] += s_ovf ? 1 : 0; }
#endif
        if (!s_ovf) {
          // P threads per survivor (a power of two, neighbouring lanes), thread part p counts among the survivors p, p + P, ...; eight entries
          // are read before they are compared (a loop of dependent LDS reads costs a full LDS latency per entry); butterfly sum over the P lanes
          const int S = s_scnt;
          if (S <= 256 && a.bitonic) {
            // (last session of round 6) Up to 256 survivors -- the reference's W = 200 leaves ~1.1 W -- are SORTED, one per thread of waves 0-3, by a
            // bitonic network under the same strict order cand_better() (value descending, candidate index ascending; the padding compares
            // equal to itself and worse than every survivor, so it never moves in front of one): 36 compare-exchange steps, 33 of them inside a
            // wave, instead of S^2 = 50 000 comparisons through ~1 800 LDS reads per frame.  Position r of the sorted row IS rank r.
            unsigned khi = 0u, klo = 0u; int ix = 0x7fffffff;       // (padding: key 0 is below the key of every finite value)
            const bool act = wave < 4;                          // (scalar: the other waves only join the barriers of the three cross-wave steps)
            if (act && tid < S) { const unsigned long long k0 = dkey(sv_v[tid]); khi = (unsigned)(k0 >> 32); klo = (unsigned)k0; ix = sv_i[tid]; }
            const bool pick_shl = (unsigned)__builtin_amdgcn_update_dpp(-1, lane, 0x104, 0xf, 0xf, false) == (unsigned)(lane ^ 4);
            unsigned *const xk = reinterpret_cast<unsigned *>(sv_v);     // cross-wave steps: (key hi, key lo) pairs | indices in the survivor arrays
            // element tid of a K-block keeps the better of a pair iff (it is the lower partner) == (its block ends best-first)
#define CTCN_BSTEP(J, K) bitonic_step<J>(khi, klo, ix, ((tid & (J)) == 0) == ((tid & (K)) == 0), lane, pick_shl)
            auto lds_step = [&](int j, int k) {
              if (act) { xk[2 * tid] = khi; xk[2 * tid + 1] = klo; sv_i[tid] = ix; }
              __syncthreads();
              if (act) {
                const int pt = tid ^ j;
                const unsigned ohi = xk[2 * pt], olo = xk[2 * pt + 1]; const int oi = sv_i[pt];
                const bool take = elem_before(ohi, olo, oi, khi, klo, ix) == (((tid & j) == 0) == ((tid & k) == 0));
                khi = take ? ohi : khi; klo = take ? olo : klo; ix = take ? oi : ix;
              }
              __syncthreads();
            };
            auto tail = [&](int k) {                           // the in-wave steps j = 32 .. 1 of block size k >= 64
              if (act) { CTCN_BSTEP(32, k); CTCN_BSTEP(16, k); CTCN_BSTEP(8, k); CTCN_BSTEP(4, k); CTCN_BSTEP(2, k); CTCN_BSTEP(1, k); }
            };
            if (act) {
              CTCN_BSTEP(1, 2);
              CTCN_BSTEP(2, 4); CTCN_BSTEP(1, 4);
              CTCN_BSTEP(4, 8); CTCN_BSTEP(2, 8); CTCN_BSTEP(1, 8);
              CTCN_BSTEP(8, 16); CTCN_BSTEP(4, 16); CTCN_BSTEP(2, 16); CTCN_BSTEP(1, 16);
              CTCN_BSTEP(16, 32); CTCN_BSTEP(8, 32); CTCN_BSTEP(4, 32); CTCN_BSTEP(2, 32); CTCN_BSTEP(1, 32);
            }
            tail(64);
            lds_step(64, 128); tail(128);
            lds_step(128, 256); lds_step(64, 256); tail(256);
#undef CTCN_BSTEP
            // (lds_step ends with a barrier: the exchange area is free again; the value is read back from the candidate table, which this
            // path leaves untouched -- bit for bit the double that was scored, where inverting the key would fold -0.0 onto +0.0)
            if (act && tid < S && tid < W) { sel[tid] = ix; selv[tid] = cand[ix]; }
          } else {
          int P = 1, lgP = 0;
          while (2 * P * S <= NT && P < 16) { P *= 2; ++lgP; }
          for (int e0 = 0; e0 < S; e0 += NT >> lgP) {
            const int e = e0 + (tid >> lgP), part = tid & (P - 1);
            const bool have = e < S;
            const double mv = sv_v[have ? e : 0]; const int mi = sv_i[have ? e : 0];
            constexpr int RU = NT >= 1024 ? 4 : 8;
            int rank = 0, q = part;
            for (; q + (RU - 1) * P < S; q += RU * P) {
              double qv[RU]; int qi[RU];
#pragma unroll
              for (int u = 0; u < RU; ++u) { qv[u] = sv_v[q + u * P]; qi[u] = sv_i[q + u * P]; }
#pragma unroll
              for (int u = 0; u < RU; ++u) rank += cand_better(qv[u], qi[u], mv, mi) ? 1 : 0;
            }
            for (; q < S; q += P) rank += cand_better(sv_v[q], sv_i[q], mv, mi) ? 1 : 0;
            for (int o = 1; o < P; o <<= 1) rank += __shfl_xor(rank, o, 64);
            if (have && part == 0 && rank < W) { sel[rank] = mi; selv[rank] = mv; }
          }
          }
          m = min(W, total);
          ranked = true;
        }
        __syncthreads();
        GSTAMP(11);
      }
    }
    // (the rounds reuse red_i[], which every thread has just summed into `total` -- when the ranking block was skipped no barrier
    // separates those reads from round 0's writes; `ranked` is uniform.  A NaN score counts as "no candidate" everywhere (v > -inf).)
    if (!ranked) __syncthreads();
    for (int r = 0; r < W && !ranked; ++r) {
      double bv = -INFINITY; int bi = 0x7fffffff;
      for (int c = tid; c < ncand; c += NT) {
        const double v = cand[c];
        if (v > -INFINITY && cand_better(v, c, bv, bi)) { bv = v; bi = c; }
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
        for (int w = 1; w < NWV; ++w)
          if (red_i[w] != 0x7fffffff && (ix == 0x7fffffff || cand_better(red_v[w], red_i[w], v, ix))) { v = red_v[w]; ix = red_i[w]; }
        s_best = ix;
        if (ix != 0x7fffffff) { sel[r] = ix; selv[r] = v; cand[ix] = -INFINITY; }
      }
      __syncthreads();
      if (s_best == 0x7fffffff) break;
      ++m;
    }
    GSTAMP(4);
    // 5. materialise the new beam
    if (tid < m) {
      const int c = sel[tid];
      const int i = div_v(c), kk = c - i * V;
      if (kk == 0) {
        N.node[tid] = L.node[i]; N.len[tid] = L.len[i]; N.last[tid] = L.last[i]; N.par[tid] = L.par[i];
        N.pNB[tid] = sNB[i]; N.pB[tid] = sB[i]; N.pT[tid] = sT[i];
      } else {
        const int k = (kk - 1 < blank) ? kk - 1 : kk;
        int ip = -1;
        {
          const int ow = cown[c];
          if ((unsigned)ow < (unsigned)nb && mfrom[ow] == i && L.last[ow] == k) ip = ow;
        }
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
      const int nd = N.node[tid];
      if ((unsigned)nd < (unsigned)a.max_nodes) nslot[nd] = tid;     // (step 2 of the next frame finds a parent labelling's slot here)
    }
    __syncthreads();
    GSTAMP(5);
    nb = m;
    cur ^= 1;
  }
  __syncthreads();
#ifdef CTCN_BEAM_STATS
  if (a.stats && b == 0 && tid == 0) { for (int i = 0; i < 16; ++i) a.stats[32 + i] = gst[i]; a.stats[48] = gframes; }
#endif
  // final LM step, length normalisation and best labelling (BeamSearch.py:130-151)
  const BeamState L = beam_state(cur);
  if (status == 0 && tid == 0) {
    int st = 0;
    for (int r = 0; r < nb; ++r) if (L.len[r] == 0) st = 1;          // classes[y[-1]] on () -> IndexError
    if (s_nodes > a.max_nodes) st = 3;
    const int NB = a.nbest;
    if (st == 0) {
      // normalised scores (BeamSearch.py:147: prTotal / labelling length), then `last.sort()[0:nbest]` (:150, the reference keeps [0]): a stable
      // descending sort -- among equal scores the earlier entry (the order of BHat) comes first.  L.pT is reused for the scores, L.par as "taken"
      for (int r = 0; r < nb; ++r) {
        const double pr = L.pT[r] + lmt[(size_t)L.last[r] * (V + 1) + V] * a.alpha;
        const double tot = log_add_prob(LOG_ZERO, pr);
        const int ln = L.len[r];
        L.pT[r] = tot * (1.0 / (ln ? ln : 1));
        L.par[r] = 0;
      }
      const int nout = min(NB, nb);
      for (int k = 0; k < nout; ++k) {
        int best = -1; double bestv = 0.0;
        for (int r = 0; r < nb; ++r)
          if (!L.par[r] && (best < 0 || L.pT[r] > bestv)) { best = r; bestv = L.pT[r]; }
        L.par[best] = 1;
        const int ln = L.len[best];
        const size_t o = (size_t)b * NB + k;
        a.out_len[o] = ln; a.out_score[o] = bestv;
        int n = L.node[best];
        for (int i = ln - 1; i >= 0; --i) { a.out_ids[o * T + i] = nsym[n]; n = npar[n]; }
      }
      for (int k = nout; k < NB; ++k) { a.out_len[(size_t)b * NB + k] = 0; a.out_score[(size_t)b * NB + k] = 0.0; }
      if (a.out_count) a.out_count[b] = nout;
    } else {
      for (int k = 0; k < NB; ++k) { a.out_len[(size_t)b * NB + k] = 0; a.out_score[(size_t)b * NB + k] = 0.0; }
      if (a.out_count) a.out_count[b] = 0;
    }
    a.status[b] = st;
  } else if (tid == 0) {
    for (int k = 0; k < a.nbest; ++k) { a.out_len[(size_t)b * a.nbest + k] = 0; a.out_score[(size_t)b * a.nbest + k] = 0.0; }
    if (a.out_count) a.out_count[b] = 0;
    a.status[b] = status;
  }
}


// =====================================================================================================================
// Fast path (W <= 60, W*V <= 3328, V <= 256): the same search, restructured around the per-frame latency chain.
// The generic kernel above spends ~39 us per processed frame (cfg5): W block-wide arg-max rounds (two __syncthreads + a
// serial 4-way compare each), the double-precision log of the frame, LM gathers from global memory, trie CAS round trips
// to L2 (~1.9 us each), and -- everywhere -- chains of dependent LDS reads (~100 cycles per hop).  Here:
//   * beam_prep_kernel (fully parallel, one wave per (frame, utterance) row) takes everything that does not depend on the
//     beam state out of the chain: ln p as double for every class, p(blank), the "some p is not > 0" flag;
//   * the workgroup compacts its list of processed frames once (the skip rule 1 - p_blank < 0.1 needs no beam state), keeps
//     alpha * LM (31.7 KB at V = 62) in LDS, and every thread prefetches the next frame's ln p of ITS candidate classes;
//   * the beam lives in the registers of wave 0 (lane = beam slot); cross-slot reads are v_readlane / ds_bpermute, not LDS
//     round trips; the per-slot values the candidate scoring needs are mirrored into a small LDS record;
//   * top-W without sorting: waves 3..15 hold the <= W*V candidates (<= 4 per thread); every 16-lane row reduces its largest
//     order-preserving key (high word) on the DPP network, the W-th largest row maximum is a lower bound of the W-th best candidate
//     (exact pruning), the survivors (typically W + a few) are compacted into LDS and ranked by counting with the explicit
//     (score desc, index asc) order of BeamState.sort; more than 256 survivors fall back to block-wide arg-max rounds;
//   * trie: a 16K-slot table IN LDS (node id = slot + 1, entry = {parent id + 1 : 15 | symbol : 17}, double hashing, one ds_cmpst
//     per probe); only when it fills beyond 3/4 do new labellings go to the global-memory table of the generic kernel (64-bit
//     entries {parent : 24 | symbol : 16 | id : 24});
//   * round 4: the per-frame serial chain is split over three waves.  Wave 1 owns the trie (a node id is needed a frame later, as
//     the parent id of the labelling's children); wave 0 recognises "this slot's parent labelling was created in this frame" from
//     (grandparent id, parent's last class) keys, which need old ids only.  Wave 2 computes the first level of the stay entries'
//     log-adds (it needs no parent slot) next to wave 0's new-beam work; wave 0 publishes the beam record through an LDS flag, not a
//     barrier.  cfg5 batch 1 300 -> 1 098 us (peaky), 3 949 -> 2 796 us (flat), labellings and scores bit-equal to the round-3 kernel.
// Every score is computed by the same expressions on the same operands as in beam_kernel: bit-identical results.  (Against the
// C oracle: identical labellings; float64 scores bit-equal on the golden sets and within 2 ulp elsewhere -- ocml's exp / log vs glibc's.)
// =====================================================================================================================
constexpr int FAST_WMAX = 64;

struct FastArgs {
  const double *lgd; const float *pb; const unsigned char *zf;
  const int32_t *lens; const double *lm; double alpha; int W, blank;
  int32_t *out_ids, *out_len; double *out_score; int32_t *status; int T, B, V;
  int nbest; int32_t *out_count;
  unsigned long long *ht; int *node_par; int *node_sym; int ht_size, max_nodes;
  int trie_slots;       // LDS trie slots (power of two)
#ifdef CTCN_BEAM_STATS
  long long *stats;     // development instrumentation (tools/mb_beam.py): cycles per phase of workgroup 0
#endif
};
#ifdef CTCN_BEAM_STATS
#define BSTAMP(i) do { const long long now_ = clock64(); zst[i] += now_ - zlast; zlast = now_; } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void beam_prep_kernel(const float *__restrict__ x, int input_is_prob, double *__restrict__ lgd,
                                                        float *__restrict__ pb, unsigned char *__restrict__ zf, size_t rows, int V, int blank) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + row * V;
  bool bad = false;
  for (int k = lane; k < V; k += 64) {
    const float p = input_is_prob ? xr[k] : expf(xr[k]);
    if (!(p > 0.0f)) bad = true;
    lgd[row * V + k] = log((double)p);                 // math.log of the float32 value widened to double
    if (k == blank) pb[row] = p;
  }
  const bool any_bad = __any(bad);
  if (lane == 0) zf[row] = any_bad ? 1 : 0;
}

// order-preserving map double -> uint64 (total order of the IEEE values; -0.0 / NaN do not occur among the scores)
__device__ __forceinline__ unsigned long long f64_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax_step(unsigned v) {
  const unsigned s = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
  return max(v, s);
}
// max over the 64 lanes on the DPP network: row_shr 1/2/4/8 leave each row's maximum in its lane 15, row_bcast15 / row_bcast31
// carry it across the rows; lane 63 holds the result (max is idempotent: lanes without a source keep their own value)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = dpp_umax_step<0x111, 0xf>(v);   // row_shr:1
  v = dpp_umax_step<0x112, 0xf>(v);   // row_shr:2
  v = dpp_umax_step<0x114, 0xf>(v);   // row_shr:4
  v = dpp_umax_step<0x118, 0xf>(v);   // row_shr:8
  v = dpp_umax_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
  v = dpp_umax_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umin_step(unsigned v) {
  const unsigned s = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
  return min(v, s);
}
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
  v = dpp_umin_step<0x111, 0xf>(v);
  v = dpp_umin_step<0x112, 0xf>(v);
  v = dpp_umin_step<0x114, 0xf>(v);
  v = dpp_umin_step<0x118, 0xf>(v);
  v = dpp_umin_step<0x142, 0xa>(v);
  v = dpp_umin_step<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int lane_gather(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ double lane_gather(double v, int src_lane) {
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v)));
}

struct __attribute__((aligned(16))) Survivor { unsigned long long k; int idx; int pad; };
constexpr int SURV_MAX = 256;

constexpr int FAST_NCT = 1024 - 192;                           // candidate threads: waves 3..15
constexpr int FAST_NTH = 1024, FAST_NWV = FAST_NTH / 64;      // 16 waves: the parallel phases are instruction-issue bound (~10 cycles per
                                                               // dependent instruction and wave), so more waves per SIMD is what shortens them
template <int NPT, bool LM_LDS>
__device__ __forceinline__ void beam_fast_body(FastArgs a) {
  constexpr int NTH = FAST_NTH, NWV = FAST_NWV;
  // dynamic LDS: [alpha*LM (V+1)^2 doubles] | cand[W*V] doubles | lg[2][V] doubles (by frame parity) | mslot[W*V] ints | flist[T] ints | trie[slots] uints
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  __shared__ Survivor surv[SURV_MAX];                                 // candidates above the pruning bound (order-preserving key, index)
  __shared__ double red_v[NWV];
  __shared__ int red_i[NWV + 1];
  __shared__ unsigned gmax[64];                                        // per 16-lane row: largest candidate key (high word)
  __shared__ double bm_pB[FAST_WMAX], bm_pT[FAST_WMAX], selv[FAST_WMAX];
  __shared__ int selp[FAST_WMAX];                                     // per selected candidate: its position in surv[] (the speculative stay totals are stored by it)
  __shared__ double tots[64];                                         // per survivor: log_add(prBlank', prNonBlank') of the next frame if it is selected (wave 2)
  __shared__ int selm[FAST_WMAX];                                     // per selected candidate: 128 | slot whose merged entry it holds, 0: none, -1: see mslot[]
  __shared__ int bm_c1[2][FAST_WMAX], sel[FAST_WMAX];                 // context class of every slot, by frame parity
  __shared__ int f_node[FAST_WMAX], f_len[FAST_WMAX], f_last[FAST_WMAX];   // final beam (dumped once, after the last frame)
  __shared__ double f_pT[FAST_WMAX];
  __shared__ int woff[NWV];
  __shared__ double stayv[FAST_WMAX], homev[FAST_WMAX];
  __shared__ int ns[FAST_WMAX];
  __shared__ double enbv[FAST_WMAX], totv[FAST_WMAX];                // this frame's e.nb of every slot (e.t is homev) | log_add(prBlank', prNonBlank') of the next frame (wave 2)
  __shared__ int nid[2][FAST_WMAX];                                   // node id of every beam slot, double-buffered by frame parity (wave 1)
  __shared__ int s_totflag, s_beamflag, s_decflag;                    // frame number + 1 whose totv[] | beam record (bm_*) | wave 1's decode of the selection is complete
  __shared__ int s_fault;                                             // a bounded wait inside the workgroup ran out (status 4)
  __shared__ int s_cnt;
  __shared__ unsigned s_theta;
  __shared__ int s_nfl, s_gnodes;                // processed frames | 0 while the LDS trie takes inserts, else 1 + global nodes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, V = a.V, W = a.W, B = a.B, T = a.T, blank = a.blank;
  const int V1 = V + 1;
  double *lmA = reinterpret_cast<double *>(fsm);
  double *cand = lmA + (LM_LDS ? V1 * V1 : 0);
  double *lg2 = cand + W * V;
  int *mslot = reinterpret_cast<int *>(lg2 + 2 * V);
  int *flist = mslot + W * V;
  unsigned *trie = reinterpret_cast<unsigned *>(flist + T);
  const int TS = a.trie_slots;
  const unsigned tmask = (unsigned)TS - 1u;
  const int tshift = 32 - (31 - __builtin_clz((unsigned)TS));      // top log2(TS) bits of a 32-bit hash
  unsigned long long *ht = a.ht + (size_t)b * a.ht_size;
  int *npar = a.node_par + (size_t)b * a.max_nodes;
  int *nsym = a.node_sym + (size_t)b * a.max_nodes;
  const unsigned htmask = (unsigned)a.ht_size - 1u;
  constexpr int TMASK = (1 << 29) - 1;

  if (LM_LDS)
    for (int i = tid; i < V1 * V1; i += NTH) lmA[i] = a.lm[i] * a.alpha;        // the same product the generic kernel forms per use
  for (int i = tid; i < W * V; i += NTH) mslot[i] = -1;
  for (int i = tid; i < TS; i += NTH) trie[i] = 0u;
  const int nframes = min(max(a.lens[b], 0), T);
  if (tid == 0) {
    bm_c1[0][0] = V; bm_pB[0] = 0.0; bm_pT[0] = 0.0;                             // the empty labelling: prBlank = prTotal = 0 (BeamSearch.py:83-87)
    s_nfl = 0; s_gnodes = 0; s_totflag = 0; s_beamflag = 0; s_decflag = 0; s_fault = 0;
  }
  if (tid < 64) gmax[tid] = 0u;                                                  // (rows 52..63 belong to no candidate wave: they stay 0)
  if (tid < 2 * FAST_WMAX) nid[0][tid] = 0;                                       // (both parities: node 0 = the empty labelling)
  __syncthreads();
  // frames the search processes, in order (BeamSearch.py:93-94: skip when 1 - p_blank < 0.1, a float32 compare), each with its
  // "p_blank of the previous frame < 0.9" bit (:63) and its "log(0)" bit
  for (int t0 = 0; t0 < nframes; t0 += NTH) {
    const int t = t0 + tid;
    bool keep = false;
    int word = 0;
    if (t < nframes) {
      const float pbl = a.pb[(size_t)t * B + b];
      keep = !((1.0f - pbl) < 0.1f);
      const bool rep = t > 0 && a.pb[(size_t)(t - 1) * B + b] < 0.9f;
      word = t | (rep ? 1 << 30 : 0) | (a.zf[(size_t)t * B + b] ? 1 << 29 : 0);
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) woff[wave] = __popcll(mask);
    __syncthreads();
    int base = s_nfl;
    for (int w = 0; w < wave; ++w) base += woff[w];
    if (keep) flist[base + __popcll(mask & ((1ull << lane) - 1ull))] = word;
    __syncthreads();
    if (tid == 0) { int add = 0; for (int w = 0; w < NWV; ++w) add += woff[w]; s_nfl += add; }
    __syncthreads();
  }
  const int nfl = s_nfl;

  // Wave 0 owns the beam (lane = slot) and the serial chain, wave 1 the prefix trie (node ids are only needed a frame later), wave 2 the
  // log-add of every slot's stay entry that needs no parent slot (round 4); waves 3..15 (832 threads) own the candidates.
  // That log-add -- exp and log in f64, ~1.8 k cycles of one wave -- is the longest link of the frame.  Wave 2 forms it for every SURVIVOR of
  // the pruning bound (at most 64, one per lane) while the candidate waves still rank them: up to 1 + exp(.) before the barrier that ends the
  // rank count, the logarithm behind it; wave 0 picks the totals of the selected survivors up by their position (selp[] -> tots[]).  With
  // more than 64 survivors the total is formed behind the decode of the selection as before (totv[]).  (Measured and not kept: a counter
  // instead of that barrier so that wave 2 need not hold it up -- thirteen LDS atomics and two polling waves cost more than the barrier;
  // the logarithm's reciprocal / quotient stage before the barrier as well: the barrier is late, +3 %; a logarithm with the parts that are
  // constant on [1, 2] folded -- bit-equal to the library's on 2^24 values -- behind the barrier: no faster, 912 vs 908 us.)
  // Every role runs ITS OWN frame loop below (same barriers, in the same order): the kernel sits at its register limit, and with the roles
  // interleaved phase by phase in one loop every role's state was live everywhere -- the allocator spilled loop-invariant addresses to
  // scratch and re-loaded them inside the frame's critical path.  In separate branches the live ranges do not overlap.
  constexpr int NCT = FAST_NCT;
  // ln p row of the next frame, one frame ahead, held by threads 128 .. 128 + V - 1 (waves 2..5: wave 0's instruction stream is the
  // critical chain); lg[p] = row of the frames of parity p, written at the top of the frame BEFORE the one that uses it, so that every
  // reader finds it behind the selection's barriers
  double nlg = 0.0;
  const int lgk = tid - 128;
  auto fetch_lg = [&](int fword) {
    if (lgk >= 0 && lgk < V) nlg = a.lgd[((size_t)(fword & TMASK) * B + b) * V + lgk];
  };
  if (nfl > 0) fetch_lg(flist[0]);
  if (lgk >= 0 && lgk < V) lg2[lgk] = nlg;
  if (nfl > 1) fetch_lg(flist[1]);
  auto frame_top = [&](int j) {                                  // (holders only) the next frame's ln p row: its last readers were frame j - 1's
    if (lgk >= 0 && lgk < V && j + 1 < nfl) lg2[((j + 1) & 1) * V + lgk] = nlg;
    if (j + 2 < nfl) fetch_lg(flist[j + 2]);
  };
  int nb = 1, status = 0;
  // the serial chain (wave 0) and its two helpers win the issue arbitration of their SIMDs against the scoring waves they share them with (0.3-0.9 %)
  if (wave == 0) __builtin_amdgcn_s_setprio(3); else if (wave == 2) __builtin_amdgcn_s_setprio(2); else if (wave == 1) __builtin_amdgcn_s_setprio(1);
  // (round 4, measured on one box against the interleaved build -- cfg5 batch 1 098 us peaky / 2 796 us flat -- and NOT kept: the pruning bound
  // formed by the scoring waves in the shadow of wave 0's chain, with a barrier of their own or ranked by the last wave to arrive: 1 167-1 230 /
  // 2 804-2 903 (13 waves x the extra instructions are issue-bound and end up BEHIND wave 0); candidate state packed into one word + opaque
  // indices so that nothing spills: 1 158 / 2 929 (the unpacking costs the parallel phases more than the scratch re-loads did); the bound ranked
  // by waves 0..3 only: 1 192 / 2 987.  The parallel phases cost (instructions per wave) x (waves per SIMD) x ~4.5 cycles whatever their
  // dependences are.  The bound in the scoring phase again ON this role-split build (995 / 2 552): 991 / 2 402, with wave 2 publishing the beam
  // record early 998 / 2 410, with a second bound from the stay totals 979 / 2 434 -- the scoring waves' own barrier waits for the slowest of
  // thirteen issue-bound waves, 1.6 k cycles in the very phase the bound was meant to hide in; not kept for 5 % of the flat regime.)
  __syncthreads();
#ifdef CTCN_BEAM_STATS
  long long zst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, zlast = clock64(), zrounds = 0, ziters = 0;
  const long long zt0 = zlast;
#endif
  // more than SURV_MAX candidates above the bound (never seen on the synthetic regimes): W block-wide arg-max rounds over the candidate
  // table (wave 0 patches it first so that it holds this frame's stay / merged entries), exactly as the generic kernel does.  All threads.
  auto arg_max_rounds = [&]() -> int {
    const int ncand = nb * V;
    int got = 0;
    for (int r = 0; r < W; ++r) {
      double bv = -INFINITY; int bi = 0x7fffffff;
      for (int c = tid; c < ncand; c += NTH) {
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
      lds_barrier();
      if (tid == 0) {
        double v = red_v[0]; int ix = red_i[0];
        for (int w = 1; w < NWV; ++w)
          if (red_i[w] != 0x7fffffff && (ix == 0x7fffffff || cand_better(red_v[w], red_i[w], v, ix))) { v = red_v[w]; ix = red_i[w]; }
        red_i[NWV] = ix;
        if (ix != 0x7fffffff) { sel[r] = ix; selv[r] = v; selm[r] = -1; cand[ix] = -INFINITY; }
      }
      lds_barrier();
      if (red_i[NWV] == 0x7fffffff) break;
      ++got;
    }
    return got;
  };
  // waves 0..2: decode of the selection -- new slot `lane` <- candidate sel[lane]: which old slot it comes from, fresh labelling or copy
  struct Dec { bool act, fresh; int src, sym; double sv; };
  auto decode_sel = [&](int j, int m) -> Dec {
    Dec d;
    const int rr = min(lane, FAST_WMAX - 1);
    const int c = sel[rr];
    d.sv = selv[rr];
    d.act = lane < m;
    const int i = d.act ? (int)(((float)c + 0.5f) * (1.0f / (float)V)) : 0;       // c / V (exact for c < 2^20)
    const int kk = c - i * V;
    d.sym = (kk - 1 < blank) ? kk - 1 : kk;
    // does this candidate hold the merged entry of a slot?  The candidate's thread looked that up when it loaded the value (selm[]: one LDS
    // round trip less on the serial chain); -1: the selection came from the arg-max rounds, mslot[] has it
    int sm = selm[rr];
    if (sm < 0) { const int ms = mslot[d.act ? c : 0]; sm = (kk != 0 && ms >= 0 && (ms >> 8) == j && (ms & 128)) ? (128 | (ms & 63)) : 0; }
    const bool merged = d.act && (sm & 128);
    d.fresh = d.act && kk != 0 && !merged;
    d.src = merged ? (sm & 63) : i;
    return d;
  };
  const bool first_ok = nfl > 0 && !(flist[0] & (1 << 29));
  const bool rep0 = nfl > 0 && ((flist[0] >> 30) & 1);

  if (wave >= 3) {
    // ======================================= waves 3..15: the candidates =======================================================
    // slot i of thread u = tid - 192 is c = ((37 * u) mod 832) + 832 * i -> (beam ci, class ck; ck < 0: the stay slot).  Any bijection works
    // (the selection ranks by explicit (score, index)); the multiplier spreads neighbouring candidates -- the classes of one beam -- over
    // different 16-lane rows, which keeps the pruning bound of the selection tight.
    const int u = tid - 192;
    int cc[NPT], ci[NPT], ck[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int c = ((37 * u) % NCT) + i * NCT;
      cc[i] = min(c, W * V - 1);
      ci[i] = c < W * V ? c / V : FAST_WMAX;                      // beyond the table: never valid (nb <= W <= FAST_WMAX)
      const int kk = c - (c / V) * V;
      ck[i] = kk == 0 ? -1 : ((kk - 1 < blank) ? kk - 1 : kk);
    }
    // extension scores (calcExtPr) of frame jf into cand[]: two LDS hops (the slot's context class, then LM / prBlank / prTotal), every
    // read of a hop issued before the first use (clamped addresses, selects afterwards)
    auto score_extensions = [&](bool rep_ok, int jf) {
      const double *lg = lg2 + (jf & 1) * V;
      int c1[NPT];
#pragma unroll
      for (int i = 0; i < NPT; ++i) c1[i] = bm_c1[jf & 1][min(ci[i], FAST_WMAX - 1)];
      double lmv[NPT], pbv[NPT], ptv[NPT], lk[NPT];
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int bi = min(ci[i], FAST_WMAX - 1), k = max(ck[i], 0);
        const int c1c = min(max(c1[i], 0), V);
        lmv[i] = LM_LDS ? lmA[c1c * V1 + k] : a.lm[(size_t)c1c * V1 + k] * a.alpha;
        lk[i] = lg[k];
        pbv[i] = bm_pB[bi];
        ptv[i] = bm_pT[bi];
      }
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        if (ci[i] < nb && ck[i] >= 0) {
          const double base = (c1[i] == ck[i] && rep_ok) ? pbv[i] : ptv[i];     // c1 == k <=> non-empty labelling ending in k
          cand[cc[i]] = lk[i] + lmv[i] + base;
        }
      }
    };
    if (first_ok) score_extensions(rep0, 0);
    lds_barrier();
    for (int j = 0, fw = nfl > 0 ? flist[0] : 0; j < nfl; ++j) {  // (fw: the frame's word, read one frame ahead)
      if (fw & (1 << 29)) { status = 2; break; }                   // math.log(0) in the reference: ValueError
      frame_top(j);
      BSTAMP(0);
      // P3: BHat = top-W by (prTotal desc, candidate index asc), without sorting.
      //  1. splitter: every 16-lane row reduces the largest key (high word) of its 16 * NPT candidates on the DPP network; the W-th largest
      //     of the 52 row maxima is a lower bound of the W-th best candidate (W distinct candidates reach it): exact pruning;
      //  2. the survivors (typically W + a few) are compacted into LDS and ranked by counting, the candidate threads sharing the compares.
      unsigned long long key[NPT];
      bool val[NPT];
      unsigned mhi = 0u, mpack = 0u;                                // mpack: byte i = 128 | slot when candidate i holds that slot's merged entry
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        const int bi = min(ci[i], FAST_WMAX - 1);
        double v = ck[i] < 0 ? stayv[bi] : cand[cc[i]];
        const int ms = mslot[cc[i]];
        if (ck[i] >= 0 && ms >= 0 && (ms >> 8) == j) {
          v = (ms & 128) ? homev[ms & 63] : -INFINITY;
          if (ms & 128) mpack |= (unsigned)(128 | (ms & 63)) << (8 * i);
        }
        key[i] = f64_key(v);
        val[i] = ci[i] < nb && v != -INFINITY;
        mhi = max(mhi, val[i] ? (unsigned)(key[i] >> 32) : 0u);
      }
      mhi = dpp_umax_step<0x111, 0xf>(mhi);
      mhi = dpp_umax_step<0x112, 0xf>(mhi);
      mhi = dpp_umax_step<0x114, 0xf>(mhi);
      mhi = dpp_umax_step<0x118, 0xf>(mhi);
      if ((lane & 15) == 15) gmax[u >> 4] = mhi;
      BSTAMP(8);
      lds_barrier();
      {
        // rank of row maximum e among the 52 (ties broken by the row number: a strict order); 16 threads per row maximum (gmax[52..63] stay 0)
        const int e = u >> 4, part = u & 15;
        const unsigned mine = gmax[e];
        int cnt = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int q = part * 4 + q4;
          const unsigned o = gmax[q];
          cnt += (o > mine || (o == mine && q < e)) ? 1 : 0;
        }
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);    // row_half_mirror
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x140, 0xf, 0xf, true);    // row_mirror
        if (part == 0 && cnt == W - 1) s_theta = mine;                        // (no such row when fewer than W rows hold a candidate: 0)
      }
      lds_barrier();
      {
        const unsigned theta = s_theta;
        bool keep[NPT];
        int wtot = 0, pos[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          keep[i] = val[i] && (unsigned)(key[i] >> 32) >= theta;
          const unsigned long long bal = __ballot(keep[i]);
          pos[i] = wtot + __popcll(bal & ((1ull << lane) - 1ull));
          wtot += __popcll(bal);
        }
        int wbase = 0;
        if (lane == 0 && wtot > 0) wbase = atomicAdd(&s_cnt, wtot);
        wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
        for (int i = 0; i < NPT; ++i)
          if (keep[i] && wbase + pos[i] < SURV_MAX) {
            Survivor sv;
            sv.k = key[i]; sv.idx = ci[i] * V + (ck[i] < 0 ? 0 : (ck[i] < blank ? ck[i] + 1 : ck[i])); sv.pad = (int)((mpack >> (8 * i)) & 255u);
            surv[wbase + pos[i]] = sv;
          }
      }
      BSTAMP(3);
      lds_barrier();
      // (the reads of the common case -- up to 52 survivors, 16 threads each -- are issued before the survivor count is known: one LDS round trip
      // instead of three in a row)
      // (thread index of the rank count: the candidate waves that share wave 2's SIMD -- 6, 10, 14 -- take the last survivors, which rarely
      // exist, so that wave 2's log-add has that SIMD to itself)
      const int cw = wave - 3;
      const int ur = (((cw & 3) == 3) ? 10 + (cw >> 2) : cw - (cw >> 2)) * 64 + lane;
      const Survivor me16 = surv[ur >> 4];
      Survivor oe16[4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) oe16[q4] = surv[(ur & 15) + 16 * q4];
      const int S = s_cnt;
      int total = S;
      if (S <= 52) {
        if ((ur & ~63) < S * 16) {
          const int e = ur >> 4, part = ur & 15;
          int cnt = 0;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            cnt += (part + 16 * q4 < S && (oe16[q4].k > me16.k || (oe16[q4].k == me16.k && oe16[q4].idx < me16.idx))) ? 1 : 0;
          cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);                  // quad_perm [1,0,3,2]
          cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);                  // quad_perm [2,3,0,1]
          cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);                 // row_half_mirror
          cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x140, 0xf, 0xf, true);                 // row_mirror
          if (e < S && part == 0 && cnt < W) { sel[cnt] = me16.idx; selv[cnt] = key_f64(me16.k); selm[cnt] = me16.pad; selp[cnt] = e; }
        }
      } else if (S <= SURV_MAX) {
        // rank counting: P of the 832 candidate threads per survivor, thread part p compares it with survivors p, p + P, ...; DPP butterfly sum
        const int P = S <= 104 ? 8 : (S <= 208 ? 4 : 2);
        if ((ur & ~63) < S * P) {                                 // (waves whose 64 / P survivors do not exist go straight to the barrier: the
                                                                 // phase is issue-bound, fewer waves per SIMD finish it sooner)
          const int e = ur / P, part = ur - e * P;
          const Survivor me = surv[min(e, SURV_MAX - 1)];
          int cnt = 0;
          for (int q0 = part; q0 < S; q0 += 4 * P) {
            Survivor oe[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) oe[q4] = surv[min(q0 + q4 * P, SURV_MAX - 1)];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              cnt += (q0 + q4 * P < S && (oe[q4].k > me.k || (oe[q4].k == me.k && oe[q4].idx < me.idx))) ? 1 : 0;
          }
          cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);                  // quad_perm [1,0,3,2]
          if (P >= 4) cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
          if (P >= 8) cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);     // row_half_mirror
          if (e < S && part == 0 && cnt < W) { sel[cnt] = me.idx; selv[cnt] = key_f64(me.k); selm[cnt] = me.pad; selp[cnt] = e; }
        }
      } else {
        lds_barrier();                                             // (wave 0 has patched the candidate table)
        total = arg_max_rounds();
      }
      lds_barrier();
      BSTAMP(4);
      nb = min(W, total);
      const int fwn = flist[min(j + 1, nfl - 1)];
      const bool more = j + 1 < nfl && !(fwn & (1 << 29));
      if (more) {
        // the new beam's record (context class, prBlank, prTotal of every slot) comes from wave 0 through a flag, not a barrier
        for (int spins = 0; __hip_atomic_load(&s_beamflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j + 1; ++spins) {
          if (spins > (1 << 20)) { s_fault = 1; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        BSTAMP(6);
        score_extensions((fwn >> 30) & 1, j + 1);
        BSTAMP(9);
      }
      fw = fwn;
      lds_barrier();
      BSTAMP(2);
    }
  } else if (wave == 0) {
    // ======================================= wave 0: the beam and the serial chain =============================================
    // lane r holds slot r.  node ids: 0 = the empty labelling, s + 1 = LDS trie slot s, TS + 1 + g = entry g of the global table.
    // z_mf = the slot that holds this slot's parent labelling (-1: none) -- its extension by z_last IS this labelling.
    // Wave 0 never waits for the trie: the id of a slot's OWN labelling lives with wave 1 (handed over through nid[] a frame later, when
    // a child needs it as its z_par).  What wave 0 carries instead is the parent labelling's key, (z_gpar, z_plast) = (id of the
    // grandparent labelling, last class of the parent labelling): a labelling created in this frame as (parent id p, class k) IS the
    // parent of slot r exactly when p == z_gpar[r] and k == z_plast[r] (ids are unique per (parent, class)), which needs old ids only.
    int z_len = 0, z_last = -1, z_par = -1, z_mf = -1, z_gpar = -2, z_plast = -1;
    double z_pB = 0.0, z_pNB = LOG_ZERO, z_pT = 0.0;
    double e_nb = LOG_ZERO, e_b = LOG_ZERO, e_t = LOG_ZERO;        // this frame's stay / merged entry of the slot (BeamSearch.py:99-113)
    // stay entries of every slot, merged with the extension of the parent slot that equals the labelling (same expression, same operands
    // as score_extensions writes for that candidate).  Results: e_nb / e_b / e_t in the slot's lane; for the selection: stayv[ip] = value
    // of the stay candidate (or -inf when the entry lives in the extension's place), homev[ip] = e_t, and mslot[extension candidate] =
    // {frame tag, 1: holds slot ip's merged entry | 0: removed (merged into the stay candidate), ip}
    auto stay_and_merge = [&](bool rep_ok, int jf, int tsrc) {
      const int ip = lane;
      const double *lg = lg2 + (jf & 1) * V;
      const double lgl = lg[max(z_last, 0)], lgb = lg[blank];
      const int mi = ip < nb ? z_mf : -1;
      const int src = max(mi, 0);
      const double p_pB = lane_gather(z_pB, src), p_pT = lane_gather(z_pT, src);
      // (context class of the parent labelling: its last class travels with the slot -- z_plast -- and its length is this one's minus one:
      // the LM term is read next to the ln p terms, not behind the parent slot)
      const int c1 = z_len > 1 ? max(z_plast, 0) : V, k = max(z_last, 0);
      const double lmv = LM_LDS ? lmA[c1 * V1 + k] : a.lm[(size_t)c1 * V1 + k] * a.alpha;
      const double base = (c1 == k && rep_ok) ? p_pB : p_pT;
      const double pr = lgl + lmv + base;                           // == cand[mi * V + kk(z_last)]
      double s_nb = LOG_ZERO;
      if (z_len > 0) s_nb = z_pNB + lgl;                            // BeamSearch.py:102-103
      const double s_b = z_pT + lgb;                                // :106
      const bool ext_first = mi >= 0 && mi < ip;                    // the reference meets the extension before the stay entry
      BSTAMP(6);
      // tot = log_add(s_b, s_nb) of every slot comes from wave 2 (it needs no parent slot, so it is computed next to the lines above)
      // (bounded: wave 2 reaches its store whenever this wave gets here -- both follow the wave-uniform `more` -- so the bound only turns a
      // programming error into status 4 instead of a hung workgroup)
      // -- and mslot[] below is rewritten only once wave 1, too, has decoded this frame's selection (s_decflag; set thousands of cycles ago)
      for (int spins = 0; __hip_atomic_load(&s_totflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != jf + 1 ||
                          __hip_atomic_load(&s_decflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != jf; ++spins) {
        if (spins > (1 << 20)) { s_fault = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      const double tot = tsrc >= 0 ? tots[tsrc] : totv[ip];
      double r_nb = s_nb, r_t = tot;
      if (__any(mi >= 0)) {                                          // some slot merges with its parent's extension: ONE level of log-adds
        // lanes 0..31: e.t of slot q, lanes 32..63: e.nb of slot q, q = (lane & 31) + 32 * pass -- side by side (one pass when nb <= 32)
#pragma nounroll
        for (int q0 = 0; q0 < nb; q0 += 32) {
          const int q = (lane & 31) + q0;
          const double q_snb = lane_gather(s_nb, q), q_pr = lane_gather(pr, q), q_tot = lane_gather(tot, q);
          const int q_mi = lane_gather(mi, q);
          const bool q_first = q_mi >= 0 && q_mi < q;
          const double other = lane < 32 ? q_tot : q_snb;
          const double ax = q_first ? q_pr : other, ay = q_first ? other : q_pr;      // the reference's argument order (extension met first or second)
          const double r1 = q_mi >= 0 ? log_add_prob(ax, ay) : LOG_ZERO;
          const int from = (lane & 31);                              // slot ip = q0 + from reads lanes from (e.t) and from + 32 (e.nb)
          const double a_t = lane_gather(r1, from), a_nb = lane_gather(r1, from + 32);
          if (mi >= 0 && (ip >> 5) == (q0 >> 5)) { r_nb = a_nb; r_t = a_t; }
        }
      }
      BSTAMP(7);
      if (ip < nb) {
        e_nb = r_nb; e_b = s_b; e_t = r_t;
        homev[ip] = r_t; enbv[ip] = r_nb;
        stayv[ip] = ext_first ? -INFINITY : r_t;
        if (mi >= 0) {
          const int kkl = (z_last < blank) ? z_last + 1 : z_last;
          mslot[mi * V + kkl] = (jf << 8) | (ext_first ? 128 : 0) | ip;
        }
      }
      if (lane == 0) { s_theta = 0u; s_cnt = 0; }
    };
    if (first_ok) stay_and_merge(rep0, 0, -1);
    lds_barrier();
    for (int j = 0, fw = nfl > 0 ? flist[0] : 0; j < nfl; ++j) {
      if (fw & (1 << 29)) { status = 2; break; }
      BSTAMP(0);
      lds_barrier();                                               // (row maxima)
      lds_barrier();                                               // (pruning bound)
      BSTAMP(3);
      lds_barrier();                                               // (survivors compacted)
      const int S = s_cnt;
      int total = S;
      if (S > SURV_MAX) {
        if (lane < nb) {
          cand[lane * V] = stayv[lane];
          if (z_mf >= 0) { const int kkl = (z_last < blank) ? z_last + 1 : z_last; cand[z_mf * V + kkl] = (z_mf < lane) ? homev[lane] : -INFINITY; }
        }
        lds_barrier();
        total = arg_max_rounds();
      }
      lds_barrier();                                               // (selection ranked)
      BSTAMP(4);
      const int m = min(W, total);
      const int fwn = flist[min(j + 1, nfl - 1)];
      const bool more = j + 1 < nfl && !(fwn & (1 << 29));
      const bool rep_next = (fwn >> 30) & 1;
      // P4a: the new beam in rank order; the record the next frame's extension scores need goes out through s_beamflag (wave 0 does not stop)
      const Dec d = decode_sel(j, m);
      const int tsrc = (more && S <= 64) ? (selp[min(lane, FAST_WMAX - 1)] & 63) : -1;   // where wave 2 leaves this slot's stay total (tots[]; -1: totv[])
      const bool act = d.act, fresh = d.fresh;
      const int src = d.src, sym = d.sym;
      ns[lane] = -1;
      const int g_len = lane_gather(z_len, src), g_last = lane_gather(z_last, src), g_par = lane_gather(z_par, src);
      const int g_gpar = lane_gather(z_gpar, src), g_plast = lane_gather(z_plast, src);
      const int g_nid = nid[j & 1][src];                           // id of the old slot's labelling (wave 1, previous frame)
      const int g_mf = lane_gather(z_mf, src);
      const double g_nb = lane_gather(e_nb, src), g_b = lane_gather(e_b, src), g_t = lane_gather(e_t, src);
      int n_len = g_len, n_last = g_last, n_par = g_par, n_gpar = g_gpar, n_plast = g_plast, n_mf = -1;
      double n_pNB = g_nb, n_pB = g_b, n_pT = g_t;
      if (fresh) {
        n_len = g_len + 1; n_last = sym; n_par = g_nid; n_gpar = g_len > 0 ? g_par : -2; n_plast = g_last;
        n_pNB = d.sv; n_pB = LOG_ZERO; n_pT = d.sv;
      }
      if (act) { bm_c1[(j + 1) & 1][lane] = n_len > 0 ? n_last : V; bm_pB[lane] = n_pB; bm_pT[lane] = n_pT; }
      if (act && !fresh) ns[src] = lane;                           // where the old slot's labelling went
      __hip_atomic_store(&s_beamflag, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      BSTAMP(5);
      nb = m;
      // P4b: parent slot of every new slot, then the stay / merge entries of the NEXT frame.  A fresh labelling's parent is the old slot it
      // extends; a copy's parent is the old slot's parent, wherever that went -- or, when the parent was NOT in the old beam, possibly a
      // labelling created just now: fresh lane f holds labelling (parent id n_par[f], class n_last[f]), which is slot r's parent exactly
      // when that pair equals (n_gpar[r], n_plast[r]).  The loop walks the shorter of the two lane sets.
      {
        const int pm = fresh ? src : g_mf;
        n_mf = (act && pm >= 0 && n_len > 0) ? ns[pm] : -1;
        const bool scan = act && !fresh && n_len > 1 && g_mf < 0;   // (a labelling of length 1 has the empty labelling as parent: never fresh)
        unsigned long long fmask = __ballot(fresh), smask = __ballot(scan);
        if (smask != 0ull && fmask != 0ull) {
          if (__popcll(fmask) <= __popcll(smask)) {
            while (fmask) {
              const int fl = __ffsll((long long)fmask) - 1;
              fmask &= fmask - 1;
              const int fp = __builtin_amdgcn_readlane(n_par, fl), fk = __builtin_amdgcn_readlane(n_last, fl);
              if (scan && fp == n_gpar && fk == n_plast) n_mf = fl;
            }
          } else {
            while (smask) {
              const int sl = __ffsll((long long)smask) - 1;
              smask &= smask - 1;
              const int gp = __builtin_amdgcn_readlane(n_gpar, sl), pk = __builtin_amdgcn_readlane(n_plast, sl);
              const unsigned long long hit = __ballot(fresh && n_par == gp && n_last == pk);
              if (hit != 0ull && lane == sl) n_mf = __ffsll((long long)hit) - 1;
            }
          }
        }
      }
      z_len = n_len; z_last = n_last; z_par = n_par; z_gpar = n_gpar; z_plast = n_plast; z_mf = n_mf; z_pNB = n_pNB; z_pB = n_pB; z_pT = n_pT;
      if (more) stay_and_merge(rep_next, j + 1, tsrc);
      fw = fwn;
      lds_barrier();
      BSTAMP(2);
    }
    f_len[lane] = z_len; f_last[lane] = z_last; f_pT[lane] = z_pT;
  } else {
    // ======================================= waves 1 and 2: the trie | the stay totals ==========================================
    int y_node = 0, y_lnodes = 0;                                  // wave 1: node id of slot `lane`; LDS trie nodes so far (wave-uniform)
    // wave 2: tot = log_add(prBlank', prNonBlank') (BeamSearch.py:112) of the stay entry of every slot for frame jf, from the slot's (context
    // class, prNonBlank, prTotal) -- the same expressions on the same operands as wave 0 forms for s_nb / s_b -- then the flag wave 0 waits
    // for.  It needs no parent slot, so it runs next to wave 0's new-beam / parent-slot work instead of in front of its log-add
    auto stay_total = [&](int jf, bool on, int c1, double pNB, double pT) -> double {
      const double *lg = lg2 + (jf & 1) * V;
      const double lgl = lg[c1 < V ? c1 : 0], lgb = lg[blank];
      double s_nb = LOG_ZERO;
      if (c1 < V) s_nb = pNB + lgl;
      const double s_b = pT + lgb;
      return on ? log_add_prob(s_b, s_nb) : LOG_ZERO;
    };
    auto stay_totals = [&](int jf, bool on, int c1, double pNB, double pT) {
      totv[lane] = stay_total(jf, on, c1, pNB, pT);
      __hip_atomic_store(&s_totflag, jf + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (first_ok && wave == 2) stay_totals(0, lane == 0, V, LOG_ZERO, 0.0);       // the empty labelling
    lds_barrier();
    for (int j = 0, fw = nfl > 0 ? flist[0] : 0; j < nfl; ++j) {
      if (fw & (1 << 29)) { status = 2; break; }
      frame_top(j);
      BSTAMP(0);
      lds_barrier();
      lds_barrier();
      lds_barrier();
      const int S = s_cnt;
      int total = S;
      const int fwn = flist[min(j + 1, nfl - 1)];
      const bool more = j + 1 < nfl && !(fwn & (1 << 29));
      // wave 2, while the candidate waves rank the survivors: the next frame's stay total of EVERY survivor (at most 64: one per lane), from
      // what its new slot would carry -- a fresh labelling its candidate's score, a copy the stay / merged entry of the slot it comes from --
      // so that the log-add is over when wave 0 asks for it, instead of starting behind the decode of the selection
      const bool spec = wave == 2 && more && S <= 64;
      double sp_x = LOG_ZERO, sp_s = 1.0;
      bool sp_half = false;
      if (spec) {
        const Survivor sv = surv[lane];
        const bool on = lane < S;
        const int c = on ? sv.idx : 0;
        const int i = (int)(((float)c + 0.5f) * (1.0f / (float)V));
        const int kk = c - i * V;
        const int sym = (kk - 1 < blank) ? kk - 1 : kk;
        const bool merged = on && (sv.pad & 128);
        const bool fresh = on && kk != 0 && !merged;
        const int src = merged ? (sv.pad & 63) : i;
        const double v = key_f64(sv.k);
        const int o_c1 = bm_c1[j & 1][src];
        const double o_nb = enbv[src], o_t = homev[src];
        // log_add_prob(prBlank', prNonBlank') in two halves: up to 1 + exp(.) before the barrier the candidate waves' rank count ends with, the
        // logarithm behind it (the whole chain, ~1.8 k cycles, would hold that barrier up; the store keeps the first half on this side of it)
        const int c1 = fresh ? sym : o_c1;
        const double pNB = fresh ? v : o_nb, pT = fresh ? v : o_t;
        const double *lg = lg2 + ((j + 1) & 1) * V;
        const double lgl = lg[c1 < V ? c1 : 0], lgb = lg[blank];
        const double s_nb = c1 < V ? pNB + lgl : LOG_ZERO, s_b = pT + lgb;
        sp_half = false;
        if (!on) sp_x = LOG_ZERO;
        else if (s_b <= LOG_ZERO) sp_x = s_nb;
        else if (s_nb <= LOG_ZERO) sp_x = s_b;
        else {
          double x = s_b, y = s_nb;
          if ((y - x) > 0.0) { x = s_nb; y = s_b; }
          sp_x = x; sp_s = 1 + exp(y - x); sp_half = true;
        }
        tots[lane] = sp_s;
      }
      if (S > SURV_MAX) { lds_barrier(); total = arg_max_rounds(); }
      lds_barrier();
      BSTAMP(4);
      const int m = min(W, total);
      Dec d = {false, false, 0, 0, 0.0};
      if (!spec) d = decode_sel(j, m);
      nb = m;
      if (wave == 1) {
        __hip_atomic_store(&s_decflag, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // (the decode's reads have returned: d.src is formed)
        BSTAMP(5);
        // trie node of every fresh labelling; copies keep the id of the slot they come from
        const int p_id = lane_gather(y_node, d.src);
        int id = p_id;
        if (d.fresh) {
          const int parent = p_id, sym = d.sym;
          id = -1;
          const bool lds_ok = parent <= TS;                           // a child of a global node was created after the overflow
          if (lds_ok) {
            const unsigned entry = ((unsigned)(parent + 1) << 17) | (unsigned)sym;
            // double hashing on the (unique) 32-bit entry: the probe sequence of a key is h, h + step, h + 2 step, ... with an odd step,
            // which visits every slot of the power-of-two table; at 3/4 occupancy the longest of a frame's ~20 probe chains -- what the
            // wave waits for -- is a fraction of what linear probing's clusters gave (ids are slot numbers, nothing else depends on the
            // order).  Multiplicative hashing: the HIGH bits of the products -- the low ones depend on the class and a few parent bits only
            unsigned h = (entry * 0x9E3779B1u) >> tshift;
            const unsigned step = ((entry * 0x85EBCA6Bu) >> tshift) | 1u;
            const bool may_insert = s_gnodes == 0;
            for (int probe = 0; probe < TS; ++probe) {
              unsigned prev = may_insert ? atomicCAS(&trie[h], 0u, entry) : trie[h];
              if (prev == entry) { id = (int)h + 1; break; }
              if (prev == 0u) { if (may_insert) id = (int)h + 1; break; }
              h = (h + step) & tmask;
            }
          }
          if (id < 0) {                                               // global table: {parent : 24 | symbol : 16 | id : 24}
            const unsigned long long key40 = ((unsigned long long)(unsigned)parent << 16) | (unsigned)sym;
            unsigned h = (unsigned)mix64(key40) & htmask;
            int gid = -1;
            for (int probe = 0; probe <= (int)htmask; ++probe) {
              const unsigned long long seen = __hip_atomic_load(&ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (seen != HT_EMPTY) {
                if ((seen >> 24) == key40) { id = TS + 1 + (int)(seen & 0xFFFFFFull); break; }
                h = (h + 1) & htmask;
                continue;
              }
              if (gid < 0) gid = atomicAdd(&s_gnodes, 1) - 1;         // s_gnodes = 1 + number of global nodes once overflowed
              const unsigned long long prev = atomicCAS(&ht[h], HT_EMPTY, (key40 << 24) | (unsigned long long)(unsigned)gid);
              if (prev == HT_EMPTY) { id = TS + 1 + gid; if (gid < a.max_nodes) { npar[gid] = parent; nsym[gid] = sym; } break; }
              if ((prev >> 24) == key40) { id = TS + 1 + (int)(prev & 0xFFFFFFull); break; }   // (cannot happen: keys of a frame are distinct)
              h = (h + 1) & htmask;
            }
          }
        }
        {   // LDS trie occupancy: count this frame's fresh lanes; past 3/4 the table is closed for inserts
          const unsigned long long fm = __ballot(d.fresh && id <= TS);
          y_lnodes += __popcll(fm);
          if (lane == 0 && y_lnodes * 4 > TS * 3 && s_gnodes == 0) s_gnodes = 1;
        }
        y_node = d.act ? id : 0;
        nid[(j + 1) & 1][lane] = y_node;
        BSTAMP(6);
      } else if (spec) {
        tots[lane] = sp_half ? sp_x + log(sp_s) : sp_x;
        __hip_atomic_store(&s_totflag, j + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (more) {
        // the new slot's (context class, prNonBlank, prTotal) without wave 0: a fresh labelling carries its candidate's score, a copy the
        // stay / merged entry of the slot it comes from (enbv / homev, written by wave 0 before the last barrier; bm_c1 of the old parity)
        const int o_c1 = bm_c1[j & 1][d.src];
        const double o_nb = enbv[d.src], o_t = homev[d.src];
        stay_totals(j + 1, d.act, d.fresh ? d.sym : o_c1, d.fresh ? d.sv : o_nb, d.fresh ? d.sv : o_t);
      }
      fw = fwn;
      lds_barrier();
      BSTAMP(2);
    }
  }
#ifdef CTCN_BEAM_STATS
  if (a.stats && b == 0 && (tid == 0 || tid == 192)) {
    long long *o = a.stats + (tid == 0 ? 0 : 16);
    for (int i = 0; i < 12; ++i) o[i] = zst[i];
    o[12] = zrounds; o[13] = ziters; o[14] = nfl; o[15] = clock64() - zt0;
  }
#endif
  if (wave == 0) f_node[lane] = nid[nfl & 1][lane];
  __syncthreads();
  if (status == 0 && s_fault) status = 4;
  // final LM step, length normalisation and best labelling (BeamSearch.py:130-151)
  if (status == 0 && tid == 0) {
    int st = 0;
    for (int r = 0; r < nb; ++r) if (f_len[r] == 0) st = 1;          // classes[y[-1]] on () -> IndexError
    if (s_gnodes > a.max_nodes) st = 3;
    const int NB = a.nbest;
    if (st == 0) {
      // as in beam_kernel: normalised scores, then the first `nbest` of a stable descending sort (f_pT is reused for the scores, f_last as "taken")
      for (int r = 0; r < nb; ++r) {
        const double pr = f_pT[r] + a.lm[(size_t)f_last[r] * V1 + V] * a.alpha;
        const double tot = pr;                      // == log_add_prob(LOG_ZERO, pr): its first test returns the second argument (BeamSearch.py:44-45)
        const int ln = f_len[r];
        f_pT[r] = tot * (1.0 / (ln ? ln : 1));
        f_last[r] = 0;
      }
      const int nout = min(NB, nb);
      for (int k = 0; k < nout; ++k) {
        int best = -1; double bestv = 0.0;
        for (int r = 0; r < nb; ++r)
          if (!f_last[r] && (best < 0 || f_pT[r] > bestv)) { best = r; bestv = f_pT[r]; }
        f_last[best] = 1;
        const int ln = f_len[best];
        const size_t o = (size_t)b * NB + k;
        a.out_len[o] = ln; a.out_score[o] = bestv;
        int n = f_node[best];
        for (int i = ln - 1; i >= 0; --i) {
          if (n <= TS) { const unsigned e = trie[n - 1]; a.out_ids[o * T + i] = (int)(e & 0x1FFFFu); n = (int)(e >> 17) - 1; }
          else { const int gq = n - TS - 1; a.out_ids[o * T + i] = nsym[gq]; n = npar[gq]; }
        }
      }
      for (int k = nout; k < NB; ++k) { a.out_len[(size_t)b * NB + k] = 0; a.out_score[(size_t)b * NB + k] = 0.0; }
      if (a.out_count) a.out_count[b] = nout;
    } else {
      for (int k = 0; k < NB; ++k) { a.out_len[(size_t)b * NB + k] = 0; a.out_score[(size_t)b * NB + k] = 0.0; }
      if (a.out_count) a.out_count[b] = 0;
    }
    a.status[b] = st;
  } else if (tid == 0) {
    for (int k = 0; k < a.nbest; ++k) { a.out_len[(size_t)b * a.nbest + k] = 0; a.out_score[(size_t)b * a.nbest + k] = 0.0; }
    if (a.out_count) a.out_count[b] = 0;
    a.status[b] = status;
  }
}

// the two launch forms of the fast search.  beam_fast_kernel: one workgroup per CU (120 VGPRs, up to 144 KB of dynamic LDS: the 16 K-slot
// trie).  beam_fast_kernel_occ2 (round 5, option "beam_occ2"): the same body compiled for eight waves per SIMD (<= 64 VGPRs) and launched
// with <= 68 KB of dynamic LDS, so that TWO utterances share a CU and each one's serial chain runs in the other's shadow -- the body is
// latency-bound (wave 0's ~2 k-cycle chain per frame), so the second workgroup costs the first little as long as more than one workgroup per
// CU is in flight (three 128-utterance batches on three streams: 384 workgroups on 256 CUs).
template <int NPT, bool LM_LDS>
__global__ __launch_bounds__(FAST_NTH) void beam_fast_kernel(FastArgs a) { beam_fast_body<NPT, LM_LDS>(a); }
template <int NPT, bool LM_LDS>
__global__ __launch_bounds__(FAST_NTH) __attribute__((amdgpu_waves_per_eu(8, 8))) void beam_fast_kernel_occ2(FastArgs a) { beam_fast_body<NPT, LM_LDS>(a); }

struct BeamLayout { size_t keys, ids, npar, nsym, cand, nslot, cown, total; int ht_size, max_nodes; };
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
  l.nslot = off; off += align_up((size_t)B * l.max_nodes * sizeof(int), 256);
  l.cown = off; off += align_up((size_t)B * W * V * sizeof(int), 256);
  l.total = off;
  return l;
}

// fast path: [hash table | node parents | node symbols | ln p (T,B,V) double | p_blank (T,B) | log(0) flags (T,B)]
struct FastLayout { size_t ht, npar, nsym, lgd, pb, zf, total; int ht_size, max_nodes, npt, trie_slots; bool ok, lm_lds; size_t lds; };
FastLayout fast_layout(int T, int B, int V, int W, int occ2 = 0) {
  FastLayout l = {};
  l.max_nodes = W * T + 2;
  int ht = 1024;
  while (ht < 2 * l.max_nodes) ht <<= 1;
  l.ht_size = ht;
  size_t off = 0;
  l.ht = off;   off += align_up((size_t)B * ht * sizeof(unsigned long long), 256);
  l.npar = off; off += align_up((size_t)B * l.max_nodes * sizeof(int), 256);
  l.nsym = off; off += align_up((size_t)B * l.max_nodes * sizeof(int), 256);
  l.lgd = off;  off += align_up((size_t)T * B * V * sizeof(double), 256);
  l.pb = off;   off += align_up((size_t)T * B * sizeof(float), 256);
  l.zf = off;   off += align_up((size_t)T * B, 256);
  l.total = off;
  l.npt = ceil_div(W * V, FAST_NCT);                     // candidate slots per thread of waves 3..15 (1..4: W * V <= 3 328)
  if (l.npt > 4) l.npt = 0;
  const size_t core = ((size_t)W * V + 2 * V) * sizeof(double) + ((size_t)W * V + T) * sizeof(int);   // cand, lg[2] | mslot, flist
  const size_t lm = (size_t)(V + 1) * (V + 1) * sizeof(double);
  // of the CU's 160 KB (the kernel also has ~12 KB of static LDS); occ2: two workgroups per CU, 80 KB each
  const size_t budget = occ2 ? 68 * 1024 : 144 * 1024;
  // LDS trie: as many slots as fit, at most 16 K (node ids of the LDS table must fit 14 bits); then the LM if it still fits
  // (occ2 == 2: the LM stays in global memory -- L1 / L2 gathers in the parallel scoring phase -- and the trie gets its share)
  const bool want_lm = lm <= 40 * 1024 && occ2 != 2;
  l.trie_slots = 16384;
  while (l.trie_slots > 1024 && core + (size_t)l.trie_slots * 4 + (want_lm ? lm : 0) > budget) l.trie_slots >>= 1;
  l.lm_lds = occ2 != 2 && core + (size_t)l.trie_slots * 4 + lm <= budget;
  l.lds = core + (size_t)l.trie_slots * 4 + (l.lm_lds ? lm : 0);
  l.ok = W <= 60 && l.npt > 0 && V <= 256 && l.max_nodes < (1 << 24) && T < (1 << 22) && l.lds <= budget;
  return l;
}

#ifdef CTCN_BEAM_STATS
long long *g_beam_stats_dev = nullptr;
#endif
template <int NPT>
int launch_fast(const FastLayout &l, const FastArgs &a, hipStream_t st, int occ2) {
  if (occ2) {
    if (l.lm_lds) {
      auto kern = beam_fast_kernel_occ2<NPT, true>;
      CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
      hipLaunchKernelGGL(kern, dim3(a.B), dim3(FAST_NTH), l.lds, st, a);
    } else {
      auto kern = beam_fast_kernel_occ2<NPT, false>;
      CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
      hipLaunchKernelGGL(kern, dim3(a.B), dim3(FAST_NTH), l.lds, st, a);
    }
    return CTCN_OK;
  }
  if (l.lm_lds) {
    auto kern = beam_fast_kernel<NPT, true>;
    CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(FAST_NTH), l.lds, st, a);
  } else {
    auto kern = beam_fast_kernel<NPT, false>;
    CTCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds));
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(FAST_NTH), l.lds, st, a);
  }
  return CTCN_OK;
}

}  // namespace

extern "C" size_t ctcn_beam_ws_bytes(int T, int B, int V, int W) {
  if (T <= 0 || B <= 0 || V <= 0 || W <= 0) return 0;
  return std::max(beam_layout(T, B, V, W).total, fast_layout(T, B, V, W).total);
}

extern "C" int ctcn_beam_decode_nbest(const float *x, int input_is_prob, const int32_t *lens, const double *lm, double alpha, int W,
                                      int blank, int nbest, int32_t *out_ids, int32_t *out_len, double *out_score, int32_t *out_count,
                                      int32_t *status, int T, int B, int V, void *ws, size_t ws_bytes, void *stream) {
  CTCN_REQUIRE(x && lens && lm && out_ids && out_len && out_score && status && ws, "ctcn_beam_decode: null pointer");
  CTCN_REQUIRE(T > 0 && B > 0 && V > 1 && blank >= 0 && blank < V, "ctcn_beam_decode: bad dims");
  if (W < 1 || W > BEAM_WMAX) { ctcn_set_error("ctcn_beam_decode: beam width %d outside [1,%d]", W, BEAM_WMAX); return CTCN_EUNSUPPORTED; }
  CTCN_REQUIRE(nbest >= 1 && nbest <= W, "ctcn_beam_decode_nbest: nbest %d outside [1, beam width %d]", nbest, W);
  hipStream_t st = (hipStream_t)stream;
  char *base = (char *)ws;
  const int occ2 = ctcn_get_option("beam_occ2");
  const FastLayout fl = fast_layout(T, B, V, W, occ2);
  if (fl.ok && ctcn_get_option("beam_fast") != 0) {
    if (ws_bytes < fl.total) { ctcn_set_error("ctcn_beam_decode: workspace too small (%zu < %zu)", ws_bytes, fl.total); return CTCN_EWORKSPACE; }
    CTCN_HIP(hipMemsetAsync(base + fl.ht, 0xFF, (size_t)B * fl.ht_size * sizeof(unsigned long long), st));
    FastArgs a;
    a.lgd = (const double *)(base + fl.lgd); a.pb = (const float *)(base + fl.pb); a.zf = (const unsigned char *)(base + fl.zf);
    a.lens = lens; a.lm = lm; a.alpha = alpha; a.W = W; a.blank = blank;
    a.out_ids = out_ids; a.out_len = out_len; a.out_score = out_score; a.status = status; a.T = T; a.B = B; a.V = V;
    a.nbest = nbest; a.out_count = out_count;
    a.ht = (unsigned long long *)(base + fl.ht); a.node_par = (int *)(base + fl.npar); a.node_sym = (int *)(base + fl.nsym);
    a.ht_size = fl.ht_size; a.max_nodes = fl.max_nodes; a.trie_slots = fl.trie_slots;
#ifdef CTCN_BEAM_STATS
    if (!g_beam_stats_dev) { CTCN_HIP(hipMalloc(&g_beam_stats_dev, 64 * sizeof(long long))); }
    CTCN_HIP(hipMemsetAsync(g_beam_stats_dev, 0, 64 * sizeof(long long), st));
    a.stats = g_beam_stats_dev;
#endif
    const size_t rows = (size_t)T * B;
    hipLaunchKernelGGL(beam_prep_kernel, dim3((unsigned)ceil_div_z(rows, 4)), dim3(256), 0, st, x, input_is_prob, (double *)(base + fl.lgd),
                       (float *)(base + fl.pb), (unsigned char *)(base + fl.zf), rows, V, blank);
    CTCN_LAUNCH_CHECK();
    int rc;
    switch (fl.npt) {
      case 1: rc = launch_fast<1>(fl, a, st, occ2); break;
      case 2: rc = launch_fast<2>(fl, a, st, occ2); break;
      case 3: rc = launch_fast<3>(fl, a, st, occ2); break;
      default: rc = launch_fast<4>(fl, a, st, occ2); break;
    }
    if (rc) return rc;
    CTCN_LAUNCH_CHECK();
    return CTCN_OK;
  }
  const BeamLayout l = beam_layout(T, B, V, W);
  if (ws_bytes < l.total) { ctcn_set_error("ctcn_beam_decode: workspace too small (%zu < %zu)", ws_bytes, l.total); return CTCN_EWORKSPACE; }
  CTCN_HIP(hipMemsetAsync(base + l.keys, 0xFF, (size_t)B * l.ht_size * sizeof(unsigned long long), st));
  BeamArgs a;
  a.x = x; a.input_is_prob = input_is_prob; a.lens = lens; a.lm = lm; a.alpha = alpha; a.W = W; a.blank = blank;
  a.out_ids = out_ids; a.out_len = out_len; a.out_score = out_score; a.status = status; a.T = T; a.B = B; a.V = V;
    a.nbest = nbest; a.out_count = out_count;
  a.ht_keys = (unsigned long long *)(base + l.keys); a.ht_ids = (int *)(base + l.ids);
  a.node_par = (int *)(base + l.npar); a.node_sym = (int *)(base + l.nsym); a.cand_global = (double *)(base + l.cand);
  a.node_slot = (int *)(base + l.nslot); a.cand_owner = (int *)(base + l.cown);
  a.ht_size = l.ht_size; a.max_nodes = l.max_nodes;
#ifdef CTCN_BEAM_STATS
  if (!g_beam_stats_dev) { CTCN_HIP(hipMalloc(&g_beam_stats_dev, 64 * sizeof(long long))); }
  CTCN_HIP(hipMemsetAsync(g_beam_stats_dev, 0, 64 * sizeof(long long), st));
  a.stats = g_beam_stats_dev;
#endif
  const size_t cand_bytes = (size_t)W * V * sizeof(double);
  const int gthreads = ctcn_get_option("beam_generic_threads");          // 0 (default): by beam width; 256 / 1024: forced (measurements)
  // (measured, tools/beam_generic_probe.py, 128 x 800 batches: 3 720 candidates per frame 7.9 -> 6.3 ms with 1 024 threads, 12 000: 21 -> 16 ms; 1 240: 4.7 -> 4.9)
  const bool wide = gthreads == 1024 || (gthreads != 256 && gthreads != 512 && (W > 64 || (long)W * V > 3500));
  const bool mid = gthreads == 512;
  const void *kern = wide ? reinterpret_cast<const void *>(beam_kernel<1024>) : mid ? reinterpret_cast<const void *>(beam_kernel<512>) : reinterpret_cast<const void *>(beam_kernel<256>);
  // The kernel's STATIC LDS is ~46 KB since the selection holds its survivors there; the candidate table joins it in LDS only if the
  // whole block then fits what this device gives one workgroup (160 KB on gfx950; a 64-KB device falls back to the global table
  // instead of failing the launch).
  hipFuncAttributes fa;
  CTCN_HIP(hipFuncGetAttributes(&fa, kern));
  int dev_id = 0, lds_max = 64 * 1024;
  CTCN_HIP(hipGetDevice(&dev_id));
  if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_id) != hipSuccess) lds_max = 64 * 1024;
  a.wcap = (W + 63) / 64 * 64;
  a.smax = std::max(1024, 2 * a.wcap);                         // (W <= 256: the 1 024 survivors of round 5; W = 1 024: 2 048)
  const size_t state_bytes = (size_t)a.wcap * (10 * sizeof(double) + 10 * sizeof(int)) + (size_t)a.smax * (sizeof(double) + sizeof(int));
  const size_t row_bytes = (size_t)V * sizeof(double);
  // (round 6) whatever the 160 KB hold: the candidate table first (three scans per frame: W = 200 at V = 62 is 99 KB -- in global memory until
  // round 5, whose rule was 32 KB), then the LM table (31.7 KB at V = 62)
  const size_t lm_bytes = (size_t)(V + 1) * (V + 1) * sizeof(double) + 8;
  const size_t fixed = fa.sharedSizeBytes + row_bytes + state_bytes + 256;
  a.bitonic = ctcn_get_option("beam_bitonic");
  a.cand_in_lds = (fixed + cand_bytes <= (size_t)lds_max && !ctcn_get_option("beam_cand_global")) ? 1 : 0;
  a.lm_in_lds = fixed + (a.cand_in_lds ? cand_bytes : 0) + lm_bytes <= (size_t)lds_max ? 1 : 0;
  const size_t sm = row_bytes + state_bytes + (a.cand_in_lds ? cand_bytes : 0) + (a.lm_in_lds ? lm_bytes : 0);
  if (fa.sharedSizeBytes + sm > (size_t)lds_max) {
    ctcn_set_error("ctcn_beam_decode: %zu B of LDS per workgroup needed, the device gives %d", fa.sharedSizeBytes + sm, lds_max);
    return CTCN_EUNSUPPORTED;
  }
  CTCN_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  if (wide) hipLaunchKernelGGL(beam_kernel<1024>, dim3(B), dim3(1024), sm, st, a);
  else if (mid) hipLaunchKernelGGL(beam_kernel<512>, dim3(B), dim3(512), sm, st, a);
  else hipLaunchKernelGGL(beam_kernel<256>, dim3(B), dim3(256), sm, st, a);
  CTCN_LAUNCH_CHECK();
  return CTCN_OK;
}

extern "C" int ctcn_beam_decode(const float *x, int input_is_prob, const int32_t *lens, const double *lm, double alpha, int W,
                                int blank, int32_t *out_ids, int32_t *out_len, double *out_score, int32_t *status, int T, int B,
                                int V, void *ws, size_t ws_bytes, void *stream) {
  return ctcn_beam_decode_nbest(x, input_is_prob, lens, lm, alpha, W, blank, 1, out_ids, out_len, out_score, nullptr, status, T, B, V, ws, ws_bytes, stream);
}

#ifdef CTCN_BEAM_STATS
extern "C" int ctcn_beam_stats(long long *host_out) {       // tools/mb_beam.py only
  if (!g_beam_stats_dev) return CTCN_EINVAL;
  CTCN_HIP(hipMemcpy(host_out, g_beam_stats_dev, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  return CTCN_OK;
}
#endif
