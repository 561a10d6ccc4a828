/* oracle/beam_ref.c -- CPU restatement of the reference's CTC prefix beam search.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / timed baseline.  The product path (ctc_pytorch_amd) never links or calls it.
 *
 * Restates, statement by statement, /root/reference/timit/utils/BeamSearch.py:
 *   log_add_prob :43-50   calcExtPr :52-66   BeamState.sort :29-33 (stable, descending prTotal)
 *   BeamState.norm :23-27 decode :73-153
 * in IEEE double with libm log/exp (CPython's math.log/math.exp call the same libm), so scores are
 * bit-identical to the reference on the same machine.  Python dict semantics (insertion order,
 * key = labelling tuple) are reproduced with a prefix trie: a labelling is a trie node id, the dict
 * is an insertion-ordered array + an open-addressing index keyed by node id.
 *
 * Pinned against tests/golden/decoders.{npz,json} (strings produced by the imported reference).
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libbeam_ref.so oracle/beam_ref.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LOG_ZERO (-99999999.0) /* BeamSearch.py:6 */
#define LOG_ONE 0.0

typedef struct { int parent; int sym; int depth; } Node;

typedef struct {
  Node *nodes; int n_nodes, cap_nodes;
  /* child lookup: open addressing on (parent,sym) */
  int *ctab; int ctab_cap;
} Trie;

static uint32_t mix(uint32_t a, uint32_t b) {
  uint64_t x = ((uint64_t)a << 32) | b;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}

static void trie_init(Trie *t) {
  t->cap_nodes = 1 << 16; t->nodes = (Node *)malloc(sizeof(Node) * t->cap_nodes);
  t->n_nodes = 1; t->nodes[0].parent = -1; t->nodes[0].sym = -1; t->nodes[0].depth = 0;
  t->ctab_cap = 1 << 18; t->ctab = (int *)malloc(sizeof(int) * t->ctab_cap);
  memset(t->ctab, 0xff, sizeof(int) * t->ctab_cap);
}
static void trie_free(Trie *t) { free(t->nodes); free(t->ctab); }

static void trie_rehash(Trie *t) {
  int old_cap = t->ctab_cap; int *old = t->ctab;
  t->ctab_cap = old_cap * 2; t->ctab = (int *)malloc(sizeof(int) * t->ctab_cap);
  memset(t->ctab, 0xff, sizeof(int) * t->ctab_cap);
  for (int i = 0; i < old_cap; ++i) if (old[i] >= 0) {
    int id = old[i]; uint32_t h = mix((uint32_t)t->nodes[id].parent, (uint32_t)t->nodes[id].sym) & (t->ctab_cap - 1);
    while (t->ctab[h] >= 0) h = (h + 1) & (t->ctab_cap - 1);
    t->ctab[h] = id;
  }
  free(old);
}

static int trie_child(Trie *t, int parent, int sym) {
  uint32_t h = mix((uint32_t)parent, (uint32_t)sym) & (t->ctab_cap - 1);
  while (t->ctab[h] >= 0) {
    int id = t->ctab[h];
    if (t->nodes[id].parent == parent && t->nodes[id].sym == sym) return id;
    h = (h + 1) & (t->ctab_cap - 1);
  }
  if (t->n_nodes == t->cap_nodes) { t->cap_nodes *= 2; t->nodes = (Node *)realloc(t->nodes, sizeof(Node) * t->cap_nodes); }
  int id = t->n_nodes++;
  t->nodes[id].parent = parent; t->nodes[id].sym = sym; t->nodes[id].depth = t->nodes[parent].depth + 1;
  t->ctab[h] = id;
  if (t->n_nodes * 2 > t->ctab_cap) trie_rehash(t);
  return id;
}

/* BeamState: insertion-ordered entries + index by node id */
typedef struct {
  int *y; double *prT, *prNB, *prB; int n, cap;
  int *idx; int idx_cap; /* open addressing: slot -> entry index, -1 empty */
} State;

static void state_init(State *s, int cap) {
  s->cap = cap; s->n = 0;
  s->y = (int *)malloc(sizeof(int) * cap);
  s->prT = (double *)malloc(sizeof(double) * cap);
  s->prNB = (double *)malloc(sizeof(double) * cap);
  s->prB = (double *)malloc(sizeof(double) * cap);
  s->idx_cap = 1; while (s->idx_cap < 4 * cap) s->idx_cap <<= 1;
  s->idx = (int *)malloc(sizeof(int) * s->idx_cap);
  memset(s->idx, 0xff, sizeof(int) * s->idx_cap);
}
static void state_clear(State *s) { s->n = 0; memset(s->idx, 0xff, sizeof(int) * s->idx_cap); }
static void state_free(State *s) { free(s->y); free(s->prT); free(s->prNB); free(s->prB); free(s->idx); }

/* addLabelling (:68-71): adds labelling if it does not exist yet; returns entry index */
static int state_add(State *s, int y) {
  uint32_t h = mix(0x9e3779b9u, (uint32_t)y) & (s->idx_cap - 1);
  while (s->idx[h] >= 0) {
    if (s->y[s->idx[h]] == y) return s->idx[h];
    h = (h + 1) & (s->idx_cap - 1);
  }
  int e = s->n++;
  s->y[e] = y; s->prT[e] = LOG_ZERO; s->prNB[e] = LOG_ZERO; s->prB[e] = LOG_ZERO; /* BeamEntry() :11-14 */
  s->idx[h] = e;
  return e;
}

static double log_add_prob(double log_x, double log_y) { /* :43-50 */
  if (log_x <= LOG_ZERO) return log_y;
  if (log_y <= LOG_ZERO) return log_x;
  if ((log_y - log_x) > 0.0) { double t = log_x; log_x = log_y; log_y = t; }
  return log_x + log(1 + exp(log_y - log_x));
}

/* BeamState.sort()[0:W] (:29-33,96): stable descending by prTotal -> first W entry indices */
static int top_w(const State *s, int W, int *out) {
  int n = s->n, m = n < W ? n : W;
  /* selection: repeatedly take the best not yet taken; ties -> lowest insertion index (stable) */
  char *taken = (char *)calloc(n > 0 ? n : 1, 1);
  for (int r = 0; r < m; ++r) {
    int best = -1;
    for (int i = 0; i < n; ++i) {
      if (taken[i]) continue;
      if (best < 0 || s->prT[i] > s->prT[best]) best = i;
    }
    taken[best] = 1; out[r] = best;
  }
  free(taken);
  return m;
}

/* status: 0 ok, 1 IndexError (best labelling empty at the end, :135), 2 ValueError (math.log(<=0)) */
/* nbest >= 1: the first `nbest` labellings of the final `last.sort()` (:150; the reference keeps element [0]); outputs [B][nbest]...,
   out_count[B] (may be NULL) = labellings returned (the final beam can hold fewer), the rest have length 0 / score 0 */
int beam_ref_decode_nbest(const float *probs /* [B][T][V] = exp(lp), float32 */, int B, int T, int V,
                    const int *lens, const double *lm /* [(V+1)][(V+1)] ln-probs; row V = <s>, col V = </s> */,
                    double alpha, int W, int blank, int nbest, int *out_ids /* [B][nbest][T] */, int *out_len /* [B][nbest] */,
                    double *out_score /* [B][nbest] */, int *out_count /* [B] or NULL */, int *status /* [B] */) {
  Trie trie; State a, b; State *last = &a, *curr = &b;
  int cap = W * V + W + 8;
  state_init(&a, cap); state_init(&b, cap);
  int *bhat = (int *)malloc(sizeof(int) * ((W > nbest ? W : nbest) > 0 ? (W > nbest ? W : nbest) : 1));
  double *lg = (double *)malloc(sizeof(double) * V);
  for (int bi = 0; bi < B; ++bi) {
    const float *mat = probs + (size_t)bi * T * V;
    trie_init(&trie);
    state_clear(last); state_clear(curr);
    int e0 = state_add(last, 0);              /* y=() :83-87 */
    last->prB[e0] = LOG_ONE; last->prT[e0] = LOG_ONE;
    int st = 0;
    for (int t = 0; t < lens[bi] && !st; ++t) {
      const float *row = mat + (size_t)t * V;
      if ((1.0f - row[blank]) < 0.1f) continue;           /* :93-94, float32 compare (NumPy>=2 scalar rules) */
      state_clear(curr);
      int m = top_w(last, W, bhat);                          /* :96 */
      for (int k = 0; k < V; ++k) lg[k] = 0.0;
      /* math.log(mat[t,k]) is evaluated lazily in the reference; a zero prob only raises when touched.
         Every non-blank k and blank are touched for every beam, so evaluate all up front. */
      for (int k = 0; k < V; ++k) {
        double p = (double)row[k];
        if (!(p > 0.0)) { st = 2; break; }
        lg[k] = log(p);
      }
      if (st) break;
      for (int r = 0; r < m; ++r) {
        int le = bhat[r]; int y = last->y[le];
        int ylen = trie.nodes[y].depth; int ylast = trie.nodes[y].sym;
        double prNonBlank = LOG_ZERO;
        if (ylen > 0) prNonBlank = last->prNB[le] + lg[ylast];       /* :102-103 */
        double prBlank = last->prT[le] + lg[blank];                  /* :106 */
        int ce = state_add(curr, y);                                 /* :108-113 */
        curr->prNB[ce] = log_add_prob(curr->prNB[ce], prNonBlank);
        curr->prB[ce] = log_add_prob(curr->prB[ce], prBlank);
        double prTotal = log_add_prob(prBlank, prNonBlank);
        curr->prT[ce] = log_add_prob(curr->prT[ce], prTotal);
        for (int k = 0; k < V; ++k) {                                /* :116-125 */
          if (k == blank) continue;
          int newY = trie_child(&trie, y, k);
          /* calcExtPr :52-66 */
          int c1 = ylen ? ylast : V;
          double bigramProb = lm[(size_t)c1 * (V + 1) + k] * alpha;
          double pr;
          if (ylen && ylast == k && mat[(size_t)(t - 1) * V + blank] < 0.9f)
            pr = lg[k] + bigramProb + last->prB[le];
          else
            pr = lg[k] + bigramProb + last->prT[le];
          int ne = state_add(curr, newY);
          curr->prNB[ne] = log_add_prob(curr->prNB[ne], pr);
          curr->prT[ne] = log_add_prob(curr->prT[ne], pr);
        }
      }
      State *tmp = last; last = curr; curr = tmp;                    /* :128 */
    }
    for (int k = 0; k < nbest; ++k) { out_len[(size_t)bi * nbest + k] = 0; out_score[(size_t)bi * nbest + k] = 0.0; }
    if (out_count) out_count[bi] = 0;
    if (!st) {
      int m = top_w(last, W, bhat);                                  /* :130 */
      state_clear(curr);
      for (int r = 0; r < m && !st; ++r) {                           /* :133-141 */
        int le = bhat[r]; int y = last->y[le];
        if (trie.nodes[y].depth == 0) { st = 1; break; }             /* classes[y[-1]] on () -> IndexError */
        int c1 = trie.nodes[y].sym;
        double pr = last->prT[le] + lm[(size_t)c1 * (V + 1) + V] * alpha;
        int ne = state_add(curr, y);
        curr->prNB[ne] = log_add_prob(curr->prNB[ne], pr);
        curr->prT[ne] = log_add_prob(curr->prT[ne], pr);
      }
      if (!st) {
        for (int i = 0; i < curr->n; ++i) {                          /* norm :23-27 */
          int len = trie.nodes[curr->y[i]].depth;
          curr->prT[i] = curr->prT[i] * (1.0 / (len ? len : 1));
        }
        int nout = top_w(curr, nbest, bhat);                         /* :148-150: last.sort()[0] -- here [0:nbest] */
        for (int k = 0; k < nout; ++k) {
          int best = bhat[k];
          int y = curr->y[best]; int len = trie.nodes[y].depth;
          size_t o = (size_t)bi * nbest + k;
          out_len[o] = len; out_score[o] = curr->prT[best];
          for (int i = len - 1, n = y; i >= 0; --i, n = trie.nodes[n].parent) out_ids[o * T + i] = trie.nodes[n].sym;
        }
        if (out_count) out_count[bi] = nout;
      }
    }
    status[bi] = st;
    trie_free(&trie);
  }
  free(bhat); free(lg); state_free(&a); state_free(&b);
  return 0;
}

int beam_ref_decode(const float *probs, int B, int T, int V, const int *lens, const double *lm, double alpha, int W, int blank,
                    int *out_ids /* [B][T] */, int *out_len /* [B] */, double *out_score /* [B] */, int *status /* [B] */) {
  return beam_ref_decode_nbest(probs, B, T, V, lens, lm, alpha, W, blank, 1, out_ids, out_len, out_score, 0, status);
}
