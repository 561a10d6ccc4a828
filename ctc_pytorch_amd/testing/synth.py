"""Synthetic, seed-reproducible inputs for the CTC hot path.

Imported by tests/, tools/, bench.py's input generation and oracle/gen_golden.py -- never by
the product modules of this package.  Contains no arithmetic of the path,
only NumPy ``RandomState`` data generators, so that the same tensors can be
re-created here (next to the reference) and on the GPU box (without it).

Shapes follow SURVEY.md §8(d) / BASELINE.md §3:
  x ~ N(0,1) f32 (B,T,F); labels U{2..V-1} (no blank=0 / UNK=1); lengths as the
  reference collate stores them (fractions len/Tmax in float32,
  /root/reference/timit/utils/data_loader.py:137).
"""
import math
import numpy as np

# The 61 TIMIT phones minus 'q' (the reference drops it, conf/phones.60-48-39.map:47) = 60 symbols.
TIMIT_60 = (
    "aa ae ah ao aw ax ax-h axr ay b bcl ch d dcl dh dx eh el em en eng epi er ey f g gcl "
    "h# hh hv ih ix iy jh k kcl l m n ng nx ow oy p pau pcl r s sh t tcl th uh uw ux v w y z zh"
).split()
assert len(TIMIT_60) == 60


def int2char(num_class=62):
    """index2word table as Vocab builds it (data_loader.py:16-17): 0=blank, 1=UNK, then units."""
    d = {0: "blank", 1: "UNK"}
    for i in range(2, num_class):
        d[i] = TIMIT_60[i - 2] if i - 2 < len(TIMIT_60) else "u%03d" % i
    return d


def make_batch(seed, B, T, F, V, min_len=None, lab_lo=30, lab_hi=60, full_length=False):
    """Padded utterance minibatch in the reference's input contract.

    Returns dict(x (B,T,F) f32 zero-padded, lens (B,) int, frac (B,) f32 = len/T,
    targets (B,Lmax) i64 zero-padded, tgt_len (B,) i64)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, T, F)).astype(np.float32)
    if full_length:
        lens = np.full((B,), T, dtype=np.int64)
    else:
        lo = T // 2 if min_len is None else min_len
        lens = rs.randint(lo, T + 1, size=B).astype(np.int64)
        lens[0] = T  # one utterance defines Tmax, as in a real collate
    for b in range(B):
        x[b, lens[b]:] = 0.0
    tl = rs.randint(lab_lo, lab_hi + 1, size=B).astype(np.int64)
    # keep every sample CTC-feasible even after a /2 time stride: need T_b/2 >= L + repeats
    tl = np.minimum(tl, np.maximum(1, lens // 5))
    Lmax = int(tl.max())
    targets = np.zeros((B, Lmax), dtype=np.int64)
    for b in range(B):
        targets[b, : tl[b]] = rs.randint(2, V, size=tl[b])
    frac = np.array([np.float32(float(l) / float(T)) for l in lens], dtype=np.float32)
    return dict(x=x, lens=lens, frac=frac, targets=targets, tgt_len=tl)


def fill_state_dict(shapes, seed):
    """Deterministic parameter values for a CTC_Model state_dict.

    ``shapes``: ordered list of (key, shape).  RNN/Linear/Conv weights ~ U(-b,b) with
    b = 1/sqrt(fan), BN gamma ~ 1+0.1N, beta ~ 0.1N,
    running_mean=0, running_var=1, num_batches_tracked=0."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(shape)
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = np.zeros(shape, dtype=np.float32)
        elif leaf == "running_var":
            out[key] = np.ones(shape, dtype=np.float32)
        elif "batch_norm" in key or key.startswith("fc.0."):
            if leaf == "weight":
                out[key] = (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
            else:
                out[key] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
        else:
            if len(shape) == 1:
                fan = shape[0]
            elif len(shape) == 2:
                fan = shape[1]
            else:
                fan = int(np.prod(shape[1:]))
            b = 1.0 / math.sqrt(max(fan, 1))
            out[key] = rs.uniform(-b, b, size=shape).astype(np.float32)
    return out


def write_arpa(path, units, seed=7, n_bigrams=600):
    """Synthetic phone bigram LM in the exact text format NgramLM.initngrams parses
    (/root/reference/timit/utils/NgramLM.py:38-56): header lines ``\\1-grams:`` /
    ``\\2-grams:``, TAB-separated ``log10prob<TAB>token[<TAB>log10backoff]``."""
    rs = np.random.RandomState(seed)
    toks = ["<s>", "</s>", "<unk>"] + list(units)
    lines = ["", "\\data\\", "ngram 1=%d" % len(toks), "ngram 2=%d" % n_bigrams, "", "\\1-grams:"]
    for t in toks:
        p = -3.0 * rs.random_sample()
        if t == "</s>":
            lines.append("%.6f\t%s" % (p, t))
        else:
            bo = -1.0 * rs.random_sample()
            lines.append("%.6f\t%s\t%.6f" % (p, t, bo))
    lines += ["", "\\2-grams:"]
    seen = set()
    firsts = [t for t in toks if t != "</s>"]
    seconds = [t for t in toks if t != "<s>"]
    while len(seen) < n_bigrams:
        a = firsts[rs.randint(len(firsts))]
        b = seconds[rs.randint(len(seconds))]
        if (a, b) in seen:
            continue
        seen.add((a, b))
        lines.append("%.6f\t%s %s" % (-3.0 * rs.random_sample(), a, b))
    lines += ["", "\\end\\", ""]
    with open(path, "w") as f:
        f.write("\n".join(lines))


def _log_softmax(z):
    z = z.astype(np.float64)
    m = z.max(axis=-1, keepdims=True)
    return (z - m - np.log(np.exp(z - m).sum(axis=-1, keepdims=True))).astype(np.float32)


def make_logits(seed, T, B, V, regime="peaky", blank_frac=0.6):
    """(T,B,V) f32 *logits* for decoder tests.
    peaky: 8*onehot(random CTC path with ~60 % blank frames) + N(0,1); flat: 3*N(0,1)."""
    rs = np.random.RandomState(seed)
    if regime == "flat":
        return (3.0 * rs.standard_normal((T, B, V))).astype(np.float32)
    z = rs.standard_normal((T, B, V)).astype(np.float32)
    for b in range(B):
        t = 0
        while t < T:
            if rs.random_sample() < blank_frac:
                k, run = 0, rs.randint(1, 6)
            else:
                k, run = rs.randint(2, V), rs.randint(1, 4)
            z[t : t + run, b, k] += 8.0
            t += run
    return z


def make_logprobs(seed, T, B, V, regime="peaky"):
    return _log_softmax(make_logits(seed, T, B, V, regime))
