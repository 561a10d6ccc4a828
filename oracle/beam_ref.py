"""ctypes wrapper + LM-table builder for the C beam-search oracle  --  TEST INFRASTRUCTURE ONLY.

`build()` compiles oracle/beam_ref.c with gcc into oracle/_build/libbeam_ref.so (git-ignored, but it
travels with gpurun snapshots).  `arpa_table()` restates the ARPA parser / back-off lookup of
/root/reference/timit/utils/NgramLM.py:25-78 and tabulates get_bi_prob for every (prev, next) class
pair the decoder can ask for: table[c1][c2] with row V = '<s>' (c1 == '') and column V = '</s>'.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SO = os.path.join(BUILD, "libbeam_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "beam_ref.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", SO, src, "-lm"])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.beam_ref_decode.restype = ctypes.c_int
        _lib.beam_ref_decode_nbest.restype = ctypes.c_int
    return _lib


def parse_arpa(fn):
    """NgramLM.initngrams (:25-58): tab-separated entries, log10 -> ln, 'UNK' aliases '<unk>'."""
    scale = math.log(10)
    unigram, bigram = {}, {}
    recording = 0
    with open(fn, "r") as f:
        for lines in f.readlines():
            line = lines.strip("\n")
            if line == "\\1-grams:":
                recording = 1
                continue
            if line == "\\2-grams:":
                recording = 2
                continue
            if recording in (1, 2):
                tgt = unigram if recording == 1 else bigram
                parts = line.split("\t")
                if len(parts) == 3:
                    tgt[parts[1]] = [scale * float(parts[0]), scale * float(parts[2])]
                elif len(parts) == 2:
                    tgt[parts[1]] = [scale * float(parts[0]), 0.0]
    unigram["UNK"] = unigram["<unk>"]
    return unigram, bigram


def get_bi_prob(unigram, bigram, w1, w2):
    """NgramLM.get_bi_prob (:65-78)."""
    if w1 == "":
        w1 = "<s>"
    if w2 == "":
        w2 = "</s>"
    key = w1 + " " + w2
    if key not in bigram:
        return unigram[w1][1] + unigram[w2][0]
    return bigram[key][0]


def arpa_table(fn, int2char, blank=0):
    V = len(int2char)
    unigram, bigram = parse_arpa(fn)
    tab = np.full((V + 1, V + 1), np.nan, dtype=np.float64)
    for c1 in range(V + 1):
        if c1 == blank:
            continue
        w1 = "" if c1 == V else int2char[c1]
        for c2 in range(V + 1):
            if c2 == blank:
                continue
            w2 = "" if c2 == V else int2char[c2]
            tab[c1, c2] = get_bi_prob(unigram, bigram, w1, w2)
    return tab


def decode_ids(probs_btv, lens, lm_table, alpha, W, blank=0):
    """probs_btv: (B,T,V) float32 = exp(log-probs).  Returns (ids list per utt, scores, status)."""
    probs = np.ascontiguousarray(probs_btv, dtype=np.float32)
    B, T, V = probs.shape
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    lm = np.ascontiguousarray(lm_table, dtype=np.float64)
    out_ids = np.zeros((B, T), dtype=np.int32)
    out_len = np.zeros(B, dtype=np.int32)
    score = np.zeros(B, dtype=np.float64)
    status = np.zeros(B, dtype=np.int32)
    P = ctypes.c_void_p
    lib().beam_ref_decode(P(probs.ctypes.data), B, T, V, P(lens.ctypes.data), P(lm.ctypes.data),
                          ctypes.c_double(alpha), W, blank, P(out_ids.ctypes.data), P(out_len.ctypes.data),
                          P(score.ctypes.data), P(status.ctypes.data))
    return [list(out_ids[b, : out_len[b]]) for b in range(B)], score, status


def decode_ids_nbest(probs_btv, lens, lm_table, alpha, W, nbest, blank=0):
    """The first `nbest` labellings of the final `last.sort()` (BeamSearch.py:150 keeps [0]).  Returns (ids: per utterance a list of up to
    nbest label lists, scores (B, nbest) float64, status (B))."""
    probs = np.ascontiguousarray(probs_btv, dtype=np.float32)
    B, T, V = probs.shape
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    lm = np.ascontiguousarray(lm_table, dtype=np.float64)
    out_ids = np.zeros((B, nbest, T), dtype=np.int32)
    out_len = np.zeros((B, nbest), dtype=np.int32)
    score = np.zeros((B, nbest), dtype=np.float64)
    count = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    P = ctypes.c_void_p
    lib().beam_ref_decode_nbest(P(probs.ctypes.data), B, T, V, P(lens.ctypes.data), P(lm.ctypes.data), ctypes.c_double(alpha), W, blank, nbest,
                                P(out_ids.ctypes.data), P(out_len.ctypes.data), P(score.ctypes.data), P(count.ctypes.data), P(status.ctypes.data))
    return [[list(map(int, out_ids[b, k, : out_len[b, k]])) for k in range(count[b])] for b in range(B)], score, status


def decode_strings(probs_btv, lens, lm_table, alpha, W, int2char, blank=0):
    ids, score, status = decode_ids(probs_btv, lens, lm_table, alpha, W, blank)
    res = []
    for b, seq in enumerate(ids):
        if status[b] == 1:
            raise IndexError("tuple index out of range")          # BeamSearch.py:135 on an empty labelling
        if status[b] == 2:
            raise ValueError("math domain error")                 # math.log(0) (BeamSearch.py:64,66,103,106)
        res.append(" ".join(int2char[int(k)] for k in seq))        # :151
    return res
