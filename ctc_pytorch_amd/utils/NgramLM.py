"""ARPA bigram language model: host-side parser + dense table for the beam-search kernel.

Same class surface as the reference's timit/utils/NgramLM.py (LanguageModel :11-90): text ARPA with
TAB-separated fields, log10 -> ln, back-off lookup p(w2|w1) = bigram[w1 w2] or backoff(w1)+unigram(w2),
'UNK' aliased to '<unk>'.  `table()` tabulates get_bi_prob over all (previous class, next class) pairs so
that the device kernel (csrc/decode.hip) does one load per LM query.
"""
import math

import numpy as np

n_grams = ["unigram", "bigram", "trigram", "4gram", "5gram"]


class LanguageModel:
    """Attributes the reference exposes: unigram / bigram dicts token(-pair) -> [ln prob, ln back-off], start / end / unk."""

    _SECTION = {"\\1-grams:": "unigram", "\\2-grams:": "bigram"}

    def __init__(self, arpa_file=None, n_gram=2, start="<s>", end="</s>", unk="<unk>"):
        self.n_gram, self.start, self.end, self.unk = n_gram, start, end, unk
        self.scale = math.log(10)          # ARPA stores log10; the decoder works in ln
        self.initngrams(arpa_file)

    def initngrams(self, fn):
        """Parse the \\1-grams: and \\2-grams: sections: 'log10p<TAB>tokens[<TAB>log10 back-off]' per entry."""
        self.unigram, self.bigram = {}, {}
        if self.n_gram == 3:
            self.trigrame = {}
        target = None
        with open(fn, "r") as f:           # open(None) raises TypeError exactly as in the reference (an LM is mandatory)
            for raw in f:
                line = raw.rstrip("\n")
                if line in self._SECTION:
                    target = getattr(self, self._SECTION[line])
                    continue
                if target is None:
                    continue
                parts = line.split("\t")
                if len(parts) in (2, 3):
                    backoff = self.scale * float(parts[2]) if len(parts) == 3 else 0.0
                    target[parts[1]] = [self.scale * float(parts[0]), backoff]
        self.unigram["UNK"] = self.unigram[self.unk]

    def get_uni_prob(self, wid):
        return self.unigram[wid][0]

    def get_bi_prob(self, w1, w2):
        """ln p(w2 | w1) with back-off; '' stands for sentence start (w1) / end (w2)."""
        prev, nxt = w1 or self.start, w2 or self.end
        hit = self.bigram.get(prev + " " + nxt)
        if hit is not None:
            return hit[0]
        return self.unigram[prev][1] + self.unigram[nxt][0]      # KeyError for a phone missing from the ARPA, as the reference

    def score_bg(self, sentence):
        words = sentence.strip().split()
        if not words:
            raise IndexError("score_bg: empty sentence")         # the reference indexes words[0]
        chain = [self.start] + words + [self.end]
        return float(sum(self.get_bi_prob(a, b) for a, b in zip(chain[:-1], chain[1:])))

    def table(self, classes, blank_index=0):
        """(V+1)x(V+1) float64: [c1][c2] = get_bi_prob(classes[c1], classes[c2]); row V = '<s>', column V = '</s>'.
        The blank class is never queried by the decoder (NaN there)."""
        V = len(classes)
        names = [classes[c] for c in range(V)] + [""]
        live = [c for c in range(V + 1) if c != blank_index]
        tab = np.full((V + 1, V + 1), np.nan, dtype=np.float64)
        for c1 in live:
            tab[c1, live] = [self.get_bi_prob(names[c1], names[c2]) for c2 in live]
        return tab
