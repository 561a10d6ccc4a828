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
    def __init__(self, arpa_file=None, n_gram=2, start="<s>", end="</s>", unk="<unk>"):
        self.n_gram = n_gram
        self.start = start
        self.end = end
        self.unk = unk
        self.scale = math.log(10)          # ARPA stores log10; the decoder works in ln
        self.initngrams(arpa_file)

    def initngrams(self, fn):
        self.unigram = {}
        self.bigram = {}
        if self.n_gram == 3:
            self.trigrame = {}
        section = 0
        with open(fn, "r") as f:           # open(None) raises TypeError exactly as in the reference (an LM is mandatory)
            for raw in f.readlines():
                line = raw.strip("\n")
                if line == "\\1-grams:":
                    section = 1
                    continue
                if line == "\\2-grams:":
                    section = 2
                    continue
                if section not in (1, 2):
                    continue
                fields = line.split("\t")
                table = self.unigram if section == 1 else self.bigram
                if len(fields) == 3:
                    table[fields[1]] = [self.scale * float(fields[0]), self.scale * float(fields[2])]
                elif len(fields) == 2:
                    table[fields[1]] = [self.scale * float(fields[0]), 0.0]
        self.unigram["UNK"] = self.unigram[self.unk]

    def get_uni_prob(self, wid):
        return self.unigram[wid][0]

    def get_bi_prob(self, w1, w2):
        """ln p(w2 | w1) with back-off; '' stands for sentence start (w1) / end (w2)."""
        if w1 == "":
            w1 = self.start
        if w2 == "":
            w2 = self.end
        key = w1 + " " + w2
        if key not in self.bigram:
            return self.unigram[w1][1] + self.unigram[w2][0]      # KeyError for a phone missing from the ARPA, as the reference
        return self.bigram[key][0]

    def score_bg(self, sentence):
        val = 0.0
        words = sentence.strip().split()
        val += self.get_bi_prob(self.start, words[0])
        for i in range(len(words) - 1):
            val += self.get_bi_prob(words[i], words[i + 1])
        val += self.get_bi_prob(words[-1], self.end)
        return val

    def table(self, classes, blank_index=0):
        """(V+1)x(V+1) float64: [c1][c2] = get_bi_prob(classes[c1], classes[c2]); row V = '<s>', column V = '</s>'.
        The blank class is never queried by the decoder (NaN there)."""
        V = len(classes)
        tab = np.full((V + 1, V + 1), np.nan, dtype=np.float64)
        for c1 in range(V + 1):
            if c1 == blank_index:
                continue
            w1 = "" if c1 == V else classes[c1]
            for c2 in range(V + 1):
                if c2 == blank_index:
                    continue
                tab[c1, c2] = self.get_bi_prob(w1, "" if c2 == V else classes[c2])
        return tab
