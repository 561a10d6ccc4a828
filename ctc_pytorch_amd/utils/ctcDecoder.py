"""Greedy and beam CTC decoders -- class surface of the reference's timit/utils/ctcDecoder.py
(Decoder :9-149, GreedyDecoder :152-166, BeamDecoder :168-192).

String conversion and CER/WER scoring are host logic (python, as in the reference); the arithmetic --
arg-max over classes, path collapse, prefix beam search -- runs in libctcn.so on the GPU.  `decode`
accepts the CPU tensor the reference hands over (test_ctc.py:85-86) and moves it to the device itself.
"""
import numpy as np
import torch

from ctc_pytorch_amd import ops


def _to_device(t):
    return t if t.is_cuda else t.to("cuda")


class Decoder(object):
    def __init__(self, int2char, space_idx=1, blank_index=0):
        self.int_to_char = int2char
        self.space_idx = space_idx
        self.blank_index = blank_index
        self.num_word = 0
        self.num_char = 0

    def decode(self):
        raise NotImplementedError

    def phone_word_error(self, prob_tensor, frame_seq_len, targets, target_sizes):
        strings = self.decode(prob_tensor, frame_seq_len)
        targets = self._unflatten_targets(targets, target_sizes)
        target_strings = self._process_strings(self._convert_to_strings(targets))
        cer = 0
        wer = 0
        for x in range(len(target_strings)):
            cer += self.cer(strings[x], target_strings[x])
            wer += self.wer(strings[x], target_strings[x])
            self.num_word += len(target_strings[x].split())
            self.num_char += len(target_strings[x])
        return cer, wer

    def _unflatten_targets(self, targets, target_sizes):
        out, offset = [], 0
        for size in target_sizes:
            out.append(targets[offset:offset + size])
            offset += size
        return out

    def _process_strings(self, seqs, remove_rep=False):
        return [self._process_string(seq, remove_rep) for seq in seqs]

    def _process_string(self, seq, remove_rep=False):
        blank = self.int_to_char[self.blank_index]
        string = ""
        for i, char in enumerate(seq):
            if char == blank:
                continue
            if remove_rep and i != 0 and char == seq[i - 1]:
                continue
            if self.space_idx == -1:
                string = string + " " + char
            elif char == self.int_to_char[self.space_idx]:
                string += " "
            else:
                string = string + char
        return string

    def _convert_to_strings(self, seq, sizes=None):
        strings = []
        for x in range(len(seq)):
            n = sizes[x] if sizes is not None else len(seq[x])
            strings.append(self._convert_to_string(seq[x], n))
        return strings

    def _convert_to_string(self, seq, sizes):
        result = [self.int_to_char[seq[i]] for i in range(sizes)]
        return result if self.space_idx == -1 else "".join(result)

    def wer(self, s1, s2):
        vocab = set(s1.split() + s2.split())
        word2int = dict(zip(vocab, range(len(vocab))))
        return self._edit_distance([word2int[w] for w in s1.split()], [word2int[w] for w in s2.split()])

    def cer(self, s1, s2):
        return self._edit_distance(s1, s2)

    def _edit_distance(self, src_seq, tgt_seq):
        L1, L2 = len(src_seq), len(tgt_seq)
        if L1 == 0:
            return L2
        if L2 == 0:
            return L1
        prev = list(range(L2 + 1))
        for i in range(1, L1 + 1):
            cur = [i] + [0] * L2
            a = src_seq[i - 1]
            for j in range(1, L2 + 1):
                cur[j] = min(cur[j - 1] + 1, prev[j] + 1, prev[j - 1] + (0 if a == tgt_seq[j - 1] else 1))
            prev = cur
        return prev[L2]


class GreedyDecoder(Decoder):
    def decode_ids(self, prob_tensor, frame_seq_len):
        """(T,B,V) log-probs -> list of collapsed id lists (blank and frame-to-frame repeats removed)."""
        lp = _to_device(prob_tensor)
        idx = ops.argmax_last(lp)                                  # (T,B) int32, lowest index on ties
        ids, out_len = ops.greedy_collapse(idx, frame_seq_len, blank=self.blank_index)
        ids_c, len_c = ids.cpu().numpy(), out_len.cpu().numpy()
        return [list(map(int, ids_c[b, : len_c[b]])) for b in range(ids_c.shape[0])]

    def decode(self, prob_tensor, frame_seq_len):
        """Same strings as the reference: each kept frame contributes ' '+phone when space_idx == -1."""
        res = []
        for seq in self.decode_ids(prob_tensor, frame_seq_len):
            chars = [self.int_to_char[k] for k in seq]
            if self.space_idx == -1:
                res.append("".join(" " + c for c in chars))
            else:
                sp = self.int_to_char[self.space_idx]
                res.append("".join(" " if c == sp else c for c in chars))
        return res


class BeamDecoder(Decoder):
    def __init__(self, int2char, beam_width=200, blank_index=0, space_idx=-1, lm_path=None, lm_alpha=0.01):
        self.beam_width = beam_width
        super().__init__(int2char, space_idx=space_idx, blank_index=blank_index)
        from ctc_pytorch_amd.utils import BeamSearch as uBeam
        from ctc_pytorch_amd.utils import NgramLM as uNgram
        lm = uNgram.LanguageModel(arpa_file=lm_path)
        self._decoder = uBeam.ctcBeamSearch(int2char, beam_width, lm, lm_alpha=lm_alpha, blank_index=blank_index)

    def decode(self, prob_tensor, frame_seq_len=None):
        """prob_tensor (T,B,V) log-probs (CPU or device).  exp() is taken on the device in float32."""
        lp = _to_device(prob_tensor)
        if frame_seq_len is None:
            frame_seq_len = [lp.shape[0]] * lp.shape[1]
        ids, _ = self._decoder.decode_ids(lp, frame_seq_len, input_is_prob=False)
        return [" ".join(self.int_to_char[k] for k in seq) for seq in ids]
