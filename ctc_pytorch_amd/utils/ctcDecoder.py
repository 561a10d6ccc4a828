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
    """Host-side string assembly and scoring shared by both decoders (reference class of the same name,
    ctcDecoder.py:9-149): label ids -> phone strings, Levenshtein-based CER / WER, running totals."""

    def __init__(self, int2char, space_idx=1, blank_index=0):
        self.int_to_char, self.space_idx, self.blank_index = int2char, space_idx, blank_index
        self.num_word = self.num_char = 0

    def decode(self):
        raise NotImplementedError

    # ---- scoring ---------------------------------------------------------------------------------------
    @staticmethod
    def _edit_distance(src_seq, tgt_seq):
        """Levenshtein distance (unit costs) of two strings or two lists of hashables (ctcDecoder.py:131-150), in the library's host code
        (ctcn_levenshtein): the reference's interpreter loop over an (L1 + 1) x (L2 + 1) table is seconds per long utterance."""
        n_src, n_tgt = len(src_seq), len(tgt_seq)
        if 0 in (n_src, n_tgt):
            return n_src + n_tgt
        if isinstance(src_seq, str) and isinstance(tgt_seq, str):
            a = np.frombuffer(src_seq.encode("utf-32-le", "surrogatepass"), dtype=np.int32)
            b = np.frombuffer(tgt_seq.encode("utf-32-le", "surrogatepass"), dtype=np.int32)
        else:
            ids = {}
            a = np.asarray([ids.setdefault(t, len(ids)) for t in src_seq], dtype=np.int32)
            b = np.asarray([ids.setdefault(t, len(ids)) for t in tgt_seq], dtype=np.int32)
        from ctc_pytorch_amd import _lib
        d = _lib.lib().ctcn_levenshtein(a.ctypes.data, len(a), b.ctypes.data, len(b))
        if d < 0:
            raise RuntimeError("ctcn_levenshtein failed")
        return int(d)

    def cer(self, s1, s2):
        return self._edit_distance(s1, s2)

    def wer(self, s1, s2):
        a, b = s1.split(), s2.split()
        ids = {}
        for w in a + b:
            ids.setdefault(w, len(ids))
        return self._edit_distance([ids[w] for w in a], [ids[w] for w in b])

    def phone_word_error(self, prob_tensor, frame_seq_len, targets, target_sizes):
        hyps = self.decode(prob_tensor, frame_seq_len)
        refs = self._process_strings(self._convert_to_strings(self._unflatten_targets(targets, target_sizes)))
        char_errs = word_errs = 0
        for hyp, ref in zip(hyps, refs):
            char_errs += self.cer(hyp, ref)
            word_errs += self.wer(hyp, ref)
            self.num_word += len(ref.split())
            self.num_char += len(ref)
        return char_errs, word_errs

    # ---- id / string plumbing ----------------------------------------------------------------------------
    @staticmethod
    def _unflatten_targets(targets, target_sizes):
        bounds = np.concatenate([[0], np.cumsum([int(n) for n in target_sizes])])
        return [targets[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1)]

    def _convert_to_string(self, seq, sizes):
        symbols = [self.int_to_char[seq[i]] for i in range(sizes)]
        return symbols if self.space_idx == -1 else "".join(symbols)

    def _convert_to_strings(self, seq, sizes=None):
        return [self._convert_to_string(row, len(row) if sizes is None else sizes[k]) for k, row in enumerate(seq)]

    def _process_string(self, seq, remove_rep=False):
        """Drop blanks (and, optionally, frame-to-frame repeats); phones are joined as ' '+phone when the vocabulary
        has no space symbol (space_idx == -1), else the space symbol becomes ' '."""
        blank = self.int_to_char[self.blank_index]
        space = None if self.space_idx == -1 else self.int_to_char[self.space_idx]
        pieces = []
        for pos, sym in enumerate(seq):
            if sym == blank or (remove_rep and pos > 0 and sym == seq[pos - 1]):
                continue
            pieces.append(" " + sym if space is None else (" " if sym == space else sym))
        return "".join(pieces)

    def _process_strings(self, seqs, remove_rep=False):
        return [self._process_string(one, remove_rep) for one in seqs]


class GreedyDecoder(Decoder):
    def decode_ids(self, prob_tensor, frame_seq_len):
        """(T,B,V) log-probs -> list of collapsed id lists (blank and frame-to-frame repeats removed)."""
        lp = _to_device(prob_tensor)
        idx = ops.argmax_last(lp)                                  # (T,B) int32, lowest index on ties
        ids, out_len = ops.greedy_collapse(idx, frame_seq_len, blank=self.blank_index)
        ids_c, len_c = ids.cpu().numpy(), out_len.cpu().numpy()
        return [list(map(int, ids_c[b, : len_c[b]])) for b in range(ids_c.shape[0])]

    def _strings(self, ids_c, len_c):
        """Collapsed ids (B, T) + lengths -> the reference's strings: each kept frame contributes ' ' + phone when space_idx == -1, else the
        space symbol becomes ' ' (ctcDecoder.py:80-92) -- one native pass (ops.join_tokens) over a vocabulary with that already applied."""
        voc = getattr(self, "_voc", None)
        snap = tuple(self.int_to_char.items()) if isinstance(self.int_to_char, dict) else tuple(self.int_to_char)     # (content, not identity: an edited vocabulary must not decode through the old one)
        if voc is None or voc[0] != snap or voc[1] != self.space_idx:
            items = self.int_to_char.items() if isinstance(self.int_to_char, dict) else enumerate(self.int_to_char)
            if self.space_idx == -1:
                words = {k: " " + w for k, w in items}
            else:
                sp = self.int_to_char[self.space_idx]
                words = {k: (" " if w == sp else w) for k, w in items}
            voc = self._voc = (snap, self.space_idx, words)
        return ops.join_tokens(ids_c, len_c, voc[2], "")

    def decode(self, prob_tensor, frame_seq_len):
        """Same strings as the reference: each kept frame contributes ' '+phone when space_idx == -1."""
        lp = _to_device(prob_tensor)
        idx = ops.argmax_last(lp)
        ids, out_len = ops.greedy_collapse(idx, frame_seq_len, blank=self.blank_index)
        return self._strings(ids.cpu().numpy(), out_len.cpu().numpy())


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
        return self._decoder.decode_strings_async(lp, frame_seq_len, input_is_prob=False)()

    def decode_async(self, prob_tensor, frame_seq_len=None):
        """decode() enqueued on the current stream: returns a callable that waits for this batch alone and returns its strings
        (steps/test_ctc.decode_and_score keeps three batches in flight on three streams: a batch of <= 128 utterances occupies at most
        half of the device for as long as its longest utterance lasts, and the host-side string assembly and scoring of one batch overlaps
        with the search of the next)."""
        lp = _to_device(prob_tensor)
        if frame_seq_len is None:
            frame_seq_len = [lp.shape[0]] * lp.shape[1]
        return self._decoder.decode_strings_async(lp, frame_seq_len, input_is_prob=False)      # (strings assembled in native host code)
