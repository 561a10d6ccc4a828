"""CTC prefix beam search -- class surface of the reference's timit/utils/BeamSearch.py (ctcBeamSearch :35-153),
executed by the gfx950 kernel in csrc/decode.hip (one workgroup per utterance, scores in float64)."""
import torch

from ctc_pytorch_amd import ops

LOG_ZERO = -99999999.0
LOG_ONE = 0.0


class ctcBeamSearch(object):
    def __init__(self, classes, beam_width, lm, lm_alpha=0.01, blank_index=0):
        self.classes = classes
        self.beamWidth = beam_width
        self.lm_alpha = lm_alpha
        self.lm = lm
        self.blank_index = blank_index
        self._table = None

    def _lm_table(self):
        if self._table is None:
            n = len(self.classes)
            self._table = self.lm.table([self.classes[i] for i in range(n)], self.blank_index)
        return self._table

    def _lm_table_on(self, device):
        """The (V+1) x (V+1) table as a float64 tensor on `device`, uploaded once."""
        key = (device.type, device.index)
        if getattr(self, "_table_dev", None) is None or self._table_dev[0] != key:
            self._table_dev = (key, torch.as_tensor(self._lm_table(), dtype=torch.float64).to(device))
        return self._table_dev[1]

    @staticmethod
    def _checked(ids, score, status):
        if (status == 2).any():
            raise ValueError("math domain error")
        if (status == 1).any():
            raise IndexError("tuple index out of range")
        if (status != 0).any():
            raise RuntimeError("beam search kernel status %s" % status)
        return ids, score

    def decode_ids_async(self, x_tbv, lens, input_is_prob=False):
        """decode_ids enqueued on the current stream; returns a callable that waits for this search alone and returns (ids, scores)
        or raises what decode_ids raises."""
        h = ops.beam_decode_async(x_tbv, lens, self._lm_table_on(x_tbv.device), self.lm_alpha, self.beamWidth, self.blank_index, input_is_prob)
        return lambda: self._checked(*h.result())

    def decode_strings_async(self, x_tbv, lens, input_is_prob=False):
        """decode() enqueued on the current stream; returns a callable that waits for this search alone and returns the reference's strings
        (' '.join of the classes, BeamSearch.py:152-153), assembled in native host code from the pinned result buffer."""
        h = ops.beam_decode_async(x_tbv, lens, self._lm_table_on(x_tbv.device), self.lm_alpha, self.beamWidth, self.blank_index, input_is_prob)
        return lambda: self._checked(*h.strings(self.classes, " "))[0]

    def decode_ids(self, x_tbv, lens, input_is_prob=False):
        """x (T,B,V) device tensor -> (list of id lists, float64 scores).  Raises what the reference raises:
        IndexError when an empty labelling reaches the final LM step (BeamSearch.py:135), ValueError on log(0)."""
        ids, score, status = ops.beam_decode(x_tbv, lens, self._lm_table(), self.lm_alpha, self.beamWidth, self.blank_index,
                                             input_is_prob)
        if (status == 2).any():
            raise ValueError("math domain error")
        if (status == 1).any():
            raise IndexError("tuple index out of range")
        if (status != 0).any():
            raise RuntimeError("beam search kernel status %s" % status)
        return ids, score

    def decode(self, inputs, inputs_list):
        """inputs: (B,T,V) tensor of probabilities exp(lp) as the reference passes it (ctcDecoder.py:189-191)."""
        x = inputs.transpose(0, 1)
        if not x.is_cuda:
            x = x.to("cuda")
        return self.decode_strings_async(x, inputs_list, input_is_prob=True)()

    def decode_nbest(self, inputs, inputs_list, nbest):
        """The `nbest` best labellings per utterance, best first, as strings -- `last.sort()[0:nbest]` where the reference's decode keeps
        element [0] (BeamSearch.py:150; SURVEY section 8f-4, the optional n-best output).  Returns (strings: B lists of up to nbest
        strings, scores: (B, nbest) float64 length-normalised log-probabilities)."""
        x = inputs.transpose(0, 1)
        if not x.is_cuda:
            x = x.to("cuda")
        ids, score, status = ops.beam_decode_nbest(x, inputs_list, self._lm_table(), self.lm_alpha, self.beamWidth, nbest, self.blank_index,
                                                   input_is_prob=True)
        self._checked(ids, score, status)
        return [[" ".join(self.classes[k] for k in seq) for seq in utt] for utt in ids], score
