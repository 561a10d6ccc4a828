"""Input pipeline producing the hot path's input contract -- counterpart of the reference's
timit/utils/data_loader.py (Vocab :13-47, SpeechDataset :50-116, create_input :119-140, SpeechDataLoader
:148-151) and of the two feature helpers it uses from timit/utils/tools.py (make_context :66-75, skip_feat :77-86).

kaldiio is not available, so Kaldi archives are read by the small binary reader below (uncompressed float /
double matrices, `scp` entries of the form `utt path:offset`).  Batches have exactly the reference layout:
(inputs (B,Tmax,F) float32 zero-padded, input_sizes (B) float32 FRACTIONS len/Tmax, targets (B,Lmax) int64
zero-padded, target_sizes (B) int64, utt_list).
"""
import struct

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


# ---- Kaldi binary matrix I/O -----------------------------------------------------------------------------
def read_kaldi_matrix(path_with_offset):
    """'file.ark:12345' -> float32 ndarray (rows, cols).  Supports binary 'FM ' and 'DM ' matrices."""
    if ":" in path_with_offset and not path_with_offset.rsplit(":", 1)[1].strip() == "":
        path, off = path_with_offset.rsplit(":", 1)
        off = int(off)
    else:
        path, off = path_with_offset, 0
    with open(path, "rb") as f:
        f.seek(off)
        head = f.read(15)                                 # "\0B" + "FM " + "\4" rows(4) + "\4" cols(4): one read instead of nine
        if head[:2] != b"\0B":
            raise ValueError("%s: not a binary Kaldi object at offset %d" % (path, off))
        tok = head[2:5]
        if tok not in (b"FM ", b"DM "):
            raise NotImplementedError("Kaldi matrix type %r (compressed matrices are not supported)" % head[2:5].split(b" ")[0].decode("latin-1"))
        if head[5:6] != b"\4" or head[10:11] != b"\4":
            raise ValueError("%s: malformed matrix header at offset %d" % (path, off))
        rows, cols = struct.unpack("<i", head[6:10])[0], struct.unpack("<i", head[11:15])[0]
        dt = np.dtype("<f4") if tok == b"FM " else np.dtype("<f8")
        data = np.empty((rows, cols), dtype=dt)
        if f.readinto(data) != data.nbytes:               # straight into the array that is returned (no bytes object, no second copy)
            raise ValueError("%s: truncated matrix at offset %d" % (path, off))
    return data if dt == np.dtype("<f4") else data.astype(np.float32)


def write_kaldi_ark(ark_path, scp_path, mats):
    """mats: dict utt -> float32 (rows, cols).  Writes a binary ark + scp (as Kaldi copy-feats would)."""
    with open(ark_path, "wb") as fa, open(scp_path, "w") as fs:
        for utt, m in mats.items():
            m = np.ascontiguousarray(m, dtype="<f4")
            fa.write(utt.encode() + b" ")
            fs.write("%s %s:%d\n" % (utt, ark_path, fa.tell()))
            fa.write(b"\0BFM " + b"\4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]))
            fa.write(m.tobytes())


# ---- feature helpers -------------------------------------------------------------------------------------
def make_context(feature, left, right, as_view=False):
    """Splice `left` past and `right` future frames (edge frames repeated) -> (T, (left+1+right)*F).  as_view: return the overlapping
    window view itself (read-only use: the dataset hands it to the collate function, whose copy into the batch is then the only one)."""
    if left == 0 and right == 0:
        return feature
    # row t of the result = rows t - left .. t + right of the edge-padded matrix back to back, i.e. ONE contiguous run of (left + 1 + right) * F
    # floats of it: a strided window view, materialised by a single row-wise copy (the stack of 9 gathered copies this replaces ran at
    # 0.7 M frames/s for the shipped 243-d configuration, below what the GPU trains at)
    feature = np.ascontiguousarray(feature)
    T, F = feature.shape
    if T == 0:
        return np.zeros((0, (left + 1 + right) * F), dtype=feature.dtype)
    padded = np.concatenate([np.repeat(feature[:1], left, axis=0), feature, np.repeat(feature[-1:], right, axis=0)])
    item = padded.itemsize
    view = np.lib.stride_tricks.as_strided(padded, shape=(T, (left + 1 + right) * F), strides=(F * item, item))
    return view if as_view else np.ascontiguousarray(view)


def skip_feat(feature, skip):
    """Keep every `skip`-th frame starting at 0."""
    if skip in (0, 1):
        return feature
    return feature[::skip]


# ---- vocabulary / dataset / collate -----------------------------------------------------------------------
class Vocab(object):
    """Phone / character inventory of a `units` file (timit/utils/data_loader.py:12-48): ids 0 and 1 are the CTC blank and UNK, every other
    symbol gets the next free id in order of first appearance; a line is either `symbol` or `key symbol symbol ...` (the key is dropped).
    Public surface the drivers and checkpoints read: word2index, index2word, word2count, n_words (+ add_word / add_sentence for callers that
    extend an inventory).  Built in one pass: the file's symbols are counted, then numbered from the counter's insertion order."""

    RESERVED = ("blank", "UNK")

    def __init__(self, vocab_file):
        self.vocab_file = vocab_file
        self.word2index = {w: i for i, w in enumerate(self.RESERVED)}
        self.index2word = dict(enumerate(self.RESERVED))
        self.word2count = {}
        self.read_lang()

    @property
    def n_words(self):
        return len(self.word2index)

    def _absorb(self, symbols):
        """Count `symbols` (an iterable) and number the unseen ones behind the current inventory."""
        from collections import Counter
        seen = Counter(symbols)                    # insertion-ordered: first appearance decides the id
        for w, c in seen.items():
            if w in self.word2index:               # (a reserved name inside the file raises KeyError in the reference's counter too)
                self.word2count[w] += c
                continue
            k = len(self.word2index)
            self.word2index[w], self.index2word[k], self.word2count[w] = k, w, c

    def add_word(self, word):
        self._absorb((word,))

    def add_sentence(self, sentence):
        self._absorb(sentence.split(" "))

    def read_lang(self):
        def symbols(fh):
            for raw in fh:
                fields = raw.strip().split(" ")
                yield from (fields[1:] if len(fields) > 1 else fields)
        with open(self.vocab_file, "r") as fh:
            self._absorb(symbols(fh))


class SpeechDataset(Dataset):
    def __init__(self, vocab, scp_path, lab_path, opts):
        self.vocab = vocab
        self.left_ctx = opts.left_ctx
        self.right_ctx = opts.right_ctx
        self.n_skip_frame = opts.n_skip_frame
        self.n_downsample = opts.n_downsample
        paths = []
        with open(scp_path, "r") as rf:
            for line in rf:
                utt, path = line.strip().split(" ")
                paths.append((utt, path))
        labels = {}
        unk = vocab.word2index["UNK"]
        with open(lab_path, "r") as rf:
            for line in rf:
                utt, label = line.strip().split(" ", 1)
                labels[utt] = [vocab.word2index.get(c, unk) for c in label.split()]
        assert len(paths) == len(labels)
        self.item = [(path, labels[utt], utt) for utt, path in paths]

    def __getitem__(self, idx):
        path, label, utt = self.item[idx]
        # (the spliced / frame-skipped matrix stays a strided view of the matrix just read -- an array of its own -- until create_input copies
        # it into the batch: one pass over the 9x larger spliced data instead of three)
        feat = skip_feat(make_context(read_kaldi_matrix(path), self.left_ctx, self.right_ctx, as_view=True), self.n_skip_frame)
        seq_len, dim = feat.shape
        if seq_len % self.n_downsample != 0:
            pad_len = self.n_downsample - seq_len % self.n_downsample
            feat = np.vstack([feat, np.zeros((pad_len, dim), dtype=feat.dtype)])
        return torch.from_numpy(feat), torch.LongTensor(label), utt

    def __len__(self):
        return len(self.item)


def create_input(batch):
    tmax = max(x[0].size(0) for x in batch)
    feat_size = batch[0][0].size(1)
    lmax = max(x[1].size(0) for x in batch)
    n = len(batch)
    # (every element is written exactly once -- the utterance's frames, then zeros behind them -- instead of zero-filling the whole padded
    # batch first: the batch is 3-25 MB and this runs on the thread that also enqueues the training step)
    data = torch.empty(n, tmax, feat_size)
    label = torch.zeros(n, lmax, dtype=torch.long)
    input_sizes = torch.empty(n)
    target_sizes = torch.empty(n, dtype=torch.long)
    utt_list = []
    for i, (feature, lab, utt) in enumerate(batch):
        t = feature.size(0)
        data[i, :t] = feature
        if t < tmax:
            data[i, t:].zero_()
        label[i, : lab.size(0)] = lab
        input_sizes[i] = t / tmax                        # python float -> float32 element: the fraction the path consumes
        target_sizes[i] = lab.size(0)
        utt_list.append(utt)
    return data, input_sizes, label, target_sizes, utt_list


class SpeechDataLoader(DataLoader):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.collate_fn = create_input


class DevicePrefetcher(object):
    """Double-buffered asynchronous host->device staging of the batches of a SpeechDataLoader.

    While step n computes, batch n+1 is copied from pinned host memory on a dedicated copy stream; the iterator yields
    the loader's 5-tuple with `inputs`, `input_sizes` (the float32 length fractions), `targets` and `target_sizes` already
    resident on the device; run_epoch does the length conversion of train_ctc.py:46 there (same float32 product and
    truncation), so no pageable copy blocks the host in the middle of a step.  4.1 MB per cfg2 batch = ~65 us of PCIe Gen5
    time, hidden behind a 16 ms step."""

    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        # Two staging slots of persistent pinned host memory, grown on demand to the largest batch seen.  (A fresh
        # tensor.pin_memory() per batch is a hipHostMalloc per step: it drains the device and cost 10-18 ms per cfg2 step.)
        self._slots = [{"bufs": {}, "copied": None} for _ in range(2)]
        self._turn = 0

    def __len__(self):
        return len(self.loader)

    @staticmethod
    def _pinned(slot, key, t):
        buf = slot["bufs"].get(key)
        if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
            buf = torch.empty(max(64, int(t.numel() * 1.25)), dtype=t.dtype, pin_memory=True)
            slot["bufs"][key] = buf
        view = buf[:t.numel()].view(t.shape)
        # single-threaded memcpy on purpose: Tensor.copy_ fans a 4 MB copy out to every host core, and the OpenMP team spinning
        # next to the HIP runtime's threads cost 8-10 ms per cfg2 step (tools/epoch_probe.py); one core copies 4 MB in 0.4 ms
        np.copyto(view.numpy(), t.detach().numpy())
        return view

    def _stage(self, batch, stream):
        inputs, input_sizes, targets, target_sizes, utt_list = batch[:5]
        slot = self._slots[self._turn]
        self._turn ^= 1
        if slot["copied"] is not None:
            slot["copied"].synchronize()         # the H2D copies that last read this slot (two batches ago: long done)
        sizes = input_sizes if torch.is_tensor(input_sizes) else torch.as_tensor(np.asarray(input_sizes, dtype=np.float32))
        with torch.cuda.stream(stream):
            dev = [self._pinned(slot, k, t).to(self.device, non_blocking=True) for k, t in enumerate((inputs, sizes, targets, target_sizes))]
            ready = torch.cuda.Event()
            ready.record(stream)
        slot["copied"] = ready
        return (dev[0], dev[1], dev[2], dev[3], utt_list) + tuple(batch[5:]), ready   # (+ the global batch size of a DP shard)

    def __iter__(self):
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher stages batches on a GPU; got device %s" % self.device)
        stream = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        try:
            staged = self._stage(next(it), stream)
        except StopIteration:
            return
        while staged is not None:
            batch, ready = staged
            try:
                staged = self._stage(next(it), stream)      # overlaps with the consumer's step on `batch`
            except StopIteration:
                staged = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in batch[:4]:
                t.record_stream(cur)
            yield batch
