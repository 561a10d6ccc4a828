"""Host side of the path, timed on the host alone (no GPU): the decoders' string assembly and edit distances in the library's host code
against the interpreter expressions they replace, and the input pipeline's throughput on one thread (toy corpus written to a temp dir).
    python tools/host_bench.py [> profiles/r04_host_side.txt]"""
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tm(f, n):
    f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e3


def _rows(a, b):
    """the interpreter form of the edit distance (two rolling rows)"""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def main():
    from ctc_pytorch_amd import ops
    from ctc_pytorch_amd.utils import data_loader as dl
    from ctc_pytorch_amd.utils.ctcDecoder import Decoder
    V = 62
    phones = ["blank", "UNK"] + ["p%d" % i for i in range(V - 2)]          # (2-5 byte words, like the 60 TIMIT phones)
    rs = np.random.RandomState(0)
    print("host threads: 1; python %s; numpy %s" % (sys.version.split()[0], np.__version__))
    for name, lo, hi in (("peaky-like (60-120 tokens per utterance)", 60, 121), ("flat-like (400-720 tokens per utterance)", 400, 721)):
        B, T = 128, 800
        ids = rs.randint(1, V, size=(B, T)).astype(np.int32)
        lens = rs.randint(lo, hi, size=B).astype(np.int32)
        py = lambda: [" ".join(map(phones.__getitem__, ids[b, : lens[b]].tolist())) for b in range(B)]
        nat = lambda: ops.join_tokens(ids, lens, phones, " ")
        assert py() == nat()
        print("string assembly of a 128-utterance batch, %s: interpreter join %.3f ms, ctcn_join_tokens %.3f ms" % (name, tm(py, 20), tm(nat, 100)))
    dec = Decoder({0: "_"}, space_idx=-1)
    for n in (150, 600, 1700):
        a = "".join(rs.choice(list("abcdefgh ")) for _ in range(n))
        b = "".join(rs.choice(list("abcdefgh ")) for _ in range(n - n // 20))
        t0 = time.perf_counter()
        want = _rows(a, b)
        t_py = (time.perf_counter() - t0) * 1e3
        assert dec.cer(a, b) == want
        print("edit distance of two %d-character strings: interpreter rows %.1f ms, ctcn_levenshtein %.3f ms" % (n, t_py, tm(lambda: dec.cer(a, b), 20)))
    with tempfile.TemporaryDirectory() as d:
        N = 256
        mats = {"utt%04d" % i: rs.standard_normal((rs.randint(400, 801), 40)).astype(np.float32) for i in range(N)}
        dl.write_kaldi_ark(os.path.join(d, "f.ark"), os.path.join(d, "f.scp"), mats)
        with open(os.path.join(d, "units"), "w") as f:
            f.write("\n".join(phones[2:]) + "\n")
        with open(os.path.join(d, "lab"), "w") as f:
            for u in mats:
                f.write(u + " " + " ".join(phones[2 + rs.randint(60)] for _ in range(rs.randint(30, 60))) + "\n")
        vocab = dl.Vocab(os.path.join(d, "units"))
        for ctx, what in ((0, "40-d features"), (4, "9-frame splice, 360-d")):
            opts = types.SimpleNamespace(left_ctx=ctx, right_ctx=ctx, n_skip_frame=1, n_downsample=1)
            ds = dl.SpeechDataset(vocab, os.path.join(d, "f.scp"), os.path.join(d, "lab"), opts)
            ld = dl.SpeechDataLoader(ds, batch_size=32, shuffle=False, num_workers=0)
            best = 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                fr = sum(int(b[0].shape[0] * b[0].shape[1]) for b in ld)
                best = max(best, fr / (time.perf_counter() - t0))
            print("input pipeline (ark -> SpeechDataset -> create_input, batch 32, no workers), %s: %.2f M padded frames/s" % (what, best / 1e6))


if __name__ == "__main__":
    main()
