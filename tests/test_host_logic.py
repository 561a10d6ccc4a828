"""CPU tests: the C-ABI library loads and exports every symbol of include/ctcn.h (no compute calls), the host
logic mirrors the reference (decoder strings / scoring, LM table, length conversion, input pipeline, LR schedule),
the torch-CPU counterpart is pinned to the golden vectors, and the data-parallel plumbing works under gloo."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as tnn

from oracle import np_ref as R
from oracle import torch_cpu
from ctc_pytorch_amd.testing import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def test_build_and_symbols():
    import __graft_entry__ as ge
    from ctc_pytorch_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        ge.build()
    hdr = open(os.path.join(ROOT, "include", "ctcn.h")).read()
    declared = set(re.findall(r"\b(ctcn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), "libctcn.so does not export %s" % name
    assert set(_lib.exported_symbols()) == declared, set(_lib.exported_symbols()) ^ declared
    assert L.ctcn_version() >= 100


def test_product_has_no_cpu_fallback():
    from ctc_pytorch_amd import nn, ops
    from ctc_pytorch_amd.models.model_ctc import CTC_Model
    m = CTC_Model(rnn_param={"rnn_input_size": 40, "rnn_hidden_size": 8, "rnn_layers": 1, "rnn_type": nn.LSTM,
                             "bidirectional": True, "batch_norm": True}, num_class=10, drop_out=0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 6, 40))
    with pytest.raises(RuntimeError):
        ops.log_softmax(torch.zeros(3, 4))
    with pytest.raises(ValueError):
        CTC_Model(rnn_param=None)
    # the product never imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ctc_pytorch_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", ""), (dirpath, f)


def test_default_precision_is_the_benchmarked_mode():
    """README / DESIGN / bench.py call bf16x3 (precision 1) the default: the library must agree, and CTCN_PRECISION must be
    honoured by the package itself (not only by bench.py), so that steps/train_ctc.main and the unmodified reference drivers
    train in the mode the headline number is measured in."""
    from ctc_pytorch_amd import ops
    if "CTCN_PRECISION" not in os.environ:
        assert ops.DEFAULT_PRECISION == 1 and ops.get_precision() == 1
    code = "from ctc_pytorch_amd import ops; print(ops.get_precision(), ops.DEFAULT_PRECISION)"
    for env_val, want in (("0", "0 0"), ("1", "1 1")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CTCN_PRECISION=env_val), cwd=ROOT, capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.split("\n")[0].strip() == want, (out.stdout, out.stderr[-500:])
    bad = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CTCN_PRECISION="7"), cwd=ROOT, capture_output=True, text=True)
    assert bad.returncode != 0 and "CTCN_PRECISION" in bad.stderr
    with pytest.raises(ValueError):
        ops.set_precision(2)


def test_state_dict_surface_matches_reference_keys():
    from ctc_pytorch_amd import nn
    from ctc_pytorch_amd.models.model_ctc import CTC_Model
    for tag in ("lstm2x32", "gru2x24", "rnn2x20_uni_nobn", "cnn_lstm2x16", "cnn_pool_lstm2x16", "cnn_bigbank_lstm2x16"):
        z = load("model_" + tag)
        want = {k[len("after."):]: z[k].shape for k in z.files if k.startswith("after.")}
        base = {"rnn_input_size": 121 if "bigbank" in tag else 40, "bidirectional": True, "batch_norm": True, "rnn_layers": 2}
        if tag == "lstm2x32":
            m = CTC_Model(rnn_param=dict(base, rnn_hidden_size=32, rnn_type=nn.LSTM), num_class=62)
        elif tag == "gru2x24":
            m = CTC_Model(rnn_param=dict(base, rnn_hidden_size=24, rnn_type=nn.GRU), num_class=62)
        elif tag == "rnn2x20_uni_nobn":
            m = CTC_Model(rnn_param=dict(base, rnn_hidden_size=20, rnn_type=nn.RNN, bidirectional=False, batch_norm=False), num_class=62)
        else:
            layers = [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]
            if tag == "cnn_pool_lstm2x16":
                layers = [[(1, 8), (3, 3), (1, 2), (1, 1), (2, 1)], [(8, 8), (3, 3), (1, 2), (1, 1), (3, 1)]]
            if tag == "cnn_bigbank_lstm2x16":        # the front-end of the reference's own example model (model_ctc.py:232-233)
                layers = [[(1, 32), (3, 41), (1, 2), (0, 0), None], [(32, 32), (3, 21), (2, 2), (0, 0), None]]
            cp = {"batch_norm": True, "activate_function": nn.ReLU, "layer": layers}
            m = CTC_Model(add_cnn=True, cnn_param=cp, rnn_param=dict(base, rnn_hidden_size=16, rnn_type=nn.LSTM), num_class=62)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert list(got.keys()) == list(want.keys())
        assert all(tuple(want[k]) == got[k] for k in got)
        pkg = CTC_Model.save_package(m, epoch={"n_feats": 40})
        assert set(pkg) == {"rnn_param", "add_cnn", "cnn_param", "num_class", "_drop_out", "state_dict", "epoch"}


def test_dropout_stream_offsets_are_disjoint_across_threads():
    """The Philox offsets of the dropout calls of one process (ops._next_dropout_stream) come from ONE counter: calls racing from several threads
    (two models in one process; the autograd thread next to the main thread) must never be handed overlapping counter ranges -- the same
    mask twice (the Python layer's process-wide state)."""
    import threading
    from ctc_pytorch_amd import ops
    snap = ops._drop_counter[0]
    try:
        ops._drop_counter[0] = 0
        got, sizes = [[] for _ in range(8)], [7, 64, 1001, 4, 13, 250, 3, 96]
        def work(k):
            for i in range(4000):
                n = sizes[(k + i) % 8]
                got[k].append((ops._next_dropout_stream(n)[1], (n + 3) // 4))
        ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        spans = sorted(sp for g in got for sp in g)
        assert all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, none handed out twice
        assert spans[0][0] == 0 and spans[-1][0] + spans[-1][1] == ops._drop_counter[0]
    finally:
        ops._drop_counter[0] = snap


def test_layer_cnn_one_element_kernel_branch_surface():
    """LayerCNN's Conv1d / BatchNorm1d / MaxPool1d branch (reference model_ctc.py:48-50, 54-55): the module tree and the state_dict (keys, order,
    the (Co, Ci, k) weight shape) are the reference's -- tests/golden/layer_cnn1d.npz holds its state_dict -- and the torch-CPU stack of the
    same modules reproduces the captured outputs (the fixture is what torch computes: the GPU test holds the HIP path to it)."""
    from ctc_pytorch_amd import nn
    from ctc_pytorch_amd.models.model_ctc import LayerCNN
    z = load("layer_cnn1d")
    for tag in ("a", "b"):
        cin, cout, k, s, p, pool, bn, L = (int(v) for v in z[tag + ".cfg"])
        layer = LayerCNN(cin, cout, (k,), (s,), (p,), pooling_size=pool, batch_norm=bool(bn), dropout=0.0)
        assert isinstance(layer.conv, nn.Conv1d) and isinstance(layer.pooling, nn.MaxPool1d)
        assert (layer.batch_norm is None) == (not bn) and (bn == 0 or isinstance(layer.batch_norm, nn.BatchNorm1d))
        before = {key[len(tag + ".before."):]: z[key] for key in z.files if key.startswith(tag + ".before.")}
        got = {key: tuple(v.shape) for key, v in layer.state_dict().items()}
        assert list(got.keys()) == list(before.keys()) and all(got[key] == before[key].shape for key in got)
        mods = [tnn.Conv1d(cin, cout, k, s, p)] + ([tnn.BatchNorm1d(cout)] if bn else []) + [tnn.ReLU(), tnn.MaxPool1d(pool)]
        ref = tnn.Sequential(*mods)
        ref[0].load_state_dict({"weight": torch.from_numpy(before["conv.weight"]), "bias": torch.from_numpy(before["conv.bias"])})
        if bn:
            ref[1].load_state_dict({key[len("batch_norm."):]: torch.from_numpy(v) for key, v in before.items() if key.startswith("batch_norm.")})
        ref.train()
        y = ref(torch.from_numpy(z[tag + ".x0"]))
        assert tuple(y.shape) == z[tag + ".y0"].shape and float((y.detach() - torch.from_numpy(z[tag + ".y0"])).abs().max()) < 1e-5
    assert isinstance(LayerCNN(1, 2, (3,), (1,), (0,), pooling_size=None).pooling, nn.MaxPool1d)      # MaxPool1d(None) constructs, as in the reference


@pytest.mark.parametrize("tag,cls,H,bi,bn", [("lstm2x32", tnn.LSTM, 32, True, True), ("gru2x24", tnn.GRU, 24, True, True),
                                             ("rnn2x20_uni_nobn", tnn.RNN, 20, False, False), ("cnn_lstm2x16", tnn.LSTM, 16, True, True),
                                             ("cnn_pool_lstm2x16", tnn.LSTM, 16, True, True), ("cnn_bigbank_lstm2x16", tnn.LSTM, 16, True, True)])
def test_torch_cpu_counterpart_is_pinned(tag, cls, H, bi, bn):
    z = load("model_" + tag)
    rp = {"rnn_input_size": 121 if "bigbank" in tag else 40, "rnn_hidden_size": H, "rnn_layers": 2, "rnn_type": cls, "bidirectional": bi, "batch_norm": bn}
    if tag.startswith("cnn"):
        layers = [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]
        if tag == "cnn_pool_lstm2x16":
            layers = [[(1, 8), (3, 3), (1, 2), (1, 1), (2, 1)], [(8, 8), (3, 3), (1, 2), (1, 1), (3, 1)]]
        if tag == "cnn_bigbank_lstm2x16":
            layers = [[(1, 32), (3, 41), (1, 2), (0, 0), None], [(32, 32), (3, 21), (2, 2), (0, 0), None]]
        cp = {"batch_norm": True, "activate_function": tnn.ReLU, "layer": layers}
        m = torch_cpu.TorchCpuCTCModel(add_cnn=True, cnn_param=cp, rnn_param=rp, num_class=62, drop_out=0.0)
    else:
        m = torch_cpu.TorchCpuCTCModel(rnn_param=rp, num_class=62, drop_out=0.0)
    vals = synth.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=int(z["seed_w"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    x, frac = torch.from_numpy(z["x"]), torch.from_numpy(z["frac"])
    tg, tl = torch.from_numpy(z["targets"]), torch.from_numpy(z["tgt_len"])
    losses = [torch_cpu.train_step(m, opt, x, frac, tg, tl) for _ in range(3)]
    assert np.allclose(losses, z["losses"], rtol=1e-6)
    for k, v in m.state_dict().items():
        assert np.allclose(v.numpy(), z["after." + k], atol=1e-6), k


def test_torch_cpu_counterpart_is_pinned_at_the_shipped_yaml_shape():
    """The reference's shipped configuration (timit/conf/ctc_config.yaml:11-40: 243-d spliced input, 2-layer CNN -> 1 952-wide RNN
    input, 4 x 384 BiLSTM): the torch-CPU counterpart against the fixture captured from the reference (model_ref_yaml.npz: log-probs,
    three losses, norms + strided samples of every gradient and every updated parameter)."""
    z = load("model_ref_yaml")
    rp = {"rnn_input_size": 243, "rnn_hidden_size": 384, "rnn_layers": 4, "rnn_type": tnn.LSTM, "bidirectional": True, "batch_norm": True}
    cp = {"batch_norm": True, "activate_function": tnn.ReLU, "layer": [[(1, 32), (3, 3), (1, 2), (1, 1), None], [(32, 32), (3, 3), (2, 2), (1, 1), None]]}
    m = torch_cpu.TorchCpuCTCModel(add_cnn=True, cnn_param=cp, rnn_param=rp, num_class=41, drop_out=0.0)
    vals = synth.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed=int(z["seed_w"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    x, frac = torch.from_numpy(z["x"]), torch.from_numpy(z["frac"])
    tg, tl = torch.from_numpy(z["targets"]), torch.from_numpy(z["tgt_len"])
    stride = int(z["sample_stride"])
    lp = m(x)
    assert tuple(lp.shape) == tuple(z["lp"].shape) and np.allclose(lp.detach().numpy(), z["lp"], atol=1e-5)
    in_len = (frac * lp.size(0)).long()
    assert np.array_equal(in_len.numpy(), z["in_len"])
    loss = tnn.CTCLoss(reduction="sum")(lp, tg, in_len, tl) / x.shape[0]
    opt.zero_grad()
    loss.backward()
    for k, p in m.named_parameters():
        assert np.allclose(p.grad.reshape(-1)[::stride].numpy(), z["gsample." + k], atol=1e-6, rtol=1e-4), k
    opt.step()
    losses = [float(loss)] + [torch_cpu.train_step(m, opt, x, frac, tg, tl) for _ in range(2)]
    assert np.allclose(losses, z["losses"], rtol=1e-5), (losses, z["losses"])


def test_bench_self_launches_under_torchrun_for_n_gpus():
    """VERDICT r2 #1a: `python bench.py --gpus N` (no launcher, the form the driver uses for N = 1) must become the torch.distributed.run
    command line for N > 1 instead of asserting; under a launcher (WORLD_SIZE set) and for N = 1 it runs in place."""
    import argparse
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ns = argparse.Namespace(mode="train", gpus=8)
    argv, env = bench.self_launch_argv(ns, ["--gpus", "8", "--steps", "5", "--warmup", "2"], {"PATH": "/usr/bin"})
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(argv[argv.index("--master-port") + 1]) < 65536
    assert argv[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and argv[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.self_launch_argv(ns, [], {"WORLD_SIZE": "8"}) is None                       # already under a launcher
    assert bench.self_launch_argv(argparse.Namespace(mode="train", gpus=1), [], {}) is None
    assert bench.self_launch_argv(argparse.Namespace(mode="decode", gpus=4), [], {}) is None


def test_overlap_decision_is_rank_invariant_only_for_even_shards():
    """the early all-reduce of a gradient slice may only be issued when every rank takes the same decision, i.e.
    when the global minibatch splits evenly (parallel.overlap_is_rank_invariant)."""
    from ctc_pytorch_amd import parallel
    try:
        parallel.set_batch_split(None, None)
        assert parallel.overlap_is_rank_invariant()
        parallel.set_batch_split(32, 32)
        assert parallel.overlap_is_rank_invariant()                                            # single process: world size 1
        import unittest.mock as mock
        with mock.patch.object(parallel, "world_size", lambda: 2):
            parallel.set_batch_split(64, 32)
            assert parallel.overlap_is_rank_invariant()
            parallel.set_batch_split(33, 17)
            assert not parallel.overlap_is_rank_invariant()
    finally:
        parallel.set_batch_split(None, None)


def test_length_conversion_matches_reference_table():
    from ctc_pytorch_amd.steps.train_ctc import frames_from_fraction
    rows = load("length_table")["rows"]
    sel = rows[(rows[:, 1] == 800) & (rows[:, 2] == 400)]
    frac = torch.tensor([float(n) / 800 for n in sel[:, 0]], dtype=torch.float32)
    assert np.array_equal(frames_from_fraction(frac, 400), sel[:, 3])
    assert (sel[:, 3] < sel[:, 0] * 400 // 800).any()          # the float32 under-count cases exist and are reproduced


def test_decoder_host_logic_and_scoring():
    from ctc_pytorch_amd.utils.ctcDecoder import Decoder
    meta = json.load(open(os.path.join(G, "decoders.json")))
    assert Decoder("abcde", 1, 2)._convert_to_strings([[1, 2, 1, 0, 3], [1, 2, 1, 1, 1]]) == meta["kat_convert"]
    d = Decoder(synth.int2char(62), space_idx=-1, blank_index=0)
    for key in ("greedy_peaky", "beam_peaky_W5_a0.1"):
        sc = meta["score_" + key]
        assert sum(d.cer(a, b) for a, b in zip(meta[key], meta["labels"])) == sc["total_cer"]
        assert sum(d.wer(a, b) for a, b in zip(meta[key], meta["labels"])) == sc["total_wer"]
    assert d._edit_distance("", "abc") == 3 and d._edit_distance("abc", "") == 3 and d._edit_distance("kitten", "sitting") == 3
    assert d._process_string(["blank", "a", "a", "blank", "a", "b"], remove_rep=True) == " a a b"


def test_lm_table_matches_reference():
    from ctc_pytorch_amd.utils.NgramLM import LanguageModel
    i2c = synth.int2char(62)
    lm = LanguageModel(os.path.join(G, "lm_phone_bg.arpa"))
    tab = lm.table([i2c[i] for i in range(62)])
    want = load("lm_table")["lm_table"]
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(tab), np.isnan(want)) and np.array_equal(tab[ok], want[ok])
    with pytest.raises(KeyError):
        lm.get_bi_prob("aa", "not-a-phone")
    with pytest.raises(TypeError):
        LanguageModel(None)


def test_input_pipeline_contract(tmp_path):
    from ctc_pytorch_amd.utils import data_loader as dl
    rs = np.random.RandomState(0)
    mats = {"utt%d" % i: rs.standard_normal((n, 40)).astype(np.float32) for i, n in enumerate((53, 100, 77))}
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    dl.write_kaldi_ark(ark, scp, mats)
    for line in open(scp):
        utt, p = line.split()
        assert np.array_equal(dl.read_kaldi_matrix(p), mats[utt])
    units = tmp_path / "units"
    units.write_text("\n".join(synth.TIMIT_60) + "\n")
    lab = tmp_path / "phn_text"
    lab.write_text("utt0 aa b zz\nutt1 sh iy\nutt2 k ae t s\n")
    vocab = dl.Vocab(str(units))
    assert vocab.n_words == 62 and vocab.index2word[0] == "blank" and vocab.index2word[1] == "UNK"

    class O:
        left_ctx, right_ctx, n_skip_frame, n_downsample = 0, 2, 2, 2
    ds = dl.SpeechDataset(vocab, scp, str(lab), O)
    f0, l0, u0 = ds[0]
    assert f0.shape == (28, 120) and l0.tolist() == [vocab.word2index["aa"], vocab.word2index["b"], 1]   # 53 ->27 -> pad to 28; zz -> UNK
    # make_context / skip_feat against a direct restatement of tools.py:66-86
    m = mats["utt0"]
    ctx = np.hstack([m, np.vstack([m[1:], m[-1:]]), np.vstack([m[2:], m[-1:], m[-1:]])])
    assert np.array_equal(f0[:27].numpy(), ctx[::2])
    x, frac, tg, tl, utts = dl.create_input([ds[i] for i in range(3)])
    assert x.shape == (3, 50, 120) and x.dtype == torch.float32 and frac.dtype == torch.float32 and tg.dtype == torch.int64
    assert frac.tolist() == [np.float32(28 / 50), 1.0, np.float32(40 / 50)] and tl.tolist() == [3, 2, 4]
    assert float(x[0, 28:].abs().sum()) == 0 and tg[1, 2:].tolist() == [0, 0]
    # the padded batch is written element by element (no zero-fill first): dirty the allocator's free blocks and look at every padding frame
    for _ in range(3):
        junk = torch.full((3, 50, 120), float("nan"))
        del junk
        x2 = dl.create_input([ds[i] for i in range(3)])[0]
        assert torch.equal(x2, x) and not torch.isnan(x2).any()
    # left context + both edges against the gather form, as a window view and as a copy; double-precision matrices; broken headers
    m5 = mats["utt2"][:5]
    idx = np.arange(5)
    want = np.hstack([m5[np.clip(idx + d, 0, 4)] for d in range(-3, 2)])
    assert np.array_equal(dl.make_context(m5, 3, 1), want) and np.array_equal(dl.make_context(m5, 3, 1, as_view=True), want)
    assert dl.make_context(m5[:0], 3, 1).shape == (0, 200)
    dpath = str(tmp_path / "d.ark")
    with open(dpath, "wb") as f:
        f.write(b"u \0BDM \4" + np.int32(2).tobytes() + b"\4" + np.int32(3).tobytes() + np.arange(6, dtype="<f8").tobytes())
    got = dl.read_kaldi_matrix(dpath + ":2")
    assert got.dtype == np.float32 and np.array_equal(got, np.arange(6, dtype=np.float32).reshape(2, 3))
    with open(dpath, "r+b") as f:
        f.truncate(30)
    with pytest.raises(ValueError):
        dl.read_kaldi_matrix(dpath + ":2")
    with pytest.raises(ValueError):
        dl.read_kaldi_matrix(dpath + ":3")
    cpath = str(tmp_path / "c.ark")
    with open(cpath, "wb") as f:
        f.write(b"\0BCM " + b"\0" * 32)
    with pytest.raises(NotImplementedError):
        dl.read_kaldi_matrix(cpath)


def test_lr_controller_schedule():
    from ctc_pytorch_amd.steps.train_ctc import LRController

    class M:
        def __init__(self): self.v = 0
        def state_dict(self): return {"v": self.v}
        def load_state_dict(self, s): self.v = s["v"]

    class Opt(M):
        def __init__(self):
            super().__init__()
            self.param_groups = [{"lr": 1.0}]
    m, o = M(), Opt()
    c = LRController(end_adjust_acc=2, decay=0.5)
    c.end_epoch(m, o, 0.1, 100.0)            # first epoch: new best
    assert c.loss_best == 100.0 and c.adjust_rate_count == 0
    m.v = 1
    c.end_epoch(m, o, 0.2, 99.0)             # within +-delta: count 1, better true best -> snapshot
    assert c.adjust_rate_count == 1 and c.loss_best_true == 99.0 and c.model_state == {"v": 1}
    m.v = 2
    c.end_epoch(m, o, 0.15, 150.0)           # much worse: immediate decay + rollback to the snapshot
    assert c.adjust_rate_flag and c.adjust_time == 1 and m.v == 1 and c.loss_best == 99.0
    c.begin_epoch(o)
    assert o.param_groups[0]["lr"] == 0.5 and not c.adjust_rate_flag
    for _ in range(7):
        c.end_epoch(m, o, 0.1, 500.0)
    assert c.stop and c.adjust_time == 8


_DP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from ctc_pytorch_amd import parallel
rank, world, local = parallel.init_from_env(backend="gloo")
assert world == 2 and parallel.world_size() == 2
lo, hi = parallel.shard_range(7, rank, world)
assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
g = torch.full((10,), float(rank + 1))
parallel.allreduce_grads(g)
assert torch.equal(g, torch.full((10,), 3.0))
p = torch.arange(5.0) * (rank + 1)
parallel.broadcast_params(p)
assert torch.equal(p, torch.arange(5.0))
assert parallel.max_over_ranks(1.5 + rank, torch.device("cpu")) == 2.5
assert parallel.gather_over_ranks(1.5 + rank, torch.device("cpu")) == [1.5, 2.5]
# synchronised BatchNorm: the per-channel fp64 sums of the two shards are added, the count doubles
sums = torch.full((6, 2), float(rank + 1), dtype=torch.float64)
assert parallel._sync_bn_reduce(sums, 100) == 200.0 and torch.equal(sums, torch.full((6, 2), 3.0, dtype=torch.float64))
# uneven shards (7 utterances over 2 ranks, padded to the same global T_max): the global count comes from the batch split
parallel.set_batch_split(7, hi - lo)
assert parallel._sync_bn_reduce(torch.zeros((6, 2), dtype=torch.float64), 50 * (hi - lo)) == 350.0
parallel.set_batch_split(None, None)
# ShardedBatches: the same global batches on every rank, contiguous shards, global padding and fractions kept, short batch dropped
from ctc_pytorch_amd.utils.data_loader import create_input
import numpy as np
rs = np.random.RandomState(3)
items = [(torch.from_numpy(rs.standard_normal((int(rs.randint(20, 60)), 40)).astype(np.float32)),
          torch.from_numpy(rs.randint(1, 30, size=int(rs.randint(3, 9))).astype(np.int64)), "utt%%d" %% i) for i in range(8)]
batches = [create_input(items[0:5]), create_input(items[5:8]), create_input(items[7:8])]
logged = []
got = list(parallel.ShardedBatches(batches, rank, world, log=logged.append))
assert len(got) == 2 and (len(logged) == 1) == (rank == 0)
for g, w in zip(got, batches):
    n = w[0].shape[0]
    a, b = parallel.shard_range(n, rank, world)
    assert g[5] == n and g[0].shape[1] == w[0].shape[1] and g[2].shape[1] == w[2].shape[1]       # global T_max / L_max
    assert torch.equal(g[0], w[0][a:b]) and torch.equal(g[1], w[1][a:b]) and torch.equal(g[2], w[2][a:b]) and torch.equal(g[3], w[3][a:b])
    assert g[4] == w[4][a:b]
st = parallel.allreduce_stats(torch.tensor([1.0 + rank, 10.0], dtype=torch.float64))
assert torch.equal(st, torch.tensor([3.0, 20.0], dtype=torch.float64))
# overlapped gradient all-reduce: two slices are reduced "during the backward pass", the step-end call reduces the rest -- every
# element of the flat buffer exactly once
parallel.enable_overlap(True)
from ctc_pytorch_amd import ops
assert ops._grad_ready["hook"] is parallel._slice_ready
flat = torch.arange(100, dtype=torch.float32) * (rank + 1)
parallel._slice_ready([flat[10:20].view(2, 5), flat[20:30].view(5, 2)])          # adjacent views = one contiguous slice
parallel._slice_ready([flat[60:70], flat[80:90]])                                  # not contiguous: left to the end
parallel._slice_ready([flat[95:100]])
assert len(parallel._overlap["works"]) == 2
parallel.allreduce_grads(flat)
assert torch.equal(flat, torch.arange(100, dtype=torch.float32) * 3) and parallel._overlap["works"] == [] and parallel._overlap["done"] == []
parallel.allreduce_grads(flat)                                                      # nothing pending: one plain all-reduce
assert torch.equal(flat, torch.arange(100, dtype=torch.float32) * 6)
parallel.enable_overlap(False)
assert ops._grad_ready["hook"] is None
parallel.enable_sync_bn(True)
assert ops._sync_bn["reduce"] is parallel._sync_bn_reduce
parallel.enable_sync_bn(False)
assert ops._sync_bn["reduce"] is None
# uneven shards: no rank may issue the early slice all-reduce (ADVICE r2) -- the whole buffer goes with the step-end call, once
parallel.enable_overlap(True)
parallel.set_batch_split(33, 17 if rank == 0 else 16)
assert not parallel.overlap_is_rank_invariant()
flat = torch.arange(50, dtype=torch.float32) * (rank + 1)
if rank == 0:
    parallel._slice_ready([flat[10:20]])             # what ops.py would do on the rank whose shard crosses the side-stream threshold
assert parallel._overlap["works"] == [] and parallel._overlap["done"] == []
parallel.allreduce_grads(flat)
assert torch.equal(flat, torch.arange(50, dtype=torch.float32) * 3)
parallel.set_batch_split(32, 16)
assert parallel.overlap_is_rank_invariant()
parallel.set_batch_split(None, None)
parallel.enable_overlap(False)
# per-shard BatchNorm: the running statistics are averaged over the ranks at the end of a training epoch (parallel.sync_bn_buffers)
import torch.nn as tnn
bn = tnn.Sequential(tnn.BatchNorm1d(3), tnn.Linear(3, 2))
with torch.no_grad():
    bn[0].running_mean.fill_(1.0 + rank); bn[0].running_var.fill_(2.0 + 2 * rank); bn[0].num_batches_tracked.fill_(5)
parallel._batch["seen"] = 0                            # (the splits set above counted utterances: equal weights for this one)
assert parallel.sync_bn_buffers(bn) == 2
# pooled like two populations: mean 1.5, variance = E[var] + Var[mean] = 3 + 0.25 (round 4 dropped the second term)
assert torch.equal(bn[0].running_mean, torch.full((3,), 1.5)) and torch.equal(bn[0].running_var, torch.full((3,), 3.25)) and int(bn[0].num_batches_tracked) == 5
# uneven shards: weighted by the utterances each rank saw since the last merge (set_batch_split counts them)
with torch.no_grad():
    bn[0].running_mean.fill_(1.0 + rank); bn[0].running_var.fill_(2.0 + 2 * rank)
parallel.set_batch_split(4, 3 if rank == 0 else 1)
parallel.set_batch_split(None, None)
assert parallel.sync_bn_buffers(bn) == 2
assert torch.allclose(bn[0].running_mean, torch.full((3,), 1.25)) and torch.allclose(bn[0].running_var, torch.full((3,), 2.5 + 0.1875))
# against the single-process buffers: shards that differ systematically (momentum 1: the buffers are the last batch's statistics)
gen = torch.Generator().manual_seed(11)
data = torch.randn(2, 4000, 3, generator=gen) * torch.tensor([1.0, 2.0, 0.5]) + torch.tensor([[[0.0, 1.0, -2.0]], [[3.0, 1.5, -2.5]]])
one = tnn.BatchNorm1d(3, momentum=1.0).train(); one(data.reshape(-1, 3))
mine = tnn.BatchNorm1d(3, momentum=1.0).train(); mine(data[rank])
assert parallel.sync_bn_buffers(mine) == 2
assert torch.allclose(mine.running_mean, one.running_mean, atol=1e-5) and torch.allclose(mine.running_var, one.running_var, rtol=1e-3)
avg_only = 0.5 * (data[0].var(0) + data[1].var(0))
assert (one.running_var - avg_only).abs().max() > 0.05          # (what the rank average of the variances alone misses here)
parallel.enable_sync_bn(True)
assert parallel.sync_bn_buffers(bn) == 0              # global statistics already: nothing to do
parallel.enable_sync_bn(False)
# replicas-only decode (SURVEY 8e "Decode", steps/test_ctc.decode_and_score_sharded): rank r scores minibatches r, r + world, ...; the four
# totals are summed over the ranks -- same CER / WER and counts as one process over all minibatches (a host-side stand-in for the model and
# the decoder: the class logic, the scoring and the reduction are what is under test here)
from ctc_pytorch_amd.steps.test_ctc import decode_and_score, decode_and_score_sharded
from ctc_pytorch_amd.utils.ctcDecoder import Decoder
class _Model(torch.nn.Module):
    def forward(self, x):
        return x.transpose(0, 1).contiguous()                      # (B, T, V) -> (T, B, V) "log-probs"
class _Dec(Decoder):
    def decode(self, probs, lens):
        out = []
        for b in range(probs.shape[1]):
            path = probs[: int(lens[b]), b].argmax(-1).tolist()
            keep = [k for i, k in enumerate(path) if k != 0 and (i == 0 or k != path[i - 1])]
            out.append(" ".join(self.int_to_char[k] for k in keep))
        return out
i2w = {k: "w%%d" %% k for k in range(12)}
rs = np.random.RandomState(9)
loader = []
for nb in (3, 4, 2, 5, 1):
    T = int(rs.randint(8, 15))
    x = torch.from_numpy(rs.standard_normal((nb, T, 12)).astype(np.float32))
    frac = torch.from_numpy(rs.randint(T // 2, T + 1, size=nb).astype(np.float32) / T)
    tl = torch.from_numpy(rs.randint(1, 6, size=nb).astype(np.int64))
    tg = torch.from_numpy(rs.randint(1, 12, size=(nb, 5)).astype(np.int64))
    loader.append((x, frac, tg, tl, ["u"] * nb))
one = _Dec(i2w, space_idx=-1, blank_index=0)
cer1, wer1 = decode_and_score(_Model(), loader, one, i2w, "cpu", log=lambda *_: None)
two = _Dec(i2w, space_idx=-1, blank_index=0)
logged = []
cer2, wer2 = decode_and_score_sharded(_Model(), loader, two, i2w, "cpu", rank=rank, world=world, log=logged.append)
assert abs(cer1 - cer2) < 1e-9 and abs(wer1 - wer2) < 1e-9 and (two.num_char, two.num_word) == (one.num_char, one.num_word), (cer1, cer2, wer1, wer2)
assert (len(logged) == 2) == (rank == 0)
dist.barrier()
print("rank", rank, "ok")
"""


_DP_FAIL_SCRIPT = r"""
import json, os, sys, time, types, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
from ctc_pytorch_amd import parallel
mode = sys.argv[1]
rank, world, local = parallel.init_from_env(backend="gloo")
args = types.SimpleNamespace(gpus=world, steps=20, warmup=5, scaling="weak", workload="cfg2")
mon = parallel.RankMonitor(rank, world, deadline_s=float(os.environ.get("CTCN_BENCH_DEADLINE_S", "900")), poll_s=0.2,
                           on_trouble=lambda rep: print(json.dumps(bench.trouble_line(args, rep, None)), flush=True))
mon.progress("initialised", backend=dist.get_backend())
try:
    g = torch.ones(1000)
    for step in range(50):
        mon.progress("prewarm", step=step)
        if mode == "raise" and rank == 1 and step == 3:
            raise RuntimeError("injected failure of rank 1 at step 3")
        if mode == "hang" and rank == 1 and step == 3:
            time.sleep(600)                     # a rank that neither fails nor finishes: only the deadline can end this
        dist.all_reduce(g)                      # rank 0 blocks here from step 3 on: its peer never arrives
    mon.finish(status=0)
except BaseException as e:
    mon.fail(e, kernels=["rnn_fwd_tagged", "rnn_bwd_scatter"])
    sys.exit(3)
mon.close()
print("finished without trouble")
"""


@pytest.mark.parametrize("mode", ["raise", "hang"])
def test_a_failing_or_hanging_rank_still_yields_the_bench_line(tmp_path, mode):
    """the first N > 1 contact must leave evidence.  Two ranks over gloo; rank 1 raises at step 3 (or simply stops
    answering) while rank 0 sits in the all-reduce of that step.  parallel.RankMonitor -- the net bench.py runs under for N > 1 -- must
    get rank 0's JSON line out within 150 s (here: seconds): `error` naming the rank (or the deadline), every rank's last phase, the
    failing rank's message and recurrence kernels; and both processes must end with a non-zero code instead of hanging."""
    import time
    script = tmp_path / "dp_fail.py"
    script.write_text(_DP_FAIL_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613" if mode == "raise" else "29615", WORLD_SIZE="2", CTCN_DIST_TIMEOUT_S="120",
               CTCN_BENCH_DEADLINE_S="900" if mode == "raise" else "8")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script), mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    try:
        out0 = procs[0].communicate(timeout=150)[0].decode()
    finally:
        if mode == "hang":
            procs[1].kill()                     # (the sleeping rank is the launcher's to reap once rank 0 has spoken)
    procs[1].communicate(timeout=60)
    assert time.time() - t0 < 150
    line = json.loads(out0.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and line["scaling"] == "weak" and "CTCN_DP_SAFE" in line["hint"]
    assert procs[0].returncode == 3
    ranks = line["ranks"]
    assert ranks["0"]["progress"]["phase"] == "prewarm" and ranks["0"]["progress"]["step"] == 3 and "done" not in ranks["0"]
    if mode == "raise":
        assert "rank 1 failed" in line["error"] and "injected failure" in line["error"]
        assert "injected failure" in ranks["1"]["failed"]["error"] and ranks["1"]["failed"]["kernels"] == ["rnn_fwd_tagged", "rnn_bwd_scatter"]
        assert procs[1].returncode == 3
    else:
        assert "deadline" in line["error"] and "failed" not in ranks["1"] and ranks["1"]["progress"]["step"] == 3


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_flat_adam_placement_puts_the_two_input_projections_side_by_side():
    """optim.placement_order: model order, except weight_ih of both directions adjacent (ctcn_rnn_fwd / bwd then take
    [W_ih_fwd ; W_ih_rev] as one matrix); every index exactly once; modules keep their relative order."""
    from ctc_pytorch_amd.optim import placement_order
    names = ["conv.0.conv.weight", "conv.0.conv.bias", "conv.0.batch_norm.weight", "conv.0.batch_norm.bias",
             "rnns.0.rnn.weight_ih_l0", "rnns.0.rnn.weight_hh_l0", "rnns.0.rnn.weight_ih_l0_reverse", "rnns.0.rnn.weight_hh_l0_reverse",
             "rnns.1.batch_norm.weight", "rnns.1.batch_norm.bias",
             "rnns.1.rnn.weight_ih_l0", "rnns.1.rnn.weight_hh_l0", "rnns.1.rnn.weight_ih_l0_reverse", "rnns.1.rnn.weight_hh_l0_reverse",
             "fc.0.weight", "fc.0.bias", "fc.1.weight"]
    order = placement_order(names)
    assert sorted(order) == list(range(len(names)))
    placed = [names[i] for i in order]
    for l in (0, 1):
        i = placed.index("rnns.%d.rnn.weight_ih_l0" % l)
        assert placed[i:i + 4] == ["rnns.%d.rnn.%s" % (l, k) for k in ("weight_ih_l0", "weight_ih_l0_reverse", "weight_hh_l0", "weight_hh_l0_reverse")]
    rest = [n for n in placed if ".rnn." not in n]
    assert rest == [n for n in names if ".rnn." not in n]
    uni = ["rnns.0.rnn.weight_ih_l0", "rnns.0.rnn.weight_hh_l0", "fc.weight"]                # unidirectional: nothing moves
    assert placement_order(uni) == [0, 1, 2]


def test_hand_waited_reserve_loads_are_not_touched_before_their_wait():
    """rnn_bwd_scatter issues its reserve loads as inline asm and waits for them by hand (`s_waitcnt vmcnt(7)`), which hipcc
    cannot see: tools/check_untracked_loads.py compiles rnn.hip to assembly (no GPU needed) and fails if any instruction
    reads or overwrites such a destination register between the load and the wait."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_untracked_loads.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("findings 0") >= 6


REF = "/root/reference/timit"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_reference_drivers_resolve_to_the_hip_classes():
    """The drop-in boundary as a regression test: execute the TEXT of the reference's
    steps/train_ctc.py and steps/test_ctc.py (everything above their __main__ guards: imports, supported_rnn /
    supported_activate tables, argparse set-up, function definitions) with ctc_pytorch_amd/ first on sys.path, and check
    that every name the drivers use resolves to the HIP-backed class.  Runs in a subprocess: the drivers import top-level
    packages called `models`, `utils`, `steps`."""
    code = r"""
import os, sys, types
sys.path.insert(0, os.path.join(%r, "ctc_pytorch_amd"))
os.chdir(%r)                                      # the drivers do sys.path.append('./'); it must not find the reference tree
import ctc_pytorch_amd.nn as hnn
from ctc_pytorch_amd.models import model_ctc as hmodel
from ctc_pytorch_amd.utils import ctcDecoder as hdec, data_loader as hdl
def same_def(f, g):            # the drivers import the package files under their top-level names: same source, second module object
    return os.path.samefile(f.__code__.co_filename, g.__code__.co_filename) and f.__code__.co_firstlineno == g.__code__.co_firstlineno
for drv in ("train_ctc.py", "test_ctc.py"):
    ns = {"__name__": "reference_driver", "__file__": drv}
    exec(compile(open(os.path.join(%r, "steps", drv)).read(), drv, "exec"), ns)
    assert ns["nn"] is hmodel.nn is hnn, (drv, ns["nn"])
    assert ns["CTC_Model"].__module__.endswith("models.model_ctc") and same_def(ns["CTC_Model"].forward, hmodel.CTC_Model.forward)
    assert ns["nn"].CTCLoss is hnn.CTCLoss and ns["nn"].LSTM is hnn.LSTM
    assert ns["Vocab"].__module__.endswith("utils.data_loader") and same_def(ns["SpeechDataLoader"].__init__, hdl.SpeechDataLoader.__init__)
    if drv == "train_ctc.py":
        assert ns["supported_rnn"] == {"nn.LSTM": hnn.LSTM, "nn.GRU": hnn.GRU, "nn.RNN": hnn.RNN}
        assert ns["supported_activate"]["relu"] is hnn.ReLU
        assert callable(ns["run_epoch"]) and callable(ns["main"])
        loss_fn = ns["nn"].CTCLoss(reduction="sum")                      # train_ctc.py:144
        assert type(loss_fn) is hnn.CTCLoss
        m = ns["CTC_Model"](rnn_param={"rnn_input_size": 40, "rnn_hidden_size": 8, "rnn_layers": 2, "rnn_type": ns["supported_rnn"]["nn.LSTM"],
                                       "bidirectional": True, "batch_norm": True}, num_class=10, drop_out=0.1)
        assert type(m.rnns[0].rnn) is hnn.LSTM and type(m.rnns[1].batch_norm) is hnn.BatchNorm1d and type(m.fc[1]) is hnn.Linear
    else:
        assert same_def(ns["GreedyDecoder"].decode, hdec.GreedyDecoder.decode) and same_def(ns["BeamDecoder"].decode, hdec.BeamDecoder.decode)
        assert ns["Config"].__module__.endswith("steps.train_ctc") and callable(ns["test"])
print("BOUNDARY_OK")
""" % (ROOT, ROOT, REF)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and "BOUNDARY_OK" in out.stdout, out.stderr[-3000:]


def test_decode_driver_picks_the_decoder_from_the_yaml_keys(tmp_path):
    """steps/test_ctc.make_decoder: decode_type / beam_width / lm_alpha / lm_path as the reference's test() reads them
    (test_ctc.py:47-51,64-67)."""
    from ctc_pytorch_amd.steps import test_ctc as TE
    from ctc_pytorch_amd.utils.ctcDecoder import BeamDecoder, GreedyDecoder
    i2c = synth.int2char(62)

    class O:
        decode_type, beam_width, lm_alpha, lm_path = "Greedy", 7, 0.25, os.path.join(G, "lm_phone_bg.arpa")
    assert type(TE.make_decoder(O, i2c)) is GreedyDecoder
    O.decode_type = "Beam"
    d = TE.make_decoder(O, i2c)
    assert type(d) is BeamDecoder and d._decoder.beamWidth == 7 and d._decoder.lm_alpha == 0.25
    with pytest.raises(RuntimeError, match="use_gpu"):
        TE.main({"use_gpu": False})




def test_side_stream_capacity_rule_on_the_measured_shapes():
    """ops.side_stream_fits (round 4): the weight-gradient side stream is used where the idle XCDs can digest a layer's GEMMs within the
    recurrence beside them and the recurrence leaves CUs free on its own XCDs -- the decisions at the shapes whose A/B runs set the constants
    (8 XCDs x 32 CUs; LSTM = cell 0, GRU = cell 1)."""
    from ctc_pytorch_amd import ops
    fits = lambda cell, T, B, I, H: ops.side_stream_fits(cell, T, B, I, H, 2, 8, 256)
    assert fits(0, 800, 32, 640, 320)            # cfg2: 4 idle XCDs, 13.2 ms with / 15.0 without
    assert fits(0, 800, 16, 640, 320)            # one batch tile: 6 idle XCDs
    assert fits(0, 800, 24, 640, 320)
    assert not fits(0, 800, 48, 640, 320)        # 3 batch tiles: 2 idle XCDs cannot keep up (17.9 with / 17.2 without)
    assert not fits(0, 800, 40, 640, 320)
    assert not fits(0, 800, 64, 640, 320)        # no idle XCD at all
    assert fits(0, 200, 8, 768, 384)             # the shipped YAML shape: 4.69 -> 4.36 ms once its weight GEMMs left the main stream
    assert fits(0, 300, 8, 256, 128)             # cfg1
    assert fits(0, 800, 32, 768, 384)            # 27 of 32 CUs per recurrence XCD: still room to start and leave
    assert not fits(1, 1200, 32, 1024, 512)      # H = 512: every CU of the recurrence XCDs taken (41.2 with / 40.2 without)
    assert not fits(1, 1200, 64, 1024, 512)      # cfg4
    assert not ops.side_stream_fits(0, 800, 32, 640, 320, 2, 1, 256)     # a device without XCDs to split
    plan = lambda cell, T, B, I, H: ops.side_stream_plan(cell, T, B, I, H, 2, 8, 256)
    # the idle XCDs' mask: group g of a recurrence that leaves XCDs idle runs on physical XCD _XCD_ORDERS[option "xcd_interleave"][g] -- by
    # default (1) the even XCDs first, so that every XCD pair (2k, 2k + 1) hosts one recurrence XCD and one XCD of side-stream GEMMs (round 5:
    # cfg2 13.19 -> 13.06 ms per step; both GEMM XCDs of a pair next to each other 13.28; include/ctcn.h)
    assert ops.get_option("xcd_interleave") == 1
    assert plan(0, 800, 32, 640, 320) == 0xAA and plan(0, 800, 16, 640, 320) == 0xFA
    ops.set_option("xcd_interleave", 0)
    try:
        assert plan(0, 800, 32, 640, 320) == 0xF0 and plan(0, 800, 16, 640, 320) == 0xFC
    finally:
        ops.set_option("xcd_interleave", 1)
    assert plan(0, 800, 64, 512, 256) == 0xFF                            # no idle XCD, 14 free CUs per XCD: everywhere (14.97 | 16.36)
    assert plan(0, 800, 64, 256, 128) == 0xFF                            # (11.39 | 11.92)
    assert plan(0, 800, 64, 640, 320) == 0 and plan(1, 1200, 64, 1024, 512) == 0


def test_projection_pipeline_plan_on_the_measured_shapes():
    """ctcn_diag_pipeline_chunks = the pure plan behind ctcn_rnn_fwd_ex's projection pipeline (rnn.hip, round 4): chunk counts at the shapes
    whose A/B runs set its rules (8 XCDs x 32 CUs; `allow` = the XCDs a bidirectional recurrence of B rows leaves idle)."""
    from ctc_pytorch_amd import _lib
    L = _lib.lib()
    def chunks(cell, T, B, I, H):
        groups = 2 * ((B + 15) // 16)
        allow = 0xFF & ~((1 << groups) - 1)
        return L.ctcn_diag_pipeline_chunks(cell, T, B, I, H, 2, 8, 256, allow)
    assert chunks(0, 800, 32, 640, 320) == 10        # cfg2: 10 chunks of 2 560 rows, 13.21 | 13.64 ms per step with | without
    assert chunks(0, 1000, 32, 640, 320) == 12       # ragged chunks of 2 688 rows: 16.69 | 17.28
    assert chunks(0, 600, 32, 640, 320) == 8
    assert chunks(0, 700, 32, 640, 320) == 8
    assert chunks(0, 400, 32, 640, 320) == 0         # cfg3's recurrent length: 50 steps per chunk lost 7.5 %
    assert chunks(0, 800, 40, 640, 320) == 0         # two idle XCDs: 24 chunks of 34 steps had passed the one-round test and lost 14 %
    assert chunks(0, 800, 48, 640, 320) == 0
    assert chunks(0, 800, 24, 640, 320) == 8
    assert chunks(0, 800, 16, 640, 320) == 0         # 70 tiles on 192 side CUs: below 3/4 of a round
    assert chunks(0, 800, 32, 768, 384) == 0         # H = 384: a pair's flops outrun the idle XCDs (-2.7 %) ...
    assert chunks(0, 800, 32, 40, 384) == 0          # ... and five free CUs per recurrence XCD are too few even for its bottom layer (-4 %)
    assert chunks(0, 800, 32, 40, 320) == 10         # cfg2's bottom layer (ten free CUs): pipelined, +0.6 %
    assert chunks(0, 800, 32, 512, 256) == 8          # H = 256: 12.05 | 12.61
    assert chunks(1, 1200, 64, 1024, 512) == 0       # cfg4: no idle XCD
    assert chunks(1, 1200, 32, 1024, 512) == 0       # H = 512: every CU of the recurrence XCDs taken
    assert chunks(0, 200, 8, 768, 384) == 0          # the shipped YAML shape
    assert L.ctcn_diag_pipeline_chunks(0, 800, 32, 640, 320, 2, 1, 256, 0) == 0 and L.ctcn_diag_pipeline_chunks(7, 800, 32, 640, 320, 2, 8, 256, 0xF0) == -1


def test_persistent_batch_limit_matches_the_documented_limits():
    """ops.persistent_batch_limit = the chunk size of ops.rnn_layer's batch chunks (bidirectional on 8 XCDs x 32 CUs, 64 rows for
    256 < H <= 512, 128 for H <= 256, 256 at H = 128; nothing beyond H = 512 or without XCDs)."""
    from ctc_pytorch_amd import ops
    lim = lambda H, dirs: ops.persistent_batch_limit(H, dirs, 8, 256)
    assert [lim(H, 2) for H in (128, 256, 320, 384, 512)] == [256, 128, 64, 64, 64]
    assert lim(320, 1) == 128 and lim(128, 1) == 512
    assert lim(640, 2) == 0 and ops.persistent_batch_limit(320, 2, 1, 256) == 0


def test_join_tokens_equals_the_python_join():
    """ops.join_tokens / ctcn_join_tokens (hostjoin.hip, host code only): the decoders' `' '.join(self.classes[k] for k in labelling)`
    (BeamSearch.py:152-153) for a whole decoded batch in one native pass -- equal to the Python expression on ragged rows (empty, one token,
    full width), list and {id: word} vocabularies, '' and ' ' separators, non-ASCII and > 16-byte words; ids outside the vocabulary raise what
    the expression raises."""
    from ctc_pytorch_amd import ops
    from ctc_pytorch_amd.testing import synth
    rs = np.random.RandomState(3)
    V, B, T = 62, 37, 90
    phones = [synth.int2char(V)[i] for i in range(V)]
    ids = rs.randint(0, V, size=(B, T)).astype(np.int32)
    lens = rs.randint(0, T + 1, size=B).astype(np.int32)
    lens[0], lens[1], lens[2] = 0, 1, T
    for words in (phones, {i: w for i, w in enumerate(phones)}, ["\u00e9", "\u00df", "\u6c34", "a-word-of-more-than-sixteen-bytes"] + phones[4:]):
        for sep in (" ", ""):
            assert ops.join_tokens(ids, lens, words, sep) == [sep.join(words[int(k)] for k in ids[b, : lens[b]]) for b in range(B)]
    assert ops.join_tokens(ids[:, ::2], lens // 2, phones) == [" ".join(phones[int(k)] for k in ids[b, ::2][: lens[b] // 2]) for b in range(B)]   # strided view
    assert ops.join_tokens(np.zeros((0, 5), np.int32), np.zeros(0, np.int32), phones) == []
    holes = {i: w for i, w in enumerate(phones) if i != 7}
    with pytest.raises(KeyError):
        ops.join_tokens(np.full((1, 3), 7, np.int32), np.array([3], np.int32), holes)
    assert ops.join_tokens(np.full((1, 3), 8, np.int32), np.array([3], np.int32), holes) == [" ".join([phones[8]] * 3)]
    #  a hole must not touch its neighbours: the id in front of it, the id behind it, two holes in a row, a hole at the end
    assert ops.join_tokens(np.array([[0, 1, 3]], np.int32), np.array([3], np.int32), {0: "ab", 1: "cd", 3: "ef"}) == ["ab cd ef"]
    two = {0: "ab", 1: "cd", 4: "gh"}
    assert ops.join_tokens(np.array([[1, 4, 0]], np.int32), np.array([3], np.int32), two) == ["cd gh ab"]
    for k in (2, 3):
        with pytest.raises(KeyError):
            ops.join_tokens(np.array([[0, k]], np.int32), np.array([2], np.int32), two)
    assert ops.join_tokens(np.array([[6, 5, 6]], np.int32), np.array([3], np.int32), holes) == [" ".join([phones[6], phones[5], phones[6]])]
    # a vocabulary edited in place is seen (the cache is confirmed by content)
    edit = list(phones)
    assert ops.join_tokens(np.array([[2, 3]], np.int32), np.array([2], np.int32), edit) == [phones[2] + " " + phones[3]]
    edit[3] = "zz"
    assert ops.join_tokens(np.array([[2, 3]], np.int32), np.array([2], np.int32), edit) == [phones[2] + " zz"]
    with pytest.raises(IndexError):
        ops.join_tokens(np.full((1, 3), V, np.int32), np.array([2], np.int32), phones)
    with pytest.raises(IndexError):
        ops.join_tokens(np.full((1, 3), -1, np.int32), np.array([2], np.int32), phones)
    with pytest.raises(ValueError):
        ops.join_tokens(ids, lens[:-1], phones)
    with pytest.raises(ValueError):
        ops.join_tokens(ids, lens, phones, sep=", ")
    # the raw entry point: a capacity that is too small is reported, not overrun
    from ctc_pytorch_amd import _lib
    blob, off, ln, longest, Vv = ops._vocabulary(phones)
    out, oo = np.zeros(8, np.uint8), np.zeros(B + 1, np.int64)
    assert _lib.lib().ctcn_join_tokens(ids.ctypes.data, T, lens.ctypes.data, B, blob, off.ctypes.data, ln.ctypes.data, Vv, 32, out.ctypes.data, 8, oo.ctypes.data) == -4
    assert _lib.lib().ctcn_join_tokens(None, T, lens.ctypes.data, B, blob, off.ctypes.data, ln.ctypes.data, Vv, 32, out.ctypes.data, 8, oo.ctypes.data) == -1


def test_greedy_decoder_strings_equal_the_reference_expression():
    """GreedyDecoder._strings (the host end of GreedyDecoder.decode): ' ' + phone per kept frame when the vocabulary has no space symbol, the
    space symbol as ' ' otherwise (ctcDecoder.py:80-92,152-166), list and dict vocabularies."""
    from ctc_pytorch_amd.utils.ctcDecoder import GreedyDecoder
    from ctc_pytorch_amd.testing import synth
    rs = np.random.RandomState(5)
    V, B, T = 62, 9, 40
    i2c = synth.int2char(V)
    ids = rs.randint(1, V, size=(B, T)).astype(np.int32)
    lens = rs.randint(0, T + 1, size=B).astype(np.int32)
    lens[0] = 0
    for voc in (i2c, [i2c[i] for i in range(V)]):
        g = GreedyDecoder(voc, space_idx=-1, blank_index=0)
        assert g._strings(ids, lens) == ["".join(" " + voc[int(k)] for k in ids[b, : lens[b]]) for b in range(B)]
        g = GreedyDecoder(voc, space_idx=5, blank_index=0)
        sp = voc[5]
        assert g._strings(ids, lens) == ["".join(" " if voc[int(k)] == sp else voc[int(k)] for k in ids[b, : lens[b]]) for b in range(B)]


def test_decoder_edit_distance_equals_the_reference_table():
    """Decoder.cer / wer / _edit_distance (ctcn_levenshtein, host code) against the oracle's restatement of ctcDecoder.py:131-150: strings
    (non-ASCII included), word lists, lists of arbitrary hashables, empty sides, and the symmetric / long cases the rolling row must get right."""
    import random
    from ctc_pytorch_amd import _lib
    from ctc_pytorch_amd.utils.ctcDecoder import Decoder
    random.seed(4)
    dec = Decoder({0: "_", 1: "a"}, space_idx=-1)
    alphabet = "abc \u00e9\u6c34"
    for trial in range(200):
        n, m = random.randint(0, 30), random.randint(0, 30)
        s = "".join(random.choice(alphabet) for _ in range(n))
        t = "".join(random.choice(alphabet) for _ in range(m))
        want = int(R.edit_distance([ord(c) for c in s], [ord(c) for c in t]))
        assert dec.cer(s, t) == want and dec.cer(t, s) == want
        assert dec.wer(s, t) == int(R.edit_distance(_word_ids(s, t)[0], _word_ids(s, t)[1]))
        assert Decoder._edit_distance(list(s), tuple(t)) == want
    assert dec.cer("", "abc") == 3 and dec.cer("abc", "") == 3 and dec.cer("", "") == 0 and dec.wer("a b", "") == 2
    long_a, long_b = "ab" * 700, "ba" * 650 + "c"
    assert dec.cer(long_a, long_b) == int(R.edit_distance([ord(c) for c in long_a], [ord(c) for c in long_b]))
    assert _lib.lib().ctcn_levenshtein(None, 3, None, 0) == -1


def _word_ids(s, t):
    ids = {}
    return [ids.setdefault(w, len(ids)) for w in s.split()], [ids.setdefault(w, len(ids)) for w in t.split()]


def test_option_table_enumerates_and_round_trips():
    """ctcn_option_name lists every name ctcn_set_option / ctcn_get_option know (the table the GPU tests' state harness snapshots,
    tests/conftest.py); unknown names are refused; a set value is normalised as documented and read back; ops.state_snapshot /
    restore_state put a moved option, precision and the learnt state back."""
    from ctc_pytorch_amd import _lib, ops
    L = _lib.lib()
    names = ops.option_names()
    assert len(names) == len(set(names)) >= 35 and "xcd_interleave" in names and "xcd_interleave_force" in names and "rnn_persistent" in names
    assert L.ctcn_option_name(len(names)) is None and L.ctcn_option_name(-1) is None
    assert L.ctcn_set_option(b"no_such_option", 1) < 0 and L.ctcn_get_option(b"no_such_option") == -1
    assert all(ops.get_option(n) >= -1 for n in names)
    snap = ops.state_snapshot()
    assert snap["options"]["xcd_interleave"] == 1 and snap["options"]["xcd_interleave_force"] == 0
    try:
        ops.set_option("xcd_interleave", 9)
        assert ops.get_option("xcd_interleave") == 5                    # clamped as include/ctcn.h says
        ops.set_option("bwd_poll_delay", -7)
        assert ops.get_option("bwd_poll_delay") == -1
        ops.set_precision(1 - snap["precision"])
        ops._fallback_shapes.add((0, 320, 2, 96))
        ops._drop_counter[0] += 5
        moved = ops.state_snapshot()
        assert moved != snap and moved["fallback_shapes"] == sorted(snap["fallback_shapes"] + [(0, 320, 2, 96)])
    finally:
        ops.restore_state(snap)
    assert ops.state_snapshot() == snap


def test_every_entry_point_rejects_null_and_zero_arguments_without_a_gpu():
    """The C ABI's error behaviour on the host side: every entry point of include/ctcn.h called with null pointers and zero sizes answers with
    an error code (or 0 bytes for the size queries) BEFORE it touches the device -- no crash, no HIP call -- each in a process of its own (a
    division by a zero dimension in a workspace query was found this way)."""
    code = r"""
import ctypes, sys
from ctc_pytorch_amd import _lib
L = _lib.lib()
bad = []
for name, (res, args) in _lib._SIGS.items():
    vals = []
    for a in args:
        if a in (ctypes.c_double, ctypes.c_float):
            vals.append(0.0)
        elif a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "contents"):
            vals.append(None)
        else:
            vals.append(0)
    sys.stdout.write(name + "\n"); sys.stdout.flush()
    r = getattr(L, name)(*vals)
    if name in ("ctcn_version", "ctcn_last_error", "ctcn_rnn_last_kernel", "ctcn_device_cus", "ctcn_device_xcds", "ctcn_set_status_buffer", "ctcn_comm_destroy",
                "ctcn_levenshtein", "ctcn_option_name"):
        continue                                  # (queries and no-ops that have no failing form with these arguments)
    if name.endswith("_bytes"):
        if r != 0: bad.append((name, r))
    elif not (isinstance(r, int) and r < 0):
        bad.append((name, r))
print("BAD", bad)
"""
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert p.returncode == 0, "crashed in %s: %s" % ((p.stdout.strip().splitlines() or ["?"])[-1], p.stderr[-400:])
    assert p.stdout.strip().splitlines()[-1] == "BAD []", p.stdout[-600:]
    # a communicator handle the library did not hand out never reaches RCCL (which would dereference it)
    import ctypes
    from ctc_pytorch_amd import _lib
    L = _lib.lib()
    fake = ctypes.create_string_buffer(4096)
    assert L.ctcn_comm_allreduce_sum_f32(ctypes.cast(fake, ctypes.c_void_p), ctypes.cast(fake, ctypes.c_void_p), 4, None) == -1
    assert b"not a communicator" in L.ctcn_last_error()
    assert L.ctcn_comm_destroy(ctypes.cast(fake, ctypes.c_void_p)) == -1 and L.ctcn_comm_destroy(None) == 0


def test_bench_rccl_log_summary_keeps_two_lines_of_a_kind_and_counts_the_rest(tmp_path):
    """bench.summarize_rccl_log (the `comm.rccl_info` of a multi-rank line): topology / tuning lines of an NCCL_DEBUG=INFO file, the per-channel
    'Tree N' / 'Ring N' / 'Channel N' lines folded to the first two of a kind plus a count, everything else dropped, at most `limit` lines."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    log = ["h:1:1 [0] NCCL INFO Dmabuf feature disabled without NCCL_DMABUF_ENABLE=1",
           "h:1:2 [0] NCCL INFO cudaDriverVersion 0",
           "h:1:2 [0] NCCL INFO NET/Socket : Using [0]eth0:10.0.0.1<0>",
           "h:1:2 [0] NCCL INFO comm 0x55 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0"]
    log += ["h:1:2 [0] NCCL INFO Tree %d : -1 -> 0 -> 1/-1/-1" % i for i in range(64)]
    log += ["h:1:2 [0] NCCL INFO Channel %02d/128 : 0 1 2 3 4 5 6 7" % i for i in range(128)]
    log += ["h:1:2 [0] NCCL INFO Ring %d : 7 -> 0 -> 1 comm 0x55 nRanks 08 busId 8b000" % i for i in range(128)]
    log += ["h:1:2 [0] NCCL INFO Connected all rings", "h:1:2 [0] NCCL INFO something unrelated", "h:1:2 [0] NCCL INFO P2P Chunksize set to 524288"]
    p = tmp_path / "rccl.log"
    p.write_text("\n".join(log) + "\n")
    out = bench.summarize_rccl_log(str(p))
    assert sum("NCCL INFO Tree " in l for l in out) == 2 and sum("NCCL INFO Ring " in l for l in out) == 2 and sum("NCCL INFO Channel " in l for l in out) == 2
    assert out[-3:] == ["(128 'Channel N' lines in all)", "(128 'Ring N' lines in all)", "(64 'Tree N' lines in all)"]
    assert any("NET/Socket" in l for l in out) and any("nRanks 8" in l for l in out) and any("Connected all rings" in l for l in out) and any("P2P Chunksize" in l for l in out)
    assert not any("cudaDriverVersion" in l or "unrelated" in l for l in out)
    assert len(bench.summarize_rccl_log(str(p), limit=5)) == 5 + 3
