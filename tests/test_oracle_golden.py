"""Pin the oracle (oracle/np_ref.py, oracle/beam_ref.c) against the golden vectors that
oracle/gen_golden.py captured from the imported reference (torch 2.10 CPU).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_ref as R
from oracle import beam_ref
from ctc_pytorch_amd.testing import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.mark.parametrize("kind", ["lstm", "gru", "rnn"])
def test_rnn_layer(kind):
    z = load("rnn_" + kind)
    w = [(z["w.rnn.weight_ih_l0"], z["w.rnn.weight_hh_l0"]),
         (z["w.rnn.weight_ih_l0_reverse"], z["w.rnn.weight_hh_l0_reverse"])]
    y, saved = R.birnn_fwd(kind, z["x"], w)
    assert maxabs(y, z["y"]) < 2e-6
    dx, grads = R.birnn_bwd(kind, z["x"], w, saved, z["dy"])
    assert maxabs(dx, z["dx"]) < 5e-6
    for d, suf in enumerate(["", "_reverse"]):
        assert maxabs(grads[d][0], z["g.rnn.weight_ih_l0" + suf]) < 2e-5
        assert maxabs(grads[d][1], z["g.rnn.weight_hh_l0" + suf]) < 2e-5


def test_batchnorm_over_tb():
    z = load("bn_tb")
    rm, rv = np.zeros_like(z["gamma"], dtype=np.float64), np.ones_like(z["gamma"], dtype=np.float64)
    for step in range(2):
        x = z["x%d" % step]
        T, B, C = x.shape
        y, mean, var = R.bn_train_fwd(x.reshape(T * B, C), z["gamma"], z["beta"])
        assert maxabs(y.reshape(T, B, C), z["y%d" % step]) < 3e-6
        dx, dg, db = R.bn_train_bwd(x.reshape(T * B, C), z["gamma"], mean, var, z["dy%d" % step].reshape(T * B, C))
        assert maxabs(dx.reshape(T, B, C), z["dx%d" % step]) < 3e-6
        assert maxabs(dg, z["dgamma%d" % step]) < 2e-5
        assert maxabs(db, z["dbeta%d" % step]) < 2e-5
        rm, rv = R.bn_running_update(rm, rv, mean, var, T * B)
        assert maxabs(rm, z["rm%d" % step]) < 1e-6
        assert maxabs(rv, z["rv%d" % step]) < 1e-6
    xe = z["x_eval"]
    T, B, C = xe.shape
    ye = R.bn_eval_fwd(xe.reshape(T * B, C), z["gamma"], z["beta"], rm, rv)
    assert maxabs(ye.reshape(T, B, C), z["y_eval"]) < 3e-6


def _conv_front_fwd(z, x):
    strides = [(1, 2), (2, 2)]
    acts, stats = [], []
    h = x[:, None, :, :]
    for n in range(2):
        w, b = z["w.%d.conv.weight" % n], z["w.%d.conv.bias" % n]
        c = R.conv2d_fwd(h, w, b, strides[n], (1, 1))
        Bn, Cn, Tn, Fn = c.shape
        c2 = c.transpose(0, 2, 3, 1).reshape(-1, Cn)
        y, mean, var = R.bn_train_fwd(c2, z["w.%d.batch_norm.weight" % n], z["w.%d.batch_norm.bias" % n])
        a = np.maximum(y, 0.0).reshape(Bn, Tn, Fn, Cn).transpose(0, 3, 1, 2)
        acts.append((h, c2, mean, var, y, (Bn, Cn, Tn, Fn)))
        h = a
    return h, acts


def test_conv_front():
    z = load("conv_front_relu")
    out, acts = _conv_front_fwd(z, z["x"].astype(np.float64))
    assert maxabs(out, z["conv_out"]) < 5e-6
    r = R.conv_layout_to_rnn(out)
    assert r.shape == z["rnn_in"].shape and maxabs(r, z["rnn_in"]) < 5e-6
    # backward
    strides = [(1, 2), (2, 2)]
    d = R.conv_layout_from_rnn(z["d_rnn_in"].astype(np.float64), 32)
    for n in (1, 0):
        h, c2, mean, var, y, (Bn, Cn, Tn, Fn) = acts[n]
        d2 = d.transpose(0, 2, 3, 1).reshape(-1, Cn) * (y > 0)
        dc2, dg, db = R.bn_train_bwd(c2, z["w.%d.batch_norm.weight" % n], mean, var, d2)
        assert maxabs(dg, z["g.%d.batch_norm.weight" % n]) < 1e-4
        assert maxabs(db, z["g.%d.batch_norm.bias" % n]) < 1e-4
        dc = dc2.reshape(Bn, Tn, Fn, Cn).transpose(0, 3, 1, 2)
        d, dw, dbias = R.conv2d_bwd(h, z["w.%d.conv.weight" % n], strides[n], (1, 1), dc)
        assert maxabs(dw, z["g.%d.conv.weight" % n]) < 2e-4
        assert maxabs(dbias, z["g.%d.conv.bias" % n]) < 2e-4
    assert maxabs(d[:, 0], z["dx"]) < 1e-5


def test_fc_logsoftmax_argmax():
    z = load("fc_lsm")
    x = z["x"]
    y, mean, var = R.bn_train_fwd(x, z["w.0.weight"], z["w.0.bias"])
    logits = y @ z["w.1.weight"].astype(np.float64).T
    assert maxabs(logits, z["logits"]) < 5e-6
    T, B, V = z["lp"].shape
    lp = R.log_softmax(logits.reshape(T, B, V))
    assert maxabs(lp, z["lp"]) < 5e-6
    assert np.array_equal(R.argmax_first(z["lp"]), z["argmax"])
    dlog = R.log_softmax_bwd(lp, z["dlp"].astype(np.float64)).reshape(T * B, V)
    assert maxabs(dlog.T @ y, z["g.1.weight"]) < 5e-5
    dy = dlog @ z["w.1.weight"].astype(np.float64)
    dx, dg, db = R.bn_train_bwd(x, z["w.0.weight"], mean, var, dy)
    assert maxabs(dx, z["dx"]) < 1e-5
    assert maxabs(dg, z["g.0.weight"]) < 5e-5


def test_ctc_loss_and_grad():
    z = load("ctc_loss")
    B = z["lp"].shape[1]
    nll, grad = R.ctc_loss(z["lp"], z["targets"], z["in_len"], z["tgt_len"])
    assert np.allclose(nll, z["nll"], rtol=2e-6, atol=2e-5)
    assert abs(nll.sum() / B - float(z["loss"])) < 1e-4
    assert maxabs(grad / B, z["dlp"]) < 1e-5   # torch runs the lattice in f32 log-space
    dlog = R.log_softmax_bwd(z["lp"].astype(np.float64), grad / B)
    assert maxabs(dlog, z["dlogits"]) < 1e-5
    # zero beyond each length; rows sum to ~0 after log_softmax backward
    for b in range(B):
        assert np.all(grad[z["in_len"][b]:, b] == 0)
    # infeasible sample -> +inf and NaN rows exactly where torch has them
    nll2, grad2 = R.ctc_loss(z["lp"], z["targets"], z["in_len_inf"], z["tgt_len"])
    assert np.isinf(nll2[4]) and np.isinf(z["nll_inf"][4])
    assert np.array_equal(np.isnan(grad2), np.isnan(z["dlp_inf"]))
    ok = ~np.isnan(grad2)
    assert maxabs((grad2 / B)[ok], z["dlp_inf"][ok]) < 1e-5


def test_length_table():
    rows = load("length_table")["rows"]
    for Tmax in np.unique(rows[:, 1]):
        for Tout in np.unique(rows[rows[:, 1] == Tmax, 2]):
            sel = rows[(rows[:, 1] == Tmax) & (rows[:, 2] == Tout)]
            frac = np.array([np.float32(float(n) / float(Tmax)) for n in sel[:, 0]], dtype=np.float32)
            assert np.array_equal(R.frames_from_fraction(frac, int(Tout)), sel[:, 3])


def test_greedy_and_scoring():
    z = load("decoders")
    meta = json.load(open(os.path.join(G, "decoders.json")))
    i2c = synth.int2char(62)
    for regime in ("peaky", "flat"):
        assert np.array_equal(R.argmax_first(z["lp_" + regime]), z["argmax_" + regime])
        assert R.greedy_strings(z["lp_" + regime], meta["lens"], i2c) == meta["greedy_" + regime]
    errs, toks = R.compute_wer(z["argmax_peaky"].T, meta["lens"], z["wer_targets"], z["wer_tgt_len"])
    assert [errs, toks] == meta["compute_wer"]
    sc = meta["score_greedy_peaky"]
    tc = sum(R.edit_distance(a, b) for a, b in zip(meta["greedy_peaky"], meta["labels"]))
    tw = sum(R.edit_distance(a.split(), b.split()) for a, b in zip(meta["greedy_peaky"], meta["labels"]))
    assert (tc, tw) == (sc["total_cer"], sc["total_wer"])


def test_beam_search_strings():
    z = load("decoders")
    meta = json.load(open(os.path.join(G, "decoders.json")))
    i2c = synth.int2char(62)
    tab = beam_ref.arpa_table(os.path.join(G, "lm_phone_bg.arpa"), i2c)
    want_tab = load("lm_table")["lm_table"]
    ok = ~np.isnan(want_tab)
    assert np.array_equal(ok, ~np.isnan(tab)) and np.array_equal(tab[ok], want_tab[ok])
    n = 0
    for key, want in meta.items():
        if not key.startswith("beam_"):
            continue
        _, regime, w, a = key.split("_")
        lp = z["lp_" + regime]
        probs = np.exp(lp.astype(np.float32)).transpose(1, 0, 2)   # float32 exp like torch.exp (ctcDecoder.py:190)
        got = beam_ref.decode_strings(probs, meta["lens"], tab, float(a[1:]), int(w[1:]), i2c)
        assert got == want, key
        n += 1
    assert n >= 10


def test_beam_search_nbest_pins_the_oracle_to_the_reference():
    """n-best: oracle/beam_ref.c's first `nbest` labellings == the reference's whole final `last.sort()` captured by
    oracle/gen_golden.py (decoders_nbest.json), entry 0 == the string the reference's decode returns."""
    z = load("decoders")
    meta = json.load(open(os.path.join(G, "decoders.json")))
    nb = json.load(open(os.path.join(G, "decoders_nbest.json")))
    i2c = synth.int2char(62)
    tab = beam_ref.arpa_table(os.path.join(G, "lm_phone_bg.arpa"), i2c)
    assert len(nb) >= 4
    for key, rec in nb.items():
        _, regime, w, a = key.split("_")
        W = int(w[1:])
        probs = np.exp(z["lp_" + regime].astype(np.float32)).transpose(1, 0, 2)
        N = min(W, 5)
        ids, score, st = beam_ref.decode_ids_nbest(probs, meta["lens"], tab, float(a[1:]), W, N)
        assert not st.any()
        assert ids == rec["labellings"], key
        for b, utt in enumerate(ids):
            assert " ".join(i2c[k] for k in utt[0]) == rec["best_string"][b] == meta["beam_%s_W%d_a%g" % (regime, W, float(a[1:]))][b]
            assert all(score[b, k] >= score[b, k + 1] for k in range(len(utt) - 1))          # sorted, best first
        one, s1, _ = beam_ref.decode_ids(probs, meta["lens"], tab, float(a[1:]), W)
        assert [u[0] for u in ids] == [list(map(int, u)) for u in one] and np.array_equal(s1, score[:, 0])


def test_beam_search_nbest_short_final_beam():
    """n-best when the final beam holds fewer labellings than asked for: every returned list is a prefix of the longer request's list, entries
    past the final beam are absent (count < nbest), and scores are sorted; two-class input where the search can only ever build a handful
    of distinct labellings."""
    rs = np.random.RandomState(3)
    T, V = 6, 3
    p = rs.dirichlet(np.ones(V) * 0.7, size=(2, T)).astype(np.float32)             # (B=2, T, V) probabilities
    tab = np.zeros((V + 1, V + 1), dtype=np.float64)
    full, fs, st = beam_ref.decode_ids_nbest(p, [T, T - 2], tab, 0.0, 8, 8)
    assert not st.any()
    for b in range(2):
        assert 1 <= len(full[b]) <= 8
        assert len({tuple(y) for y in full[b]}) == len(full[b])                        # distinct labellings
        assert all(fs[b, k] >= fs[b, k + 1] for k in range(len(full[b]) - 1))
        assert (fs[b, len(full[b]):] == 0).all()
    for N in (1, 2, 5):
        part, ps, _ = beam_ref.decode_ids_nbest(p, [T, T - 2], tab, 0.0, 8, N)
        for b in range(2):
            assert part[b] == full[b][:N] and np.array_equal(ps[b, :len(part[b])], fs[b, :len(part[b])])


def test_beam_error_paths():
    i2c = synth.int2char(62)
    tab = beam_ref.arpa_table(os.path.join(G, "lm_phone_bg.arpa"), i2c)
    probs = np.full((1, 5, 62), 1e-3, dtype=np.float32)
    probs[:, :, 0] = 0.95                                   # every frame skipped -> best labelling is ()
    with pytest.raises(IndexError):
        beam_ref.decode_strings(probs, [5], tab, 0.1, 5, i2c)
    probs = np.full((1, 5, 62), 1.0 / 62, dtype=np.float32)
    probs[0, 2, 7] = 0.0                                    # math.log(0)
    with pytest.raises(ValueError):
        beam_ref.decode_strings(probs, [5], tab, 0.1, 5, i2c)


def test_philox_restatement_matches_random123_known_answers():
    """oracle/philox.py (the generator behind ctcn_dropout) against the published Random123 philox4x32-10 vectors, and the
    statistics / determinism of the dropout mapping built on it (reference call sites model_ctc.py:34,67)."""
    from oracle import philox
    for ctr, key, want in philox.KAT:
        got = philox.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    x = np.ones(100003, dtype=np.float32)
    y = philox.dropout(x, 0.25, seed=0x123456789ABCDEF, offset=7)
    assert abs(float((y != 0).mean()) - 0.75) < 5e-3
    assert np.all(y[y != 0] == np.float32(1.0) / (np.float32(1.0) - np.float32(0.25)))
    assert np.array_equal(y, philox.dropout(x, 0.25, seed=0x123456789ABCDEF, offset=7))
    # element i of a call with offset o uses block o + i // 4: a call shifted by one block sees the stream shifted by 4 elements
    w0, w1 = philox.dropout_words(64, 5, 10), philox.dropout_words(64, 5, 11)
    assert np.array_equal(w0[4:], w1[:-4]) and not np.array_equal(w0, w1)


@pytest.mark.parametrize("shape,k", [((2, 3, 9, 8), (2, 2)), ((1, 2, 7, 5), (3, 1)), ((2, 1, 4, 6), (1, 3)), ((1, 1, 5, 5), (5, 5))])
def test_maxpool_restatement_is_pinned_to_torch(shape, k):
    """oracle/np_ref.py maxpool2d_fwd / _bwd against torch's CPU MaxPool2d (the arithmetic the reference's nn.MaxPool2d runs,
    model_ctc.py:53): values, winner positions on ties and NaNs, and the routed gradient."""
    rs = np.random.RandomState(3)
    x = np.maximum(rs.standard_normal(shape), 0).astype(np.float32)
    x.reshape(-1)[rs.randint(0, x.size, size=3)] = np.nan
    y, arg = R.maxpool2d_fwd(x, *k)
    xt = torch.from_numpy(x).requires_grad_()
    yt, idx = torch.nn.functional.max_pool2d(xt, k, return_indices=True)
    assert np.array_equal(y, yt.detach().numpy(), equal_nan=True)
    Ho, Wo = shape[2] // k[0], shape[3] // k[1]
    hh, ww = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    flat = (hh * k[0] + arg // k[1]) * shape[3] + ww * k[1] + arg % k[1]
    assert np.array_equal(flat, idx.numpy())
    dy = rs.standard_normal(y.shape).astype(np.float32)
    yt.backward(torch.from_numpy(dy))
    assert np.array_equal(R.maxpool2d_bwd(dy, arg, shape, *k), xt.grad.numpy())
