"""Decode-and-score counterpart of the reference's timit/steps/test_ctc.py:72-109 on the HIP path.

`decode_and_score(model, loader, decoder, index2word, device)` reproduces the reference loop: log-probs ->
frames = floor(float32(frac)*T_out) -> decoder.decode -> label strings ' '.join(phones) -> CER = character
Levenshtein / total label characters, WER = token Levenshtein / total tokens (greedy strings keep their leading
space, exactly as the reference scores them)."""
import argparse
import os
import sys
import time

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ctc_pytorch_amd.steps.train_ctc import frames_from_fraction  # noqa: E402


def decode_and_score(model, loader, decoder, index2word, device, verbose=False, log=print):
    model.eval()
    total_wer = 0
    total_cer = 0

    def score(decoded, targets, target_sizes):
        nonlocal total_wer, total_cer
        targets, target_sizes = targets.numpy(), target_sizes.numpy()
        labels = [" ".join(index2word[int(k)] for k in targets[i][: target_sizes[i]]) for i in range(len(targets))]
        for x in range(len(labels)):
            if verbose:
                log("origin : " + labels[x])
                log("decoded: " + decoded[x])
            total_cer += decoder.cer(decoded[x], labels[x])
            total_wer += decoder.wer(decoded[x], labels[x])
            decoder.num_word += len(labels[x].split())
            decoder.num_char += len(labels[x])

    # the beam search of a batch runs on one of three extra streams (the model forward stays on the current one: it owns the library
    # workspace) and its strings are collected later: two searches fill the device (a batch is one workgroup per utterance, <= 128 of 256
    # CUs) but each lasts as long as its LONGEST utterance -- the third one's workgroups take the CUs the shorter utterances have left
    # (cfg5: 206 k -> 246 k utt/s) -- and the host-side scoring of batch i overlaps with the searches behind it.  Same order, same results.
    NS = 3
    pipelined = hasattr(decoder, "decode_async") and torch.device(device).type == "cuda"
    streams = [torch.cuda.Stream(device=device) for _ in range(NS)] if pipelined else []
    pending = []
    with torch.no_grad():
        for i, (inputs, input_sizes, targets, target_sizes, utt_list) in enumerate(loader):
            probs = model(inputs.to(device))
            lens = frames_from_fraction(input_sizes, probs.size(0)).tolist()
            if not pipelined:
                score(decoder.decode(probs, lens), targets, target_sizes)
                continue
            st = streams[i % NS]
            st.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(st):
                wait = decoder.decode_async(probs, lens)
            probs.record_stream(st)
            pending.append((wait, targets, target_sizes))
            if len(pending) == NS + 1:           # NS searches running, one queued behind them
                w, t, ts = pending.pop(0)
                score(w(), t, ts)
        for w, t, ts in pending:
            score(w(), t, ts)
    CER = (float(total_cer) / decoder.num_char) * 100
    WER = (float(total_wer) / decoder.num_word) * 100
    log("Character error rate on test set: %.4f" % CER)
    log("Word error rate on test set: %.4f" % WER)
    return CER, WER


def decode_and_score_sharded(model, loader, decoder, index2word, device, rank=0, world=1, verbose=False, log=print):
    """Replicas-only data parallel decode (SURVEY 8e "Decode"): rank r takes minibatches r, r + world, ... of the loader, no
    collective on the data path; the four error / length totals are summed over the ranks at the end."""
    if world <= 1:
        return decode_and_score(model, loader, decoder, index2word, device, verbose=verbose, log=log)
    import torch.distributed as dist
    mine = [batch for i, batch in enumerate(loader) if i % world == rank]
    decoder.num_word = decoder.num_char = 0
    quiet = (lambda *_: None)
    if mine:
        cer, wer = decode_and_score(model, mine, decoder, index2word, device, verbose=verbose, log=quiet)
        tot = [cer * decoder.num_char / 100.0, wer * decoder.num_word / 100.0, decoder.num_char, decoder.num_word]
    else:
        tot = [0.0, 0.0, 0.0, 0.0]
    t = torch.tensor(tot, dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    tot = [float(v) for v in t.cpu()]
    decoder.num_char, decoder.num_word = int(round(tot[2])), int(round(tot[3]))
    CER, WER = tot[0] / max(tot[2], 1.0) * 100, tot[1] / max(tot[3], 1.0) * 100
    if rank == 0:
        log("Character error rate on test set: %.4f" % CER)
        log("Word error rate on test set: %.4f" % WER)
    return CER, WER


def make_decoder(opts, index2word):
    """decode_type == 'Greedy' -> GreedyDecoder, anything else -> BeamDecoder(beam_width, lm_path, lm_alpha): the choice the
    reference makes at test_ctc.py:64-67, on the same YAML keys."""
    from ctc_pytorch_amd.utils.ctcDecoder import BeamDecoder, GreedyDecoder
    if opts.decode_type == "Greedy":
        return GreedyDecoder(index2word, space_idx=-1, blank_index=0)
    return BeamDecoder(index2word, beam_width=opts.beam_width, blank_index=0, space_idx=-1, lm_path=opts.lm_path, lm_alpha=opts.lm_alpha)


def main(conf, test_loader=None, index2word=None, log=print):
    """Counterpart of the reference's test() (timit/steps/test_ctc.py:21-109) for the same YAML: checkpoint
    <checkpoint_dir>/<exp_name>/ctc_best_model.pkl -> CTC_Model -> Greedy / Beam decoder -> CER / WER over the test set.
    Returns (CER, WER).  Launched under torchrun it decodes replicas-only (utterance minibatches dealt over the ranks)."""
    from ctc_pytorch_amd import parallel
    from ctc_pytorch_amd.steps.train_ctc import Config
    opts = Config()
    for k, v in conf.items():
        setattr(opts, k, v)
    if not getattr(opts, "use_gpu", True):
        raise RuntimeError("ctc_pytorch_amd: use_gpu must be True -- the HIP path has no CPU fallback")
    rank, world, local = parallel.init_from_env()
    device = torch.device("cuda", local)
    model, package = load_package(os.path.join(opts.checkpoint_dir, opts.exp_name, "ctc_best_model.pkl"), device)
    if test_loader is None:
        from ctc_pytorch_amd.utils.data_loader import SpeechDataLoader, SpeechDataset, Vocab
        vocab = Vocab(opts.vocab_file)
        index2word = vocab.index2word
        test_loader = SpeechDataLoader(SpeechDataset(vocab, opts.test_scp_path, opts.test_lab_path, opts), batch_size=opts.batch_size,
                                       shuffle=False, num_workers=opts.num_workers)
    decoder = make_decoder(opts, index2word)
    start = time.time()
    cer, wer = decode_and_score_sharded(model, test_loader, decoder, index2word, device, rank, world, verbose=bool(getattr(opts, "verbose", False)), log=log)
    if rank == 0:
        log("time used for decode: %.4f minutes." % ((time.time() - start) / 60.0))
    return cer, wer


def load_package(path, device):
    """Rebuild a CTC_Model from a package written by CTC_Model.save_package (either side's)."""
    from ctc_pytorch_amd.models.model_ctc import CTC_Model
    package = torch.load(path, map_location="cpu", weights_only=False)
    model = CTC_Model(rnn_param=package["rnn_param"], add_cnn=package["add_cnn"], cnn_param=package["cnn_param"],
                      num_class=package["num_class"], drop_out=package["_drop_out"])
    model.load_state_dict(package["state_dict"])
    return model.to(device), package


if __name__ == "__main__":
    import yaml
    ap = argparse.ArgumentParser(description="decode + score a ctc_best_model.pkl on MI355X")
    ap.add_argument("--conf", help="conf file (same keys as timit/conf/ctc_config.yaml)")
    a = ap.parse_args()
    main(yaml.safe_load(open(a.conf, "r")))
