"""Decode-and-score counterpart of the reference's timit/steps/test_ctc.py:72-109 on the HIP path.

`decode_and_score(model, loader, decoder, index2word, device)` reproduces the reference loop: log-probs ->
frames = floor(float32(frac)*T_out) -> decoder.decode -> label strings ' '.join(phones) -> CER = character
Levenshtein / total label characters, WER = token Levenshtein / total tokens (greedy strings keep their leading
space, exactly as the reference scores them)."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ctc_pytorch_amd.steps.train_ctc import frames_from_fraction  # noqa: E402


def decode_and_score(model, loader, decoder, index2word, device, verbose=False, log=print):
    model.eval()
    total_wer = 0
    total_cer = 0
    with torch.no_grad():
        for inputs, input_sizes, targets, target_sizes, utt_list in loader:
            probs = model(inputs.to(device))
            lens = frames_from_fraction(input_sizes, probs.size(0)).tolist()
            decoded = decoder.decode(probs, lens)
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
    CER = (float(total_cer) / decoder.num_char) * 100
    WER = (float(total_wer) / decoder.num_word) * 100
    log("Character error rate on test set: %.4f" % CER)
    log("Word error rate on test set: %.4f" % WER)
    return CER, WER


def load_package(path, device):
    """Rebuild a CTC_Model from a package written by CTC_Model.save_package (either side's)."""
    from ctc_pytorch_amd.models.model_ctc import CTC_Model
    package = torch.load(path, map_location="cpu", weights_only=False)
    model = CTC_Model(rnn_param=package["rnn_param"], add_cnn=package["add_cnn"], cnn_param=package["cnn_param"],
                      num_class=package["num_class"], drop_out=package["_drop_out"])
    model.load_state_dict(package["state_dict"])
    return model.to(device), package
