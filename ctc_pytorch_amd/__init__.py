"""ctc_pytorch_amd -- MI355X (gfx950) native CTC acoustic-model training & decode path.

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C ABI of include/ctcn.h  -> libctcn.so
  _lib.py    build + ctypes loader (fails loudly when the library is missing; no CPU fallback)
  ops.py     autograd.Function wrappers over the C ABI
  nn.py      HIP-backed stand-ins for the torch.nn names the reference drivers use
  models/    CTC_Model / BatchRNN / LayerCNN        (reference timit/models/model_ctc.py)
  utils/     Decoder / GreedyDecoder / BeamDecoder, ctcBeamSearch, LanguageModel, data_loader
             (reference timit/utils/{ctcDecoder,BeamSearch,NgramLM,data_loader}.py)
  steps/     run_epoch / trainer / decode-and-score counterparts (reference timit/steps/{train,test}_ctc.py)
  parallel.py  utterance-sharded data parallelism: flat gradient buffer + one RCCL all-reduce
"""
__version__ = "0.1.0"
