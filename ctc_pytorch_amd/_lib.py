"""Loader for libctcn.so (the gfx950 HIP kernels behind the C ABI of include/ctcn.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel reports an error
every op raises (RuntimeError) -- it never silently routes through PyTorch or the oracle.

`build()` cross-compiles all csrc/*.hip for gfx950 with hipcc into ctc_pytorch_amd/libctcn.so (in-tree so
that it travels with gpurun snapshots).  `import torch` happens before the dlopen on purpose: torch bundles
its own libamdhip64.so.7 and the dynamic loader then binds libctcn.so to that same runtime (same SONAME),
so torch streams / device pointers are valid inside the library.
"""
import ctypes
import glob
import os
import subprocess

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# (CTCN_LIBCTCN: development only -- tools/flag_lottery.sh points it at a variant build of the same sources; the product loads the in-tree library)
SO_PATH = os.environ.get("CTCN_LIBCTCN") or os.path.join(_HERE, "libctcn.so")
_lib = None

c_void_p, c_int, c_float, c_double, c_size_t, c_u64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                      ctypes.c_double, ctypes.c_size_t, ctypes.c_uint64)

# name -> (restype, argtypes); mirrors include/ctcn.h one to one
P, I, F, D, Z, U = c_void_p, c_int, c_float, c_double, c_size_t, c_u64
_SIGS = {
    "ctcn_version": (I, []),
    "ctcn_last_error": (ctypes.c_char_p, []),
    "ctcn_device_cus": (I, []),
    "ctcn_device_xcds": (I, []),
    "ctcn_set_option": (I, [ctypes.c_char_p, I]),
    "ctcn_get_option": (I, [ctypes.c_char_p]),
    "ctcn_option_name": (ctypes.c_char_p, [I]),
    "ctcn_set_status_buffer": (I, [P]),
    "ctcn_gemm": (I, [I, I, I, I, I, P, I, P, I, P, I, F, I, P, Z, P]),
    "ctcn_transpose01": (I, [P, P, I, I, I, P]),
    "ctcn_copy_strided4": (I, [P, P, I, I, I, I, Z, Z, Z, Z, P]),
    "ctcn_relu_fwd": (I, [P, P, Z, P]),
    "ctcn_relu_bwd": (I, [P, P, P, Z, P]),
    "ctcn_maxpool2d_fwd": (I, [P, P, P, Z, I, I, I, I, P]),
    "ctcn_maxpool2d_bwd": (I, [P, P, P, Z, I, I, I, I, P]),
    "ctcn_rnn_scratch_bytes": (Z, [I, I, I, I]),
    "ctcn_rnn_fwd": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, I, P, Z, P]),
    "ctcn_rnn_fwd_dropout": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, F, U, U, I, P, Z, P]),
    "ctcn_rnn_bwd": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, I, P, P, Z, P]),
    "ctcn_rnn_bwd_dropout": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, I, P, P, Z, P, F, U, U, P]),
    "ctcn_rnn_bwd_weights": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, F, I, ctypes.c_uint, P, Z, P]),
    "ctcn_bn_ws_bytes": (Z, [I, I, I]),
    "ctcn_bn_fwd_train": (I, [P, P, P, P, P, P, P, P, I, I, I, F, F, I, P, Z, P, P]),
    "ctcn_bn_fwd_eval": (I, [P, P, P, P, P, P, I, I, I, F, I, P]),
    "ctcn_bn_fwd_sums": (I, [P, P, I, I, I, P, Z, P]),
    "ctcn_bn_fwd_finish": (I, [P, P, P, P, P, P, P, P, P, D, I, I, I, F, F, I, P, P]),
    "ctcn_bn_bwd_sums": (I, [P, P, P, P, P, P, I, I, I, I, P, Z, P]),
    "ctcn_bn_bwd_finish": (I, [P, P, P, P, P, P, P, P, P, P, P, D, I, I, I, I, F, P, Z, P]),
    "ctcn_bn_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, F, P, Z, P]),
    "ctcn_bn_fwd_train_dropout": (I, [P, P, P, P, P, P, P, P, I, I, I, F, F, I, P, Z, P, P, F, U, U]),
    "ctcn_bn_bwd_dropout": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, F, P, Z, P, F, U, U]),
    "ctcn_dropout": (I, [P, P, Z, F, U, U, P]),
    "ctcn_conv2d_ws_bytes": (Z, [I] * 11),
    "ctcn_conv2d_fwd": (I, [P, P, P, P] + [I] * 11 + [P]),
    "ctcn_conv2d_bwd": (I, [P, P, P, P, P, P] + [I] * 11 + [F, P, Z, P]),
    "ctcn_bctf_to_tbcf": (I, [P, P, I, I, I, I, P]),
    "ctcn_tbcf_to_bctf": (I, [P, P, I, I, I, I, P]),
    "ctcn_log_softmax_fwd": (I, [P, P, P, I, I, P]),
    "ctcn_log_softmax_bwd": (I, [P, P, P, I, I, P]),
    "ctcn_argmax": (I, [P, P, I, I, P]),
    "ctcn_rnn_fwd_ex": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, I, P, Z, P, P]),
    "ctcn_rnn_bwd_ex": (I, [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, I, P, P, Z, P, P]),
    "ctcn_ctc_fwd": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "ctcn_ctc_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "ctcn_ctc_fwd_both": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "ctcn_ctc_grad": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "ctcn_sum_f32": (I, [P, P, I, P]),
    "ctcn_adam_step": (I, [P, P, P, P, Z, F, F, F, F, F, I, P]),
    "ctcn_greedy_collapse": (I, [P, Z, Z, P, P, P, I, I, I, P]),
    "ctcn_edit_distance": (I, [P, P, P, P, P, I, I, I, I, P]),
    "ctcn_step_stats": (I, [P, P, P, I, P, P, P]),
    "ctcn_comm_unique_id": (I, [P]),
    "ctcn_comm_init": (I, [P, I, I, ctypes.POINTER(ctypes.c_void_p)]),
    "ctcn_comm_allreduce_sum_f32": (I, [P, P, Z, P]),
    "ctcn_comm_destroy": (I, [P]),
    "ctcn_diag_squat": (I, [I, I, I, ctypes.c_uint, P]),
    "ctcn_diag_pipeline_chunks": (I, [I, I, I, I, I, I, I, I, ctypes.c_uint]),
    "ctcn_rnn_last_kernel": (ctypes.c_char_p, [I]),
    "ctcn_levenshtein": (ctypes.c_longlong, [P, ctypes.c_longlong, P, ctypes.c_longlong]),
    "ctcn_join_tokens": (ctypes.c_longlong, [P, ctypes.c_longlong, P, I, P, P, P, I, I, P, ctypes.c_longlong, P]),
    "ctcn_beam_ws_bytes": (Z, [I, I, I, I]),
    "ctcn_beam_decode": (I, [P, I, P, P, D, I, I, P, P, P, P, I, I, I, P, Z, P]),
    "ctcn_beam_decode_nbest": (I, [P, I, P, P, D, I, I, I, P, P, P, P, P, I, I, I, P, Z, P]),
}


class RnnCall(ctypes.Structure):
    """ctcn_rnn_call of include/ctcn.h: the per-call extras of ctcn_rnn_fwd_ex / ctcn_rnn_bwd_ex (the library keeps no state between calls)."""
    _fields_ = [("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint64), ("drop_offset", ctypes.c_uint64), ("y_drop", ctypes.c_void_p),
                ("dy_tmp", ctypes.c_void_p), ("side_stream", ctypes.c_void_p), ("side_event", ctypes.c_void_p), ("side_ws", ctypes.c_void_p),
                ("side_ws_bytes", ctypes.c_size_t), ("xcd_allow", ctypes.c_uint), ("prelaunch_event", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("launched", ctypes.POINTER(ctypes.c_char_p))]


def sources():
    return sorted(glob.glob(os.path.join(_CSRC, "*.hip")))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> ctc_pytorch_amd/libctcn.so (cross-compiles without a GPU).  One object per source
    under csrc/_obj/ (compiled in parallel, only when the source or a header is newer), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources()
    hdrs = glob.glob(os.path.join(_CSRC, "*.h")) + [os.path.join(os.path.dirname(_HERE), "include", "ctcn.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    if jobs or not os.path.exists(SO_PATH) or any(os.path.getmtime(SO_PATH) < os.path.getmtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", SO_PATH] + objs)
    return SO_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "ctc_pytorch_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
        l = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
        for var, opt in (("CTCN_RNN_PERSISTENT", b"rnn_persistent"), ("CTCN_HANDOFF", b"handoff"), ("CTCN_GEMM_BIG_TILES", b"gemm_big_tiles"), ("CTCN_BWD_SCATTER", b"bwd_scatter"), ("CTCN_HANDOFF_TAGS", b"handoff_tags"), ("CTCN_SIDE_SPLIT_WGS", b"side_split_wgs")):
            env = os.environ.get(var)
            if env is not None:
                l.ctcn_set_option(opt, int(env))
        for var, val in os.environ.items():            # CTCN_OPT_<NAME>=<int>: any ctcn_set_option name (A/B measurements)
            if var.startswith("CTCN_OPT_"):
                check(l.ctcn_set_option(var[len("CTCN_OPT_"):].lower().encode(), int(val)), "set_option(%s)" % var)
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = lib().ctcn_last_error()
        raise RuntimeError("libctcn %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_STATUS = {}


def status_word(device):
    """Per-device sticky int32 the persistent kernels write on a hand-off timeout (0 = healthy)."""
    key = (device.type, device.index)
    t = _STATUS.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _STATUS[key] = t
        check(lib().ctcn_set_status_buffer(ctypes.c_void_p(t.data_ptr())), "set_status_buffer")
    return t


def check_status(device):
    """Synchronising health check: raises if a persistent kernel reported a hand-off timeout."""
    t = _STATUS.get((device.type, device.index))
    if t is not None:
        v = int(t.item())
        if v != 0:
            t.zero_()
            raise RuntimeError("libctcn: persistent recurrent kernel reported status %d (in-launch hand-off timed out)" % v)


_WS = {}
_WS_LOCK = __import__("threading").Lock()
# Size of a (device, stream tag) workspace: operand planes of the largest GEMM of a layer call (cfg4's dx product: 956 MB), split-K partials,
# BN / conv partial sums, hand-off tiles.  CTCN_WORKSPACE_MB sizes it for smaller models (a call that does not fit answers CTCN_EWORKSPACE or
# takes its plane-free path; nothing is silently truncated).  Allocated on first use of the tag, not at import.
WORKSPACE_BYTES = int(os.environ.get("CTCN_WORKSPACE_MB", "1024")) << 20


def workspace(device, nbytes=None, tag="main"):
    """Per-device scratch buffer (split-K partials, BN/conv partial sums).  Stream-ordered reuse: every
    library call consumes its partials before returning control to the same stream's next launch.  Calls issued on a
    second stream use their own buffer (tag).  Creation is serialised (the autograd thread and the main thread may both get here first)."""
    nbytes = WORKSPACE_BYTES if nbytes is None else nbytes
    key = (device.type, device.index, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        with _WS_LOCK:
            buf = _WS.get(key)
            if buf is None or buf.numel() < nbytes:
                buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
                _WS[key] = buf
    return buf
