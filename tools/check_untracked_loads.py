#!/usr/bin/env python3
"""Static guard for the hand-waited reserve loads of rnn_bwd_scatter (csrc/rnn.hip: ld_slab_untracked + `s_waitcnt vmcnt(7)`).

hipcc does not know that the registers written by those inline-asm buffer loads are in flight, so nothing stops it from
reading, copying or re-using one of them before the hand-placed wait.  This script compiles rnn.hip to assembly and, for every
rnn_bwd_scatter instantiation, walks the instruction stream after each such load up to the next `s_waitcnt vmcnt(N <= 7)`;
any instruction that mentions the destination register in between is reported.  usage: tools/check_untracked_loads.py
(exit status 1 on a finding; needs hipcc, no GPU)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ctc_pytorch_amd", "csrc", "rnn.hip")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "rnn.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only", "-o", out, src])
    lines = open(out).read().split("\n")

bad = 0
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*rnn_bwd_scatter.*:", l)]
for i0 in starts:
    name = lines[i0].split(":")[0]
    j = i0
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    ins = [b.strip() for b in lines[i0:j] if b.startswith("\t") and not b.strip().startswith((";", "."))]
    is_wait = [bool(re.match(r"s_waitcnt vmcnt\(([0-7])\)", t)) for t in ins]
    loads = [(k, re.match(r"buffer_load_dword (v\d+),", t).group(1)) for k, t in enumerate(ins)
             if re.match(r"buffer_load_dword v\d+, v\d+, s\[\d+:\d+\], s\d+ offen", t)]
    found = 0
    for k, reg in loads:
        m = k + 1
        while m < len(ins) and not is_wait[m]:
            t = ins[m]
            if re.search(r"\b" + reg + r"\b", t) and not t.startswith("buffer_load_dword " + reg + ","):
                print("%s: %s written by the load at #%d is touched at #%d before a wait: %s" % (name[-44:], reg, k, m, t[:90]))
                found += 1
                break
            m += 1
    print("%-48s hand-waited loads %3d, findings %d" % (name[-48:], len(loads), found))
    bad += found
sys.exit(1 if bad else 0)
