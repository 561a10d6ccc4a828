#!/bin/bash
# A library variant for A/B sessions on one GPU box (tools/gpu_session.sh 23-29): the product's objects with rnn.hip recompiled with extra flags.
#   tools/build_variant_lib.sh single -DCTCN_RED_SINGLE     -> tools/libctcn_single.so: ONE set of parked partial tiles in rnn_fwd_tagged (rounds 2-5);
#                                                              test_rnn_fwd_tagged_with_slow_item_waves fails on it (profiles/r06_divergence_root_cause.txt)
#   tools/build_variant_lib.sh base                         -> tools/libctcn_base.so: the tree as it is (the comparison build of an A/B)
# Run `python -c 'import __graft_entry__ as g; g.build()'` first (the other objects come from ctc_pytorch_amd/csrc/_obj).
set -eu
cd "$(dirname "$0")/.."; name=$1; shift
O=ctc_pytorch_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c ctc_pytorch_amd/csrc/rnn.hip -o /tmp/rnn_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/libctcn_$name.so $O/comm.o $O/conv.o $O/core.o $O/ctc.o $O/decode.o $O/diag.o $O/elementwise.o $O/gemm.o $O/hostjoin.o $O/norm.o $O/pool.o /tmp/rnn_$name.o
ls -la tools/libctcn_$name.so
