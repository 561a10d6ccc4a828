#!/bin/bash
# HBM-side traffic (L2 <-> fabric) of one recurrent layer's launches at the workloads' shapes: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in
# separate passes over tools/pmc_probe.py (PMC_PROBE_SET=recurrence), per-kernel averages by tools/pmc_dump.py.  KiB; FETCH_SIZE to be doubled per
# MI355X_MICROARCH.md (gfx950).  usage (on the GPU box): bash tools/rnn_traffic_by_shape.sh > profiles/r06_rnn_traffic_by_shape.txt
set -u
cd "$(dirname "$0")/.."; R=$PWD; O=${TMPDIR:-/tmp}/rnn_traffic_$$; mkdir -p $O
echo "# HBM-side traffic of one recurrent launch per shape (rocprofv3 --pmc, separate passes, KiB; FETCH_SIZE x 2 = bytes fetched on gfx950); algorithmic bytes per launch:"
echo "#   forward T*B*D*(G+2)*H*4 (pre-activations in, activations + c / W_hn h + y out), backward T*B*D*(2G+2)*H*4 -- cfg4 layer (T 1200, B 64, H 512, GRU): 1.18 / 1.57 GB ... see DESIGN.md section 5"
for shape in "1200 64 512 gru" "800 32 320 lstm" "400 8 384 lstm"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== T=$1 B=$2 H=$3 $4 $c"
    ( cd /tmp && PMC_PROBE_SET=recurrence PMC_T=$1 PMC_B=$2 PMC_H=$3 PMC_CELL=$4 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/p -o p -- python $R/tools/pmc_probe.py > $O/log 2>&1 )
    db=$(find $O/p -name "*.db" | head -1)
    [ -n "$db" ] && python tools/pmc_dump.py $db rnn_ 2>/dev/null | cut -c1-170
    rm -rf $O/p
  done
done
rm -rf $O
