#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/pmc_probe.py):
   pmc_to_json.py fetch.db write.db out.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-B requests of wide coalesced reads at 64 B); both counters are KiB."""
import json, sqlite3, sys

ALGO = [   # (substring of the kernel name, label, algorithmic bytes per launch at the probe shapes -- DESIGN.md section 5); first match wins
    ("rnn_fwd_tagged", "rnn_fwd_tagged cfg2 layer (T=800,B=32,H=320,D=2)", 655360000),
    ("rnn_fwd_persist", "rnn_fwd_persist cfg2 layer (T=800,B=32,H=320,D=2)", 655360000),
    ("rnn_bwd_scatter2", "rnn_bwd_scatter2 ref_yaml layer (T=200,B=8,H=384,D=2)", 200 * 8 * 2 * 384 * 4 * 10),
    ("rnn_bwd_scatter", "rnn_bwd_scatter cfg2 layer (T=800,B=32,H=320,D=2)", 655360000),
    ("gemm_planes_nt256pp_af32_kernel<2>", "gemm 25600x1280x640 (probe of bench.py; A = f32 split while staged, ping-pong tile)", 25600 * 640 * 4 + 1280 * 640 * 4 + 25600 * 1280 * 4),
    ("gemm_planes_nt256pp_af32_kernel<1>", "gemm 25600x640x2560 (dx of the layer; A = f32 d(pre-act) split while staged, ping-pong tile)", 25600 * 2560 * 4 + 640 * 2560 * 4 + 25600 * 640 * 4),
    ("gemm_planes_nt256_af32_kernel<2>", "gemm 25600x1280x640 (probe of bench.py; A = f32 split while staged)", 25600 * 640 * 4 + 1280 * 640 * 4 + 25600 * 1280 * 4),
    ("gemm_planes_nt256_af32_kernel<1>", "gemm 25600x640x2560 (dx of the layer; A = f32 d(pre-act) split while staged)", 25600 * 2560 * 4 + 640 * 2560 * 4 + 25600 * 640 * 4),
    ("gemm_tn_f32_pp_kernel", "gemm_tn_f32_pp_kernel (weight gradients of the probe layer: dW_ih 1280x640x25600 and dW_hh 1280x320x25568, both directions, averaged; algorithmic bytes of dW_ih: both operands once + the output)", 25600 * (1280 + 640) * 4 + 1280 * 640 * 4),
    ("gemm_planes_nt256pp_kernel<2>", "gemm_planes_nt256pp_kernel<2> (256 x 256 ping-pong tile; bf16 planes in, f32 out)", None),
    ("gemm_planes_nt256_kernel", "gemm_planes_nt256_kernel (bf16 planes in, f32 out)", None),
    ("split_rows_kernel", "split_rows_kernel", None),
    ("dropout_kernel", "dropout_kernel 25600x640", 131072000),
    ("bn_apply_", "bn_apply (rows of channels: four channels per lane) 25600x640", 131072000),
    ("bn_dx_", "bn_dx (rows of channels: four channels per lane) 25600x640", 196608000),
    ("beam_fast_kernel", "beam_fast_kernel cfg5 peaky (128 x 800 x 62, W=20): reads ln p (double) of the processed frames", None),
    ("beam_prep_kernel", "beam_prep_kernel cfg5 (128 x 800 x 62): lp f32 in, ln p f64 + p_blank + flags out", 800 * 128 * 62 * 12 + 800 * 128 * 5),
    # x (32,32,800,20) = 65.536 MB, y = dy (32,32,400,10) = 16.384 MB.  The counters are averaged over the kernel's five launches of a forward + backward
    # pass: forward (x in, y out) and four dgrad stride classes (dy in, a quarter of dx out each) -> (81.92 + 4 * 32.768) / 5 MB
    ("conv_mfma_kernel", "conv_mfma_kernel cfg3 layer 2 (32 -> 32, 3x3, stride 2x2; B=32, T=800): forward and the four dgrad classes, averaged", (65536000 + 16384000 + 4 * (16384000 + 16384000)) // 5),
    ("conv_wgrad_mfma_kernel", "conv_wgrad_mfma_kernel cfg3 layer 2 (dy and x in, per-chunk partials out)", 65536000 + 16384000 + 32 * 32 * 9 * 4),
]


def table(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg, dur in cur.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        out[name] = (n, avg, dur / 1e3)
    return out


fetch, write = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over tools/pmc_probe.py, precision 1; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 counts the 128-B requests of wide coalesced reads as 64 B)"}
if len(sys.argv) > 5:      # a second pair of passes (PMC_PROBE_SET=scatter2) merged into the same table
    f2, w2 = table(sys.argv[4], "FETCH_SIZE"), table(sys.argv[5], "WRITE_SIZE")
    fetch.update({k: v for k, v in f2.items() if "scatter2" in k})
    write.update({k: v for k, v in w2.items() if "scatter2" in k})
for name in sorted(set(fetch) | set(write)):
    hit = next((e for e in ALGO if e[0] in name and "queue" not in name), None)
    if hit is None:
        continue
    f = fetch.get(name, (0, 0.0, 0.0))
    w = write.get(name, (0, 0.0, 0.0))
    label, algo = hit[1], hit[2]
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    res["%s [%s]" % (label, short)] = {"launches": f[0], "fetch_kib_raw": round(f[1], 1), "write_kib": round(w[1], 1),
                                       "hbm_bytes": int(2 * f[1] * 1024 + w[1] * 1024), "algorithmic_bytes": algo, "avg_us": round(f[2], 1)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
