#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/pmc_probe.py):
   pmc_to_json.py fetch.db write.db out.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-B requests of wide coalesced reads at 64 B); both counters are KiB."""
import json, sqlite3, sys

ALGO = {   # algorithmic bytes per launch at the probe shapes
    "rnn_fwd_persist": ("rnn_fwd_persist cfg2 layer (T=800,B=32,H=320,D=2)", 655360000),
    "rnn_bwd_scatter": ("rnn_bwd_scatter cfg2 layer (T=800,B=32,H=320,D=2)", 655360000),
    "gemm_planes_nt256_kernel": ("gemm_planes_nt256_kernel 25600x1280x640 (bf16 planes in, f32 out)", 25600 * 640 * 4 + 1280 * 640 * 4 + 25600 * 1280 * 4),
    "gemm_planes_nt256_af32_kernel": ("gemm_planes_nt256_af32_kernel 25600x1280x640 (f32 A in, f32 out)", 25600 * 640 * 4 + 1280 * 640 * 4 + 25600 * 1280 * 4),
    "split_rows_kernel": ("split_rows_kernel", None),
    "dropout_kernel": ("dropout_kernel 25600x640", 131072000),
    "bn_apply_kernel": ("bn_apply_kernel 25600x640", 131072000),
    "bn_dx_kernel": ("bn_dx_kernel 25600x640", 196608000),
    "beam_fast_kernel": ("beam_fast_kernel cfg5 peaky (128 x 800 x 62, W=20): reads ln p (double) of the processed frames", None),
    "beam_prep_kernel": ("beam_prep_kernel cfg5 (128 x 800 x 62): lp f32 in, ln p f64 + p_blank + flags out", 800 * 128 * 62 * 12 + 800 * 128 * 5),
}


def table(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg, dur in cur.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        out[name] = (n, avg, dur / 1e3)
    return out


fetch, write = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over tools/pmc_probe.py, precision 1; FETCH_SIZE doubled per "
                 "MI355X_MICROARCH.md (gfx950 counts the 128-B requests of wide coalesced reads as 64 B); round 2"}
for name in sorted(set(fetch) | set(write)):
    key = next((k for k in ALGO if k + "<" in name or k + "(" in name or name.endswith(k) or (k in name and "queue" not in name)), None)
    if key is None:
        continue
    f = fetch.get(name, (0, 0.0, 0.0))
    w = write.get(name, (0, 0.0, 0.0))
    label, algo = ALGO[key]
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    res["%s [%s]" % (label, short)] = {"launches": f[0], "fetch_kib_raw": round(f[1], 1), "write_kib": round(w[1], 1),
                                       "hbm_bytes": int(2 * f[1] * 1024 + w[1] * 1024), "algorithmic_bytes": algo, "avg_us": round(f[2], 1)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
