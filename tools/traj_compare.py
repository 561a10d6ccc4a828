import base64
import numpy as np
"""Reads CTCN_TRAJ_LOG files (tools/squat_stress.run with trace=True) and, per workload, reports every run whose trace differs from the most
frequent one: the first (step, tensor) that differs in execution order.  usage: traj_compare.py <log.jsonl> ..."""
import collections, json, sys
runs = collections.defaultdict(list)
for path in sys.argv[1:]:
    for k, line in enumerate(open(path)):
        d = json.loads(line)
        if d.get("trace"):
            runs[(d["workload"], d["steps"])].append((path.split("/")[-1], k, d))
for (wl, steps), rs in sorted(runs.items()):
    keyed = collections.Counter(json.dumps(d["trace"]) for _, _, d in rs)
    ref = json.loads(keyed.most_common(1)[0][0])
    print("%s x %d steps: %d traced runs, %d distinct traces" % (wl, steps, len(rs), len(keyed)))
    for path, k, d in rs:
        step, first = 0, None
        for (na, va), (nb, vb) in zip(d["trace"], ref):
            if na == "end-of-step":
                step += 1
            elif va != vb:
                first = (step, na)
                break
        if first:
            print("  %s run %d (squat %s): first difference at step %d in %r; fallback shapes %r" % (path, k, d["squat"], first[0], first[1], d["state_ran_in"]["fallback_shapes"]))
            # every traced tensor of that step that differs, in execution order
            step, names = 0, []
            for (na, va), (nb, vb) in zip(d["trace"], ref):
                if na == "end-of-step":
                    step += 1
                elif step == first[0] and va != vb:
                    names.append(na)
            print("    differing tensors of that step, in order: %s" % names[:14])
            ad = d.get("addresses")
            if ad:
                L = max(a["layer"] for a in ad) + 1
                a0 = ad[first[0] * L] if first[0] * L < len(ad) else None          # layer 0's call of the step
                if a0:
                    print("    layer-0 buffers of that step: " + ", ".join("%s=0x%x" % (kk, vv) for kk, vv in sorted(a0.items()) if kk not in ("layer", "I")))
                ref_d = next((dd for _, _, dd in rs if json.dumps(dd["trace"]) == keyed.most_common(1)[0][0] and dd.get("layer0_per_timestep")), None)
                if ref_d and d.get("layer0_per_timestep"):
                    for kk in ("gates", "y"):
                        a_ = np.frombuffer(base64.b64decode(d["layer0_per_timestep"][kk]), dtype=np.int64).reshape(d["steps"], -1, 2)
                        b_ = np.frombuffer(base64.b64decode(ref_d["layer0_per_timestep"][kk]), dtype=np.int64).reshape(d["steps"], -1, 2)
                        bad = np.argwhere(a_[first[0]] != b_[first[0]])
                        if len(bad):
                            for dr in (0, 1):
                                ts = bad[bad[:, 1] == dr][:, 0]
                                if len(ts):
                                    print("    layer 0 %s, direction %d: %d of %d timesteps differ at that step: first t=%d last t=%d  %s" % (
                                        kk, dr, len(ts), a_.shape[1], ts.min(), ts.max(), ts[:12].tolist()))
                same = [a for a in ad if a["layer"] == 0]
                print("    layer-0 gates addresses over the run: " + " ".join("0x%x" % a["gates"] for a in same))
