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
