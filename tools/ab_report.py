"""means of the value_* legs per build over the runs of tools/ab_bench.sh (gpurun_out/ab_<build>_<steps>_<i>.json)"""
import glob
import json
import statistics
import sys
rows = {}
for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out") + "/ab_*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        print(f, "unreadable")
        continue
    rows.setdefault(f.split("/")[-1].rsplit("_", 1)[0], []).append(d)
base = {}
for key, ds in sorted(rows.items(), key=lambda kv: (kv[0].split("_")[-1], kv[0] != "ab_head_" + kv[0].split("_")[-1])):
    ks = [k for k in ds[0] if k.startswith("value") and all(d.get(k) for d in ds)]
    m = {k: statistics.mean(d[k] for d in ds) for k in ks}
    steps = key.split("_")[-1]
    if "head" in key:
        base[steps] = m
    ref = base.get(steps, {})
    print(f"{key:22s}", "  ".join(f"{k[6:] or 'value'}={m[k] / 1e6:.1f}" + (f"({(m[k] / ref[k] - 1) * 100:+.1f}%)" if k in ref and 'head' not in key else "") for k in ks))
