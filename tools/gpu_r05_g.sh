#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_hist.py -q -m gpu -p no:cacheprovider 2>&1 | tail -30 | cut -c1-300
for h in 1 2 4 8; do timeout 300 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --n-hist $h --out $O/ppo_h$h.json > /dev/null 2>&1; done
python - <<'PY'
import json, statistics
for h in (1, 2, 4, 8):
    try:
        d = json.load(open(f"gpurun_out/r05/ppo_h{h}.json")); it = d["iterations"][2:]
        r, u = statistics.median(x["rollout_s"] for x in it) * 1e3, statistics.median(x["update_s"] for x in it) * 1e3
        print("n_hist", h, "median rollout ms", round(r, 2), " update ms", round(u, 2), " -> ", round(4096 * 4 * 64 / (r + u) / 1e3, 1), "M agent-steps/s")
    except Exception as e:
        print(h, "missing", e)
PY
