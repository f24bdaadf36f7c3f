#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
for c in 1 2 3 4; do
  timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 10 --chains $c --out $O/league_c$c.json > /dev/null 2>&1
done
CDA_MLP_ROLLOUT_MT=2 timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 10 --out $O/league_mt2.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 4096 --agents 8 --trainable 2 --episode 64 --iters 8 --out $O/league_4096.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 8192 --agents 8 --trainable 2 --episode 64 --iters 6 --out $O/league_8192.json > /dev/null 2>&1
python - <<'PY'
import json
for n in ("league_c1", "league_c2", "league_c3", "league_c4", "league_mt2", "league_4096", "league_8192"):
    try:
        d = json.load(open(f"gpurun_out/r05/{n}.json")); it = d["iterations"][2:]
        print(n, round(d["value"] / 1e6, 1), "M;  rollout ms", round(sum(h["rollout_s"] for h in it) / len(it) * 1e3, 2), " update ms", round(sum(h["update_s"] for h in it) / len(it) * 1e3, 2))
    except Exception as e:
        print(n, "missing", e)
PY
