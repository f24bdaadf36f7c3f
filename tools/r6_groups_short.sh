#!/bin/bash
# the headline leg of the driver's command (20 steps, median of 5 legs) by number of group chains, alternating, same box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06/groups_short; mkdir -p $O
for i in 1 2 3; do for g in 2 3 4 6 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --groups $g --no-cpu-baseline --no-policy-leg --no-league-leg --no-extra-legs > $O/g${g}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json
for g in (2, 3, 4, 6, 8):
    ds = [json.load(open(f"gpurun_out/r06/groups_short/g{g}_{i}.json")) for i in (1, 2, 3)]
    print(g, "chains:", [round(d["value"] / 1e6, 1) for d in ds], "M; legs min/max of the first run", round(4096 * 4 / ds[0]["timed_repeats"]["max"] / 1e3, 1), round(4096 * 4 / ds[0]["timed_repeats"]["min"] / 1e3, 1), "kernel_ms", round(ds[0]["roofline"]["kernel_ms"], 4))
PY
