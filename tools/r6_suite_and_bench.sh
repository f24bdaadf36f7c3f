#!/bin/bash
# round 6: GPU suite + smoke + the driver's bench command + the default bench (through gpurun): bash tools/r6_suite_and_bench.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
T=${1:-a}; O=$R/gpurun_out/r06/$T; mkdir -p $O
export PYTHONUNBUFFERED=1
(time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider) > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_suite.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/gpu_suite.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out $O/bench_ppo.json > /dev/null 2>&1
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for n in ("bench_driver_command", "bench_default"):
    d = json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][0])
    print(n, {k: round(v / 1e6, 1) for k, v in d.items() if k.startswith("value") and isinstance(v, (int, float))}, "frac", round(d["roofline"]["frac"], 4))
x = json.load(open(f"{O}/bench_ppo.json")); print("bench_ppo", round(x["value"] / 1e6, 1), "M")
PY
