#!/bin/bash
# round 5: the evidence kept under profiles/r05/ - GPU suite + smoke, the driver's bench command, rocprofv3 kernel stats of it and of the league loop,
# the record-cost probe, the data-parallel learner's one-rank line.   Usage (through gpurun): bash tools/r5_round_end_evidence.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05/evidence; mkdir -p $O
export PYTHONUNBUFFERED=1
(time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider) > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_suite.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/gpu_suite.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --learner dp > $O/bench_learner_dp_one_rank.json 2> /dev/null
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out $O/bench_ppo.json > /dev/null 2>&1
CDA_POLICY_STEP=0 timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out $O/bench_ppo_two_launches_per_step.json > /dev/null 2>&1
for h in 1 2 8; do timeout 300 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --n-hist $h --out $O/bench_ppo_n_hist$h.json > /dev/null 2>&1; done
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 12 --out $O/bench_league.json > /dev/null 2>&1
timeout 600 python tools/record_cost_probe.py > $O/record_cost.txt 2>&1; grep -v amdgpu $O/record_cost.txt
export TMPDIR=/tmp PYTHONPATH=$R; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_league -o l -- python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 8 > $O/prof_league.log 2>&1
f=$(find $O/prof_league -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_league.csv; rm -rf $O/prof_league
cd $R
python tools/kernel_stats_table.py $O/kernel_stats.csv 10; python tools/kernel_stats_table.py $O/kernel_stats_league.csv 16
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/evidence/bench_driver_command.json") if l.startswith("{")][0])
print("driver command: value", round(d["value"] / 1e6, 1), "M; policy in loop", round((d.get("value_policy_in_loop") or 0) / 1e6, 1), "league", round((d.get("value_league_self_play") or 0) / 1e6, 1), "roofline frac", round(d["roofline"]["frac"], 4), "cpu", d.get("cpu_baseline", {}).get("value"))
for n in ("bench_ppo", "bench_ppo_two_launches_per_step", "bench_league"):
    x = json.load(open(f"gpurun_out/r05/evidence/{n}.json")); print(n, round(x["value"] / 1e6, 1), "M")
x = json.loads([l for l in open("gpurun_out/r05/evidence/bench_learner_dp_one_rank.json") if l.startswith("{")][0]); print("learner dp one rank", round(x["value"] / 1e6, 1), "M")
PY
