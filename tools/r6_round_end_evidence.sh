#!/bin/bash
# round 6: the evidence kept under profiles/r06/ - GPU suite + smoke, the driver's bench command, the default / c4 / learner-dp lines, PPO (plain, RLlib objective, log-std head),
# league, rocprofv3 kernel stats of the driver's command and of the league loop, the PMC passes of the headline shape on THIS build.   bash tools/r6_round_end_evidence.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r06/evidence; mkdir -p $O
export PYTHONUNBUFFERED=1
(time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider) > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_suite.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/gpu_suite.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"
timeout 600 python bench.py > $O/bench_default_1000_steps.json 2> $O/bench_default.err
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-league-leg > $O/bench_config_2048x8.json 2> /dev/null
timeout 600 python bench.py --learner dp > $O/bench_learner_dp_one_rank.json 2> /dev/null
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out $O/bench_ppo.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --objective rllib --out $O/bench_ppo_rllib_objective.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --log-std-head --objective rllib --out $O/bench_ppo_rllib_objective_log_std_head.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 8192 --agents 8 --trainable 2 --episode 64 --iters 10 --out $O/bench_league_8192x8.json > /dev/null 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 10 --out $O/bench_league_2048x8.json > /dev/null 2>&1
export TMPDIR=/tmp PYTHONPATH=$R; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_league -o l -- python -m gym_continuousdoubleauction_amd.league_train --fused --markets 8192 --agents 8 --trainable 2 --episode 64 --iters 8 > $O/prof_league.log 2>&1
f=$(find $O/prof_league -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_league.csv; rm -rf $O/prof_league
cd $R
python tools/kernel_stats_table.py $O/kernel_stats.csv 10 | tee $O/rocprof_summary.txt; python tools/kernel_stats_table.py $O/kernel_stats_league.csv 16 | tee -a $O/rocprof_summary.txt
BENCH_STEPS=200 BENCH_EXTRA="--groups 2 --no-policy-leg" bash tools/profile_gpu.sh r06_g2 > $O/pmc_summary_4096x4_info1_g2.txt 2>&1
BENCH_STEPS=200 BENCH_EXTRA="--groups 4 --no-policy-leg" bash tools/profile_gpu.sh r06_g4 > $O/pmc_summary_4096x4_info1_g4.txt 2>&1
cp gpurun_out/prof_r06_g2/*.json gpurun_out/prof_r06_g4/*.json $O/ 2>/dev/null
for d in gpurun_out/prof_r06_g2 gpurun_out/prof_r06_g4; do rm -rf $d/trace $d/pmc_* $d/calib_*; done
python - <<'PY'
import json
O = "gpurun_out/r06/evidence/"
one = lambda n: json.loads([l for l in open(O + n) if l.startswith("{")][0])
for n in ("bench_driver_command.json", "bench_default_1000_steps.json", "bench_config_2048x8.json"):
    d = one(n)
    print(n, {k: round(v / 1e6, 1) for k, v in d.items() if k.startswith("value") and isinstance(v, (int, float))}, "frac", round(d["roofline"]["frac"], 4), "cpu", (d.get("cpu_baseline") or {}).get("value"))
for n in ("bench_ppo", "bench_ppo_rllib_objective", "bench_ppo_rllib_objective_log_std_head", "bench_league_8192x8", "bench_league_2048x8"):
    x = json.load(open(O + n + ".json")); print(n, round(x["value"] / 1e6, 1), "M")
print("learner dp one rank", round(one("bench_learner_dp_one_rank.json")["value"] / 1e6, 1), "M")
PY
