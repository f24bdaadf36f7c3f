#!/bin/bash
# round 5, GPU call B: fixed + data-parallel tests, the dW2-fusion probe, k_step's wait attribution, RLlib-objective curve, league bench (no re-capture), record cost
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_league.py tests/test_hip_learning.py tests/test_hip_dp.py -q -m gpu -p no:cacheprovider -s > $O/tests_b.txt 2>&1; echo "tests rc=$?"; tail -25 $O/tests_b.txt | cut -c1-400
timeout 300 python tools/fb_wgrad_fusion_probe.py > $O/wgrad_fusion_experiment.txt 2>&1; echo "probe rc=$?"; cat $O/wgrad_fusion_experiment.txt
timeout 300 python tools/fb_wgrad_fusion_probe.py --rows 131072 --agents 1 > $O/wgrad_fusion_experiment_league_shape.txt 2>&1; tail -3 $O/wgrad_fusion_experiment_league_shape.txt
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 12 --out $O/bench_league.json > $O/bench_league.log 2>&1; echo "league rc=$?"
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 4096 --horizon 64 --iters 8 --out $O/bench_league_4096.json > $O/bench_league_4096.log 2>&1
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 8 --objective rllib --out $O/bench_league_rllib.json > $O/bench_league_rllib.log 2>&1
timeout 600 python tools/learning_curve.py --iters 40 --objective rllib --no-legacy > $O/learning_curve_rllib.txt 2>&1; tail -1 $O/learning_curve_rllib.txt
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 8 --objective rllib --out $O/bench_ppo_rllib.json > $O/bench_ppo_rllib.log 2>&1
python - <<'PY'
import json
for n in ("bench_league", "bench_league_4096", "bench_league_rllib", "bench_ppo_rllib"):
    try:
        d = json.load(open(f"gpurun_out/r05/{n}.json")); it = d["iterations"]
        print(n, round(d["value"] / 1e6, 1), "M;  rollout ms", [round(h["rollout_s"] * 1e3, 2) for h in it], " update ms", [round(h["update_s"] * 1e3, 2) for h in it], d.get("flagged_markets"), d.get("nav_conservation_violations"))
    except Exception as e:
        print(n, "missing", e)
PY
# k_step: phase cycles + the SQ's view of the waits
timeout 600 python tools/phase_timing.py > $O/phase_timing.txt 2>&1; tail -12 $O/phase_timing.txt | cut -c1-1200
export TMPDIR=/tmp CDA_BENCH_PRIMER_MS=0; cd /tmp
BENCH="python $R/bench.py --steps 200 --warmup 16 --repeats 1 --no-cpu-baseline --no-extra-legs --no-policy-leg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/kw/pmc_w1 -o p -- $BENCH > /dev/null 2> $O/kw_1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $O/kw/pmc_w2 -o p -- $BENCH > /dev/null 2> $O/kw_2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $O/kw/pmc_w3 -o p -- $BENCH > /dev/null 2> $O/kw_3.err
cd $R
python tools/kstep_wait_attribution.py $O/kw > $O/kstep_wait_counters.txt 2>&1; cat $O/kstep_wait_counters.txt
rm -rf $O/kw
