#!/bin/bash
# sample records: network tests + configs[4] test + PPO e2e + a rocprof pass over the PPO loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4l; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_mlp.py tests/test_hip_baseline_configs.py -q -m gpu -x > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/bench_ppo.json > $O/ppo.log 2>&1; tail -3 $O/ppo.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ppo -- python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 6 > $O/prof.log 2>&1
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
