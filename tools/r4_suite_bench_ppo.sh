#!/bin/bash
# full GPU suite + the driver's bench command + PPO e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4e; mkdir -p $O
export PYTHONPATH=$R
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -15 $O/gpu_suite.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-1500 $O/bench_driver.json
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/bench_ppo.json > $O/ppo.log 2>&1; tail -1 $O/ppo.log | cut -c1-300
