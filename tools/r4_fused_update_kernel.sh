#!/bin/bash
# the fused forward + loss + backward kernel: tests, kernel timings, cycle stamps, PPO e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4n; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_mlp.py -q -m gpu -x > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python tools/mlp_bench.py --json $O/mlp_kernels.json > $O/mlp_bench.log 2>&1; tail -32 $O/mlp_bench.log
timeout 120 python tools/mlp_timing.py --fused --block 300 > $O/stamps_fb8.txt 2>&1; cat $O/stamps_fb8.txt
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/bench_ppo.json > $O/ppo.log 2>&1; tail -2 $O/ppo.log | cut -c1-400
