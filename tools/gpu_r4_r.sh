#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; O=gpurun_out/r4r; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_mlp.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/mlp_bench.py --json $O/mlp_kernels.json 2>/dev/null | grep -E "adam_us|fused_us|wgrad_us"
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/bench_ppo.json 2>&1 | tail -2 | cut -c1-330
