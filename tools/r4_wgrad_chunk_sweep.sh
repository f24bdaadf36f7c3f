#!/bin/bash
# weight-gradient kernel: chunk-count sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4o; mkdir -p $O
export PYTHONPATH=$R
for c in 25 51 64 102 204; do
  timeout 200 python tools/mlp_bench.py --chunks $c --iters 30 2>/dev/null | grep -E '"chunks"|wgrad_us|adam_us|minibatch_step_fused_us' | tr -d '\n'; echo
done | tee $O/chunk_sweep.txt
