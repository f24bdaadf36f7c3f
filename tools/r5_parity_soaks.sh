#!/bin/bash
# round 5: the soaks kept under profiles/r05/ - env step vs oracle over random configurations (tools/r5_env_fuzz_soak.sh), fused rollouts (one policy / league, every
# history depth) replayed through the oracle, the fused update's gradient over random shapes.   Usage (through gpurun): bash tools/r5_parity_soaks.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/rollout_soak.py --configs 600 --seed 2 --quiet 2>&1 | grep -v amdgpu.ids | tee $O/rollout_soak_600.txt | tail -3
timeout 900 python tools/gradient_soak.py --configs 300 --seed 3 2>&1 | grep -v amdgpu.ids > $O/gradient_soak.txt; tail -1 $O/gradient_soak.txt
