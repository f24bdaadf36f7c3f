#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r06/sd; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_policy_step.py tests/test_hip_league.py tests/test_hip_mlp.py tests/test_hip_hist.py tests/test_hip_learning.py tests/test_hip_dp.py tests/test_hip_baseline_configs.py -q -m gpu -p no:cacheprovider > $O/tests2.txt 2>&1; echo "tests rc=$?"; tail -12 $O/tests2.txt
