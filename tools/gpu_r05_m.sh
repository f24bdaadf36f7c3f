#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_league.py tests/test_hip_learning.py tests/test_hip_dp.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
