#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4d; mkdir -p $O
export PYTHONPATH=$R
timeout 600 python tools/rollout_probe.py > $O/rollout_probe.txt 2>&1; cat $O/rollout_probe.txt | cut -c1-400
