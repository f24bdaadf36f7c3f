#!/bin/bash
# round-end evidence: rocprof kernel stats of the PPO loop and of the driver's bench command, PMC passes of the headline shape
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
bash tools/r4_rocprof_stats.sh
cd $R
BENCH_STEPS=200 BENCH_EXTRA="--groups 2" bash tools/profile_gpu.sh r04_g2 > gpurun_out/prof_r04_g2_summary.txt 2>&1
tail -30 gpurun_out/prof_r04_g2_summary.txt
ls gpurun_out/prof_r04_g2 | head -30
