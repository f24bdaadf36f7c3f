#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ppo -o ppo -- python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 6 > $O/prof_ppo.log 2>&1
f=$(find $O/prof_ppo -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_ppo.csv; head -30 $O/kernel_stats_ppo.csv | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-160
