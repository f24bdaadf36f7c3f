#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
for mt in 4 2; do
  (cd /tmp && CDA_MLP_MT=$mt timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_mt$mt -o ppo -- python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 5 > $O/prof_mt$mt.log 2>&1)
  f=$(find $O/prof_mt$mt -name "*kernel_stats.csv" | head -1)
  echo "== MT=$mt  $f"; head -25 "$f" | cut -c1-200
done
