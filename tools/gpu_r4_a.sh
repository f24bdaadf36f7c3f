#!/bin/bash
# round 4, call A: the MFMA network kernels - conventions, numerics, rollout chains; then first timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_mlp.py -x -q -m gpu > $O/test_mlp.log 2>&1; echo "rc=$?" >> $O/test_mlp.log
tail -30 $O/test_mlp.log
for mt in 4 2; do
  CDA_MLP_MT=$mt timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 6 --out $O/ppo_fused_mt$mt.json > $O/ppo_fused_mt$mt.log 2>&1
  tail -3 $O/ppo_fused_mt$mt.log | cut -c1-400
done
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 6 --no-graphs --out $O/ppo_fused_nographs.json > $O/ppo_fused_nographs.log 2>&1
tail -2 $O/ppo_fused_nographs.log | cut -c1-300
