#!/bin/bash
# parity soak on round 5's final build: HIP vs oracle on random configurations, six fresh seeds x 250 configurations; + the auto-reset / capture paths on many short episodes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; O=gpurun_out/r05; mkdir -p $O
{
echo "Final build of round 5 (episode-end capture in the cold reset paths of k_step; k_policy_step beside it in the same translation unit): HIP vs oracle on random configurations, six fresh seeds x 250 configurations"
echo "(tests/test_hip_vs_oracle_batch.py -k random_configurations: random agent counts, balances, laws incl. trend, shuffled dict orders, prefilled 0-512-order books per side)"
for s in 53001 53002 53003 53004 53005 53006; do
  echo "CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s"
  CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_hip_vs_oracle_batch.py -q -m gpu -k "random_configurations" 2>&1 | tail -1
done
} > $O/fuzz_soak.txt 2>&1
tail -13 $O/fuzz_soak.txt
