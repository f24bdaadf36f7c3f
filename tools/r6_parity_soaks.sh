#!/bin/bash
# round 6: parity soaks on the final build - env step vs oracle over random configurations (six fresh seeds x 250), fused rollouts replayed through the oracle (600
# configurations), the fused update's gradient over random shapes (300); + the suites touched by the ordered one-launch step.   bash tools/r6_parity_soaks.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export PYTHONPATH=$R PYTHONUNBUFFERED=1; O=$R/gpurun_out/r06/soaks; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_groups.py tests/test_hip_vec_facade.py tests/test_hip_facade.py tests/test_bench_contract.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
{
echo "Final build of round 6 (k_step / k_policy_step / k_run_random with the episode-metric tally paths; the network's log-std columns): HIP vs oracle on random configurations, six fresh seeds x 250 configurations"
echo "(tests/test_hip_vs_oracle_batch.py -k random_configurations: random agent counts, balances, laws incl. trend, shuffled dict orders, prefilled 0-512-order books per side)"
for s in 61001 61002 61003 61004 61005 61006; do
  echo "CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s"
  CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_hip_vs_oracle_batch.py -q -m gpu -k "random_configurations" 2>&1 | tail -1
done
} > $O/fuzz_soak_1500_configurations.txt 2>&1
tail -13 $O/fuzz_soak_1500_configurations.txt
timeout 1200 python tools/rollout_soak.py --configs 600 --seed 62 --quiet 2>&1 | grep -v amdgpu.ids | tee $O/rollout_soak_600_configurations.txt | tail -3
timeout 1200 python tools/gradient_soak.py --configs 300 --seed 63 2>&1 | grep -v amdgpu.ids > $O/gradient_soak_300_shapes.txt; tail -2 $O/gradient_soak_300_shapes.txt
