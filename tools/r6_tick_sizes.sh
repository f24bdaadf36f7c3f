#!/bin/bash
# round 6: integer tick sizes on the HIP path - the reference-cut tick goldens, the facades, and the fuzz against the oracle with ticks in the random configurations
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R; O=$R/gpurun_out/r06/tick; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_golden.py tests/test_hip_facade.py tests/test_hip_kat.py tests/test_hip_selftest.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
for s in 64001 64002; do
  echo "CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s"
  CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_hip_vs_oracle_batch.py -q -m gpu -k "random_configurations" 2>&1 | tail -1
done 2>&1 | tee $O/fuzz_with_ticks.txt
