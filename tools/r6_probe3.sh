#!/bin/bash
# Run ON THE GPU BOX (round 6): k_policy_step with windowed sampling / coalesced record stores: parity, phases, the short and long policy legs
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_hip_policy_step.py tests/test_hip_mlp.py tests/test_hip_league.py tests/test_hip_hist.py tests/test_hip_episode_metrics.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/rollout_tests.txt
python tools/policy_leg_probe.py > gpurun_out/policy_leg_probe_after2.jsonl 2> gpurun_out/policy_leg_probe.err
CDA_HIP_LIB=$R/gpurun_ab/libcda_hip_phase.so python tools/policy_step_phases.py --chains 1 > gpurun_out/policy_step_phases_1chain_b.txt 2>&1
CDA_HIP_LIB=$R/gpurun_ab/libcda_hip_phase.so python tools/policy_step_phases.py --chains 4 > gpurun_out/policy_step_phases_4chains_b.txt 2>&1
