#!/bin/bash
# round 6: the state-dependent log-std head - its tests, the suites it touches, and learning curves with it (through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r06/sd; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_policy_step.py tests/test_hip_league.py tests/test_hip_mlp.py tests/test_hip_hist.py -x -q -m gpu -p no:cacheprovider > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -12 $O/tests.txt
timeout 600 python tools/learning_curve.py --no-legacy --log-std-head > $O/curve_sd.txt 2>&1; tail -1 $O/curve_sd.txt
timeout 600 python tools/learning_curve.py --no-legacy --log-std-head --objective rllib > $O/curve_sd_rllib.txt 2>&1; tail -1 $O/curve_sd_rllib.txt
timeout 600 python tools/learning_curve.py --no-legacy > $O/curve_free.txt 2>&1; tail -1 $O/curve_free.txt
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --log-std-head --out $O/bench_ppo_sd.json > /dev/null 2>&1; python -c "import json; print('ppo sd', json.load(open('$O/bench_ppo_sd.json'))['value']/1e6)"
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out $O/bench_ppo.json > /dev/null 2>&1; python -c "import json; print('ppo', json.load(open('$O/bench_ppo.json'))['value']/1e6)"
