#!/bin/bash
# round 5, GPU call A: the league on the fused kernels - tests, bench line, learning curves; then the whole GPU suite and the PPO line (regressions)
set -u
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_league.py tests/test_hip_learning.py -q -m gpu -p no:cacheprovider -s > gpurun_out/r05/tests_league.txt 2>&1; echo "league tests rc=$?" | tee -a gpurun_out/r05/tests_league.txt
tail -40 gpurun_out/r05/tests_league.txt
timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 10 --out gpurun_out/r05/bench_league.json > gpurun_out/r05/bench_league.log 2>&1; echo "league bench rc=$?"
tail -3 gpurun_out/r05/bench_league.log | cut -c1-1500
timeout 600 python tools/learning_curve.py --iters 40 > gpurun_out/r05/learning_curve.txt 2>&1; echo "curve rc=$?"; tail -2 gpurun_out/r05/learning_curve.txt | cut -c1-3000
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_hip_league.py --deselect tests/test_hip_learning.py > gpurun_out/r05/gpu_suite_a.txt 2>&1; echo "suite rc=$?"; tail -15 gpurun_out/r05/gpu_suite_a.txt
timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 8 --out gpurun_out/r05/bench_ppo_a.json > gpurun_out/r05/bench_ppo_a.log 2>&1; echo "ppo rc=$?"; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r05/bench_ppo_a.json')); print('ppo value', d['value'], [round(h['rollout_s']*1e3,2) for h in d['iterations']], [round(h['update_s']*1e3,2) for h in d['iterations']])
    print([ (h['mean_reward'], h['episode_return']) for h in d['iterations']])
except Exception as e: print('no ppo json', e)
try:
    d=json.load(open('gpurun_out/r05/bench_league.json')); print('league value', d['value'], [round(h['rollout_s']*1e3,2) for h in d['iterations']], [round(h['update_s']*1e3,2) for h in d['iterations']]); print(d['champions'])
except Exception as e: print('no league json', e)
PY
