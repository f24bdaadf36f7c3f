#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4h; mkdir -p $O
export PYTHONPATH=$R
timeout 1200 python -m pytest tests/test_hip_vec_facade.py tests/test_hip_facade.py tests/test_hip_mlp.py tests/test_hip_golden.py tests/test_hip_groups.py tests/test_hip_baseline_configs.py -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for rep in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('driver cmd: value %.1f noinfo %.1f policy_in_loop %.1f' % (d['value']/1e6, d['value_without_info']/1e6, d['value_policy_in_loop']/1e6))"
done
timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 64 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('long: value %.1f' % (d['value']/1e6))"
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/bench_ppo.json > $O/ppo.log 2>&1
python - <<PY
import json
d=json.load(open("$O/bench_ppo.json")); h=d["iterations"][2:]
print("e2e %.1f M agent-steps/s; rollout %.2f ms update %.2f ms" % (d["value"]/1e6, 1e3*sum(x["rollout_s"] for x in h)/len(h), 1e3*sum(x["update_s"] for x in h)/len(h)))
PY
