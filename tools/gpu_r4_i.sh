#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4i; mkdir -p $O
export PYTHONPATH=$R
for wv in 8 4; do
  CDA_MLP_WAVES=$wv timeout 600 python -m pytest tests/test_hip_mlp.py -q -m gpu > $O/test_mlp_w$wv.log 2>&1; echo "waves=$wv: $(tail -1 $O/test_mlp_w$wv.log)"
  CDA_MLP_WAVES=$wv timeout 300 python tools/mlp_bench.py --json $O/bench_w$wv.json > $O/bench_w$wv.log 2>&1; python -c "
import json; d=json.load(open('$O/bench_w$wv.json')); print('waves=$wv', {k: round(v,1) for k,v in d.items() if k.endswith('_us')}, {k: round(v) for k,v in d['useful_tflops'].items()})"
  CDA_MLP_WAVES=$wv timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/ppo_w$wv.json > $O/ppo_w$wv.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/ppo_w$wv.json")); h=d["iterations"][2:]
print("waves=$wv e2e %.1f M agent-steps/s; rollout %.2f ms update %.2f ms" % (d["value"]/1e6, 1e3*sum(x["rollout_s"] for x in h)/len(h), 1e3*sum(x["update_s"] for x in h)/len(h)))
PY
done
