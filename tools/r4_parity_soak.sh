#!/bin/bash
# parity soak on the round's final build: HIP vs oracle on random configurations, fresh seeds; + the PPO loop at 2048 x 8 (BASELINE configs[3]'s shape)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; O=gpurun_out/r4soak; mkdir -p $O
{
echo "Final build of round 4 (auto reset inside the info-less step kernel, lane-offset shifts masked): HIP vs oracle on random configurations, six fresh seeds x 250 configurations"
echo "(tests/test_hip_vs_oracle_batch.py -k random_configurations: random agent counts, balances, laws incl. trend, shuffled dict orders, prefilled 0-512-order books per side)"
for s in 41001 41002 41003 41004 41005 41006; do
  echo "CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s"
  CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_hip_vs_oracle_batch.py -q -m gpu -k "random_configurations" 2>&1 | tail -1
done
} > $O/fuzz_soak.txt 2>&1
tail -13 $O/fuzz_soak.txt
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 2048 --agents 8 --horizon 64 --iters 8 --out $O/bench_ppo_2048x8.json > $O/ppo8.log 2>&1; tail -1 $O/ppo8.log | cut -c1-200
python - <<PY
import json
p=json.load(open("$O/bench_ppo_2048x8.json")); it=p["iterations"][1:]
print("2048 x 8: e2e %.1f M  rollout %.3f ms  update %.3f ms" % (p["value"]/1e6, sum(h["rollout_s"] for h in it)/len(it)*1e3, sum(h["update_s"] for h in it)/len(it)*1e3))
PY
