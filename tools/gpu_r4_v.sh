#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export PYTHONPATH=$R TMPDIR=/tmp; O=$R/gpurun_out/r4v; mkdir -p $O
for i in 1 2 3; do
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/ppo.json > $O/ppo.log 2>&1
python - <<PY
import json
p=json.load(open("$O/ppo.json")); it=p["iterations"][1:]
print("e2e %.1f M  rollout %.3f ms  update %.3f ms" % (p["value"]/1e6, sum(h["rollout_s"] for h in it)/len(it)*1e3, sum(h["update_s"] for h in it)/len(it)*1e3))
PY
done
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/tools/mlp_bench.py --iters 4 > /dev/null 2> $O/fetch.err
cd $R
python - <<PY
import csv,glob
acc={};cnt={}
for f in glob.glob("$O/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_mlp_fb(","k_mlp_wgrad"):
            if k in r["Kernel_Name"]:
                acc[k]=acc.get(k,0)+float(r["Counter_Value"]); cnt[k]=cnt.get(k,0)+1
for k in acc: print(k, "HBM read MB per launch: %.1f" % (acc[k]/cnt[k]*2048/1e6))
PY
