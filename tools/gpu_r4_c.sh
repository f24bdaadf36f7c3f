#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_hip_mlp.py -q -m gpu > $O/test_mlp.log 2>&1; tail -3 $O/test_mlp.log
for mt in 4 2; do
  CDA_MLP_MT=$mt timeout 300 python tools/mlp_bench.py --json $O/bench_mt$mt.json > $O/bench_mt$mt.log 2>&1; grep -v "^ *\"rows\|agents" $O/bench_mt$mt.log | tr '\n' ' '; echo
  timeout 120 python tools/mlp_timing.py --mt $mt > $O/timing_mt$mt.txt 2>&1; cat $O/timing_mt$mt.txt
done
timeout 120 python tools/mlp_timing.py --mt 1 --sample --rows 1024 --block 3 > $O/timing_sample.txt 2>&1; cat $O/timing_sample.txt
cd /tmp
PM="python $R/tools/mlp_bench.py --iters 5"
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -o p -- $PM > /dev/null 2> $O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -o p -- $PM > /dev/null 2> $O/pmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -o p -- $PM > /dev/null 2> $O/pmc3.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -o p -- $PM > /dev/null 2> $O/pmc4.err
cd $R
python - <<'PY'
import csv, glob, os
from collections import defaultdict
O = "gpurun_out/r4c"
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not any(x in k for x in ("mlp", "ppo_loss", "grad_reduce", "prep_rows", "k_adam")):
                continue
            short = k.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
            acc[(short, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(short, row["Counter_Name"])] += 1
        for k in sorted(acc):
            print(f"{sub} {k[0]:42s} {k[1]:34s} {acc[k]/cnt[k]:16.1f} (n={cnt[k]})")
PY
tail -3 $O/pmc2.err
for mt in 4 2; do
  CDA_MLP_MT=$mt timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/ppo_fused_mt$mt.json > $O/ppo_fused_mt$mt.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/ppo_fused_mt$mt.json")); h=d["iterations"][2:]
print("MT=$mt e2e %.1f M agent-steps/s; rollout %.2f ms update %.2f ms" % (d["value"]/1e6, 1e3*sum(x["rollout_s"] for x in h)/len(h), 1e3*sum(x["update_s"] for x in h)/len(h)))
PY
done
