#!/bin/bash
# round 5, GPU call C: k_step phase cycles + wait counters; one-workgroup-per-CU knob of k_mlp_fb on the real loops
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
for v in 0 1; do
  CDA_MLP_FB_ONE_WG=$v timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 10 --out $O/league_onewg$v.json > /dev/null 2>&1
  CDA_MLP_FB_ONE_WG=$v timeout 600 python -m gym_continuousdoubleauction_amd.ppo --iters 8 --out $O/ppo_onewg$v.json > /dev/null 2>&1
done
python - <<'PY'
import json
for n in ("league_onewg0", "league_onewg1", "ppo_onewg0", "ppo_onewg1"):
    try:
        d = json.load(open(f"gpurun_out/r05/{n}.json")); it = d["iterations"]
        print(n, round(d["value"] / 1e6, 1), "M;  rollout ms", [round(h["rollout_s"] * 1e3, 2) for h in it], " update ms", [round(h["update_s"] * 1e3, 2) for h in it])
    except Exception as e:
        print(n, "missing", e)
PY
timeout 900 python tools/phase_timing.py > $O/phase_timing.txt 2>&1; tail -12 $O/phase_timing.txt | cut -c1-1500
export TMPDIR=/tmp CDA_BENCH_PRIMER_MS=0; cd /tmp
BENCH="python $R/bench.py --steps 200 --warmup 16 --repeats 1 --no-cpu-baseline --no-extra-legs --no-policy-leg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/kw/pmc_w1 -o p -- $BENCH > /dev/null 2> $O/kw_1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $O/kw/pmc_w2 -o p -- $BENCH > /dev/null 2> $O/kw_2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $O/kw/pmc_w3 -o p -- $BENCH > /dev/null 2> $O/kw_3.err
cd $R
find $O/kw -name "*counter_collection.csv" | head -3; head -3 $(find $O/kw -name "*counter_collection.csv" | head -1) | cut -c1-600
python tools/kstep_wait_attribution.py $O/kw > $O/kstep_wait_counters.txt 2>&1; cat $O/kstep_wait_counters.txt
rm -rf $O/kw
