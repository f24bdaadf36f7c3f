#!/bin/bash
# Run ON THE GPU BOX: what a per-step wait for an action server costs k_run_random's batch (every market-wave sleeps 4 / 8 / 12 us after each step; tools/build_variant.sh
# stallN -DCDA_RR_STALL_TICKS=N, ticks of 10 ns) - the shipped library and the three variants, alternating, same box.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r06
for i in 1 2; do
  for v in 0 400 800 1200; do
    L=$R/gpurun_ab/libcda_hip_stall$v.so; [ $v = 0 ] && L=$R/gym_continuousdoubleauction_amd/libcda_hip.so
    CDA_HIP_LIB=$L python bench.py --no-cpu-baseline --no-league-leg --no-policy-leg > gpurun_out/r06/stall${v}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json
for v in (0, 400, 800, 1200):
    ds = [json.load(open(f"gpurun_out/r06/stall{v}_{i}.json")) for i in (1, 2)]
    print(f"stall {v / 100:4.1f} us per step:", "run_random_one_launch", [round(d["value_run_random_one_launch"] / 1e6, 1) for d in ds], "M agent-steps/s;",
          "us per step of the batch", [round(4096 * 4 / d["value_run_random_one_launch"] * 1e6, 2) for d in ds])
PY
