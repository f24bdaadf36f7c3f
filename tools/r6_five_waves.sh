#!/bin/bash
# Run ON THE GPU BOX: what a 96-VGPR budget (five waves per SIMD) costs the step kernels - the same bench, this tree's library against gpurun_ab/libcda_hip_w5.so
# (tools/build_variant.sh w5 -DCDA_MIN_WAVES=5), alternating.  The w5 kernels still RUN at four waves per SIMD (their cold callees keep 128 registers, the kernel
# descriptor takes the maximum): the difference is the pure cost of the spills, not yet offset by anything a fifth wave could do.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-league-leg --no-policy-leg > gpurun_out/w4_$i.json 2>/dev/null
  CDA_HIP_LIB=$R/gpurun_ab/libcda_hip_w5.so python bench.py --no-cpu-baseline --no-league-leg --no-policy-leg > gpurun_out/w5_$i.json 2>/dev/null
done
python - <<'PY'
import json
for b in ("w4", "w5"):
    ds = [json.load(open(f"gpurun_out/{b}_{i}.json")) for i in (1, 2)]
    print(b, {k: [round(d[k] / 1e6, 1) for d in ds] for k in ("value", "value_without_info", "value_one_launch", "value_run_random_one_launch")})
PY
