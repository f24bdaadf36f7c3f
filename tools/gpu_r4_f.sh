#!/bin/bash
# A/B: shape-specialised k_step<INFO, A, 4> vs the generic kernel, interleaved, driver's command and long legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4f; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_hip_golden.py tests/test_hip_vs_oracle_batch.py tests/test_hip_groups.py tests/test_hip_baseline_configs.py -q -m gpu > $O/parity.log 2>&1; tail -2 $O/parity.log
: > $O/ab.txt
for rep in 1 2 3 4 5; do
  for g in 0 1; do
    for extra in "" "--no-info"; do
      v=$(CDA_STEP_GENERIC=$g timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs $extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%.1f %.2f' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))")
      echo "generic=$g short $extra $v" >> $O/ab.txt
    done
  done
done
for rep in 1 2 3; do
  for g in 0 1; do
    for cfg in c3 c4; do
      v=$(CDA_STEP_GENERIC=$g timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 64 --config $cfg --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%.1f %.2f' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))")
      echo "generic=$g long $cfg $v" >> $O/ab.txt
    done
  done
done
python - <<'PY'
import collections, statistics
d=collections.defaultdict(list)
for l in open("gpurun_out/r4f/ab.txt"):
    p=l.split(); key=" ".join(p[:-2]); d[key].append(float(p[-2]))
for k in sorted(d): print(f"{k:40s} median {statistics.median(d[k]):8.1f} M  all {d[k]}")
PY
