#!/bin/bash
# Run ON THE GPU BOX: the driver's command (20 timed steps, five legs, the median) for a few chain counts, twice each.
# Usage: tools/short_run_probe.sh [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$PWD}
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['timed_repeats']; print('%.1f M with info (legs %.1f .. %.1f us, median %.2f)  kernel %.2f us  without info %.1f M' % (d['value']/1e6, t['min']*1e3, t['max']*1e3, t['median']*1e3, d['roofline']['kernel_ms']*1e3, (d.get('value_without_info') or 0)/1e6))"; }
for g in 1 2 3 4; do
  for rep in 1 2; do
    echo -n "groups $g: "; python $R/bench.py --gpus 1 --steps 20 --warmup 5 --groups $g --no-cpu-baseline "$@" 2>/dev/null | tail -1 | p
  done
done
