#!/bin/bash
# Run ON THE GPU BOX: bench prebuilt library variants (build_tmp/variants/*.so) on the headline workload.
# Usage: tools/bench_variants.sh [extra bench.py args]    -> one line per variant and run
R=${GRAFT_REPO_ROOT:-$PWD}
for so in $R/build_tmp/variants/*.so; do
  for rep in 1 2; do
    CDA_HIP_LIB=$so python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $so)', round(d['value']/1e6,1), 'M', round(d.get('value_with_info',0)/1e6,1), 'M with info', round(d['roofline']['kernel_ms']*1000,2), 'us', 'flagged', d['config']['flagged_markets'])"
  done
done
