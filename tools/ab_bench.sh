#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds on the headline workload.  Usage: tools/ab_bench.sh lib1.so lib2.so ...   (two rounds each)
# prints agent-steps/s (M): info on / info off, 1000 timed steps, four chains
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  for lib in "$@"; do
    a=$(CDA_HIP_LIB=$R/$lib python $R/bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_ms']*1e3,2))")
    b=$(CDA_HIP_LIB=$R/$lib python $R/bench.py --no-cpu-baseline --no-extra-legs --no-info 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_ms']*1e3,2))")
    echo "$lib  info-on: $a   info-off: $b"
  done
done
