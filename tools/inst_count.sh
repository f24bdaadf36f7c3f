#!/bin/bash
# Run ON THE GPU BOX: instruction counts per market-step by phase (skip masks, every agent passes) and by action category.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp CDA_HIP_LIB=$R/build_tmp/dbgskip.so     # hipcc <flags of __graft_entry__> -DCDA_DEBUG_SKIP
cd /tmp
run() {  # mask cat tag
  rm -rf /tmp/ic; mkdir -p /tmp/ic
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/ic -o p -- python $R/tools/inst_count_probe.py $1 $2 > /dev/null 2>&1
  python $R/tools/inst_count_summary.py /tmp/ic "$3"
}
run 0 0 "all pass: full step"
run 3 0 "all pass: - step_market - mark-to-market"
run 1 0 "all pass: - step_market (mtm kept)"
run 4 0 "all pass: - observation"
run 8 0 "all pass: - reward"
run 15 0 "all pass: load + outputs + store only"
run 31 0 "all pass: load only"
run 0 9 "uniform random actions"
run 0 2 "all bid limit"
run 0 3 "all bid modify"
run 0 4 "all bid cancel"
run 0 1 "all bid market"
run 256 9 "uniform random: - approval"
run 512 9 "uniform random: - cash / hold transfers"
run 1024 9 "uniform random: - own-order lookup (every limit order new)"
run 4096 9 "uniform random: - fill settlement"
run 4864 9 "uniform random: - approval - transfers - settlement"
