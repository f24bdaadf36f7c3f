#!/bin/bash
# round 5: where k_step's 42 % SQ_WAIT_ANY comes from - the CDA_PHASE_TIMING build's cycle stamps + three PMC passes of the bench's free-running leg
# (--pmc in its own runs, --kernel-trace only: no other trace domain), folded into one table by tools/kstep_wait_attribution.py -> profiles/r05/k_step_wait_attribution.txt.
# Usage (through gpurun): bash tools/r5_kstep_wait_attribution.sh
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/phase_timing.py > $O/phase_timing.txt 2>&1; tail -12 $O/phase_timing.txt | cut -c1-1500
export TMPDIR=/tmp CDA_BENCH_PRIMER_MS=0; cd /tmp
BENCH="python $R/bench.py --steps 200 --warmup 16 --repeats 1 --no-cpu-baseline --no-extra-legs --no-policy-leg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/kw/pmc_w1 -o p -- $BENCH > /dev/null 2> $O/kw_1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $O/kw/pmc_w2 -o p -- $BENCH > /dev/null 2> $O/kw_2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $O/kw/pmc_w3 -o p -- $BENCH > /dev/null 2> $O/kw_3.err
cd $R
python tools/kstep_wait_attribution.py $O/kw > $O/kstep_wait_counters.txt 2>&1; cat $O/kstep_wait_counters.txt
rm -rf $O/kw
