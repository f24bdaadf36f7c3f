#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes for the step kernel.
# Usage: [BENCH_STEPS=200 BENCH_EXTRA="--groups 2"] tools/profile_gpu.sh <tag>   -> writes gpurun_out/prof_<tag>/... incl. the PMC pass
# <markets>x<agents>_info<0|1>_g<groups>.json that is committed under profiles/pmc/ (bench.py reads it for that shape only)
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with sys traces).
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# one timed leg, no clock primer: every k_step dispatch of the run belongs to the measured env (two concurrent group chains)
export CDA_BENCH_PRIMER_MS=0
BENCH="python $R/bench.py --steps ${BENCH_STEPS:-200} --warmup 16 --repeats 1 --no-cpu-baseline --no-extra-legs ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $BENCH > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o p -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_FLAT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o p -- $BENCH > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq3 -o p -- $BENCH > /dev/null 2> $OUT/pmc_sq3.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o p -- python $R/tools/pmc_calib.py > /dev/null 2> $OUT/calib_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o p -- python $R/tools/pmc_calib.py > /dev/null 2> $OUT/calib_write.err
cd $R
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
