#!/bin/bash
# Run ON THE GPU BOX (through gpurun): PMC passes over the network kernels (tools/mlp_bench.py: the update's minibatch shape, the rollout's policy step).
# Each counter group in its own run with --kernel-trace only.  -> gpurun_out/prof_mlp/summary.txt + mlp_pmc.json
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_mlp
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
B="python $R/tools/mlp_bench.py --iters 4"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- $B > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/sq2 -o p -- $B > /dev/null 2> $OUT/sq2.err
cd $R
python tools/summarize_mlp_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
