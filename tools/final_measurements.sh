#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every measurement kept under profiles/<tag>/ for a round.
# Usage: tools/final_measurements.sh <tag>     -> gpurun_out/final_<tag>/...
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
(time python -m pytest tests -m gpu -q) > $O/gputest.log 2>&1
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2>/dev/null
python bench.py --groups 2 --no-cpu-baseline --no-extra-legs > $O/bench_groups2.json 2>/dev/null
python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null
python bench.py --fused 64 --steps 1024 --warmup 64 --no-cpu-baseline > $O/bench_fused.json 2>/dev/null
python bench.py --force-gather --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_forcegather_one_rank.json
python tools/handback_host_probe.py 2>&1 | grep -v amdgpu.ids > $O/handback_host_probe.txt
python tools/bigbook_speed.py > $O/bigbook_speed.json 2>/dev/null
python tools/book_census.py > $O/book_census.json 2>/dev/null
(CDA_FUZZ_CASES=250 CDA_FUZZ_SEED=20260927 python -m pytest tests/test_hip_vs_oracle_batch.py -q -m gpu -k "random_configurations") > $O/fuzz_soak.txt 2>&1
echo "markets agents  M agent-steps/s with info / without info   us per step (with info)" > $O/batch_scaling.txt
for n in 1024 2048 8192 16384 65536; do python bench.py --markets $n --no-cpu-baseline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['config']['agents'], round(d['value']/1e6,1), round(d['value_without_info']/1e6,1), round(d['ms_per_step']*1000,1))"; done >> $O/batch_scaling.txt
for a in 8 16; do python bench.py --agents $a --no-cpu-baseline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(4096, $a, round(d['value']/1e6,1), round(d['value_without_info']/1e6,1), round(d['ms_per_step']*1000,1))"; done >> $O/batch_scaling.txt
python -m gym_continuousdoubleauction_amd.ppo --iters 8 --out $O/bench_ppo.json > $O/ppo.log 2>&1
python -m gym_continuousdoubleauction_amd.ppo --iters 6 --per-sample-forward --out $O/bench_ppo_per_sample_forward.json > $O/ppo_per_sample.log 2>&1

tools/profile_gpu.sh ${TAG}_g4 > $O/profile_g4.log 2>&1
BENCH_EXTRA="--groups 2" tools/profile_gpu.sh ${TAG}_g2 > $O/profile_g2.log 2>&1
mkdir -p $O/pmc; cp $R/gpurun_out/prof_${TAG}_g4/*x*_info*_g*.json $R/gpurun_out/prof_${TAG}_g2/*x*_info*_g*.json $O/pmc/ 2>/dev/null
cp $R/gpurun_out/prof_${TAG}_g4/summary.txt $O/rocprof_summary.txt; cp $R/gpurun_out/prof_${TAG}_g2/summary.txt $O/rocprof_summary_two_chains.txt
cp $R/gpurun_out/prof_${TAG}_g4/trace_bench.json $O/bench_under_rocprof.json 2>/dev/null
cp $R/gpurun_out/prof_${TAG}_g4/trace/t_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
export PYTHONPATH=$R TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ppo_trace -o t -- python -m gym_continuousdoubleauction_amd.ppo --iters 6 > $O/ppo_under_rocprof.log 2>&1
cp $O/ppo_trace/t_kernel_stats.csv $O/kernel_stats_ppo.csv 2>/dev/null; rm -rf $O/ppo_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused_trace -o t -- python $R/bench.py --fused 64 --steps 1024 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
cp $O/fused_trace/t_kernel_stats.csv $O/kernel_stats_fused.csv 2>/dev/null; rm -rf $O/fused_trace
cd $R
tail -3 $O/gputest.log; cat $O/bench_final.json | cut -c1-400; cat $O/batch_scaling.txt
