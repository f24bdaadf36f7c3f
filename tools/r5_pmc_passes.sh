#!/bin/bash
# round 5: PMC passes of the headline shape on THIS build (profiles/pmc/*.json: what bench.py's roofline.traffic reads), two and four chains
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
BENCH_STEPS=200 BENCH_EXTRA="--groups 2 --no-policy-leg" bash tools/profile_gpu.sh r05_g2 > gpurun_out/prof_r05_g2_summary.txt 2>&1
BENCH_STEPS=200 BENCH_EXTRA="--groups 4 --no-policy-leg" bash tools/profile_gpu.sh r05_g4 > gpurun_out/prof_r05_g4_summary.txt 2>&1
tail -4 gpurun_out/prof_r05_g2_summary.txt | cut -c1-900; tail -2 gpurun_out/prof_r05_g4_summary.txt | cut -c1-900
ls gpurun_out/prof_r05_g2/*.json gpurun_out/prof_r05_g4/*.json
for d in gpurun_out/prof_r05_g2 gpurun_out/prof_r05_g4; do rm -rf $d/trace $d/pmc_* $d/calib_*; done
