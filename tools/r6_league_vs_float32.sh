#!/bin/bash
# round 6: the fused league loop's two trained policies beside the legacy float32 torch league loop, PPO_DEFAULTS and RLLIB_DEFAULTS
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r06; mkdir -p $O
timeout 900 python tools/league_curve.py --markets 512 --iters 40 --against-float32 > $O/league_vs_float32.txt 2> $O/league_vs_float32.err
timeout 900 python tools/league_curve.py --markets 512 --iters 40 --against-float32 --rllib > $O/league_vs_float32_rllib.txt 2>> $O/league_vs_float32.err
tail -6 $O/league_vs_float32.txt; tail -6 $O/league_vs_float32_rllib.txt; tail -3 $O/league_vs_float32.err
