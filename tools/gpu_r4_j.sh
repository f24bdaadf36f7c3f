#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4j; mkdir -p $O
export PYTHONPATH=$R
for lib in $R/tools/variants/*.so; do
  for wv in 8 4; do
    CDA_HIP_LIB=$lib CDA_MLP_WAVES=$wv timeout 300 python tools/mlp_bench.py --iters 10 --json $O/b.json > $O/b.log 2>&1
    python -c "
import json; d=json.load(open('$O/b.json')); print('$(basename $lib) waves=$wv fwd %.1f bwd %.1f step %.1f' % (d['forward_train_us'], d['backward_us'], d['minibatch_step_us']))"
  done
done
