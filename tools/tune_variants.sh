#!/bin/bash
# Run ON THE GPU BOX: bench prebuilt k_step variants at several batch sizes.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/variants
SRC=$R/gym_continuousdoubleauction_amd/csrc/cda_hip.hip
for w in 4 5 6; do
  hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -Os -DCDA_MIN_WAVES=$w -o $R/gpurun_out/variants/w$w.so $SRC 2>/dev/null
done
for n in 2048 4096 8192 16384 65536; do
  for w in 4 5 6; do
    CDA_HIP_LIB=$R/gpurun_out/variants/w$w.so python $R/bench.py --steps 300 --warmup 64 --no-cpu-baseline --markets $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n w$w', round(d['value']/1e6,1), 'M agent-steps/s', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
