#!/bin/bash
# Run ON THE GPU BOX: build k_step variants and bench each (A/B within one box).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/variants
SRC=$R/gym_continuousdoubleauction_amd/csrc/cda_hip.hip
run() {
  name=$1; shift
  hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared "$@" -Rpass-analysis=kernel-resource-usage -o $R/gpurun_out/variants/$name.so $SRC 2>&1 | grep -A8 "k_step" | grep "VGPRs:\|Occupancy\|ScratchSize\|VGPRs Spill" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
  CDA_HIP_LIB=$R/gpurun_out/variants/$name.so python $R/bench.py --steps 600 --warmup 64 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']/1e6,1), 'M agent-steps/s', round(d['roofline']['kernel_ms']*1000,1), 'us')"
}
run os -Os
run cap128_w5 -Os -DCDA_BOOK_CAP=128 -DCDA_MIN_WAVES=5
run cap128_w6 -Os -DCDA_BOOK_CAP=128 -DCDA_MIN_WAVES=6
run cap128_w5_o3 -O3 -DCDA_BOOK_CAP=128 -DCDA_MIN_WAVES=5
