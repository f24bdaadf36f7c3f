#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags]: an experimental build of the SAME HIP library (csrc/cda_hip.hip recompiled with the extra flags, the network
# objects reused) -> gpurun_ab/libcda_hip_NAME.so; load it with CDA_HIP_LIB=<path> (gym_continuousdoubleauction_amd/_lib.py).  For same-box A/B measurements.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/gym_continuousdoubleauction_amd/csrc
OBJ=$ROOT/gym_continuousdoubleauction_amd/build_tmp
OUT=$ROOT/gpurun_ab
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -Os -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp"
mkdir -p $OUT $OBJ
cd $CSRC
for f in cda_ppo cda_mlp; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ]; then hipcc $FLAGS -c $f.hip -o $OBJ/$f.o & fi
done
hipcc $FLAGS "$@" -c cda_hip.hip -o $OBJ/cda_hip_$NAME.o
wait
hipcc $FLAGS -shared -o $OUT/libcda_hip_$NAME.so $OBJ/cda_hip_$NAME.o $OBJ/cda_ppo.o $OBJ/cda_mlp.o $OBJ/cda_mlp_h1.o $OBJ/cda_mlp_h2.o $OBJ/cda_mlp_h3.o $OBJ/cda_mlp_h6.o $OBJ/cda_mlp_h7.o $OBJ/cda_mlp_h8.o
echo built $OUT/libcda_hip_$NAME.so
