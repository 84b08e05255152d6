#!/bin/bash
# Build a tuning variant of librecalgo_hip.so: one translation unit recompiled with extra -D flags, linked with the objects of
# the regular build into recalgorithm_amd/build/variants/librecalgo_hip_<name>.so (travels to the GPU box; selected with
# RECALGO_HIP_LIB).   usage: scripts/build_variant.sh <name> <unit, e.g. sparse> <flags...>
set -e
NAME=$1; UNIT=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
B=$R/recalgorithm_amd/build
mkdir -p $B/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fvisibility=hidden -I$R/include -I$R/recalgorithm_amd/csrc "$@" -c $R/recalgorithm_amd/csrc/$UNIT.hip -o $B/variants/${UNIT}_$NAME.o
OBJS=$(ls $B/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $B/variants/${UNIT}_$NAME.o -o $B/variants/librecalgo_hip_$NAME.so
echo $B/variants/librecalgo_hip_$NAME.so
