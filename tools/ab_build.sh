#!/bin/bash
# A/B library for same-box kernel comparisons:  bash tools/ab_build.sh NAME -DFLAG ...  ->  tools/_ab/libpase_NAME.so
# (boxes of the pool differ by +-5 %: two variants are only comparable inside ONE gpurun call; tools/step_breakdown.py and
#  bench.py load the variant named by PASE_LIB instead of pase_amd/libpase_hip.so)
# AB_SRC=dir: compile dir/pase_amd/csrc against dir/include instead of the working tree (e.g. `git archive HEAD | tar -x -C dir`)
NAME=$1; shift
cd "$(dirname "$0")/.." || exit 1
SRC=${AB_SRC:-.}
mkdir -p tools/_ab/obj_$NAME
OBJS=""
for f in $SRC/pase_amd/csrc/*.hip; do
  o=tools/_ab/obj_$NAME/$(basename $f .hip).o
  /opt/rocm/bin/hipcc -c $f -o $o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I $SRC/include -I $SRC/pase_amd/csrc -Wno-unused-result "$@" 2>/dev/null &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libpase_$NAME.so $OBJS && ls -la tools/_ab/libpase_$NAME.so
