#!/bin/bash
# Timing builds of mlp_head1.hip with one phase left out each (tools/_ab/libpase_mh_<name>.so): where does a tile's time go?
#   bash tools/mlp_head1_ablate.sh build        (CPU container: hipcc cross-compiles)
#   bash tools/mlp_head1_ablate.sh run          (GPU box)
cd "$(dirname "$0")/.." || exit 1
VARS="base: no1:-DMH_ABL_NO1 no2:-DMH_ABL_NO2 no3:-DMH_ABL_NO3 nocopy:-DMH_ABL_NOCOPY nohead:-DMH_ABL_NOHEAD nostore:-DMH_ABL_NOSTORE nopf:-DMH_ABL_NOPF nomfma:-DMH_ABL_NO1,-DMH_ABL_NO2,-DMH_ABL_NO3 trace:-DMH_TRACE"
if [ "$1" = build ]; then
  mkdir -p tools/_ab/obj_mh
  for f in pase_amd/csrc/*.hip; do
    [ "$(basename $f)" = mlp_head1.hip ] && continue
    /opt/rocm/bin/hipcc -c $f -o tools/_ab/obj_mh/$(basename $f .hip).o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include -I pase_amd/csrc -Wno-unused-result 2>/dev/null &
  done
  wait
  for v in $VARS; do
    n=${v%%:*}; fl=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc -c pase_amd/csrc/mlp_head1.hip -o tools/_ab/obj_mh/mlp_head1_$n.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include -I pase_amd/csrc -Wno-unused-result $fl 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libpase_mh_$n.so $(ls tools/_ab/obj_mh/*.o | grep -v mlp_head1_) tools/_ab/obj_mh/mlp_head1_$n.o && echo built $n
  done
else
  for v in $VARS; do
    n=${v%%:*}
    if [ $n = trace ]; then PASE_LIB=tools/_ab/libpase_mh_$n.so python tools/mlp_head1_bench.py trace 2>/dev/null | tail -2; continue; fi
    echo -n "$n: "; PASE_LIB=tools/_ab/libpase_mh_$n.so python tools/mlp_head1_bench.py fused 2>/dev/null | tail -1
  done
fi
