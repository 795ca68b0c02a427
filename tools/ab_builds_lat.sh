#!/bin/bash
# same-box A/B of two builds at batch 1 (graph replay inside the timing entry point): bash tools/ab_builds_lat.sh [rounds] [variant] [mc]
R=${1:-3}; V=${2:-full}; MC=${3:-32}
for i in $(seq 1 $R); do
  for lib in base cand; do
    if [ $lib = base ]; then export HNET_LIB_PATH=$PWD/cuahn_vio_amd/libhnet_hip_base.so; else unset HNET_LIB_PATH; fi
    echo "== $lib round $i: $(HNET_GRAPH=1 python tools/ab_bench.py '{}' --batch 1 --rounds 7 --variant $V --mc $MC 2>/dev/null | tail -1 | cut -c1-1100)"
  done
done
