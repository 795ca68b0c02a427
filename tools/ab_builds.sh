#!/bin/bash
# same-box A/B of two BUILDS of the library (cuahn_vio_amd/libhnet_hip_base.so = the source state to compare against, built from a git worktree; the tree's
# libhnet_hip.so = the candidate): processes alternate, tools/ab_bench.py prints median / min stage times of each.   bash tools/ab_builds.sh [batch] [rounds] [stages]
B=${1:-256}; R=${2:-3}; ST=${3:-}
for i in $(seq 1 $R); do
  for lib in base cand; do
    if [ $lib = base ]; then export HNET_LIB_PATH=$PWD/cuahn_vio_amd/libhnet_hip_base.so; else unset HNET_LIB_PATH; fi
    echo "== $lib round $i: $(python tools/ab_bench.py '{}' --batch $B --rounds 5 ${ST:+--stages $ST} 2>/dev/null | tail -1 | cut -c1-900)"
  done
done
