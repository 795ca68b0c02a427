#!/bin/bash
# pairs/s of the mid-size steps (BASELINE config 3: 64 pairs prior-3 N=16; config 5's per-GPU shape: 32 pairs) and the headline step on 1 .. 4 contexts of an hnet_group,
# every point in a fresh process:   tools/ctx_sweep.sh <tag>     (run on three fresh boxes: the spread between boxes is the reliability asked for)
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
for shape in "prior3 64 16" "prior3 32 16" "full 256 32"; do
  set -- $shape
  line="$TAG $1 batch=$2 N=$3:"
  for nc in 1 2 3 4; do
    v=$(python3 bench.py --variant $1 --batch $2 --mc $3 --contexts $nc --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1000,1), 'ok' if d['verify']['passed'] else 'FAIL')")
    line="$line  ${nc}ctx $v"
  done
  echo "$line"
done
