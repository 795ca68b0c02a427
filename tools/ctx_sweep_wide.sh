#!/bin/bash
# tools/ctx_sweep.sh beyond four contexts (4 / 5 / 6 / 8 members of an hnet_group): the runtime serves a process with four hardware queues (profiles/r06_ctx_sweep.log)
cd $GRAFT_REPO_ROOT
for shape in "prior3 64 16" "prior3 32 16" "full 256 32"; do
  set -- $shape
  line="$1 batch=$2 N=$3:"
  for nc in 4 5 6 8; do
    v=$(python3 bench.py --variant $1 --batch $2 --mc $3 --contexts $nc --steps 120 --warmup 24 --no-extras --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1000,1), 'ok' if d['verify']['passed'] else 'FAIL')")
    line="$line  ${nc}ctx $v"
  done
  echo "$line"
done
