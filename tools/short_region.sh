#!/bin/bash
# the driver's timed region (bench.py --steps 20 --warmup 5) on 1 .. 4 contexts, three repetitions, a fresh process per point (profiles/r06_ctx_sweep.log)
for rep in 1 2 3; do for nc in 1 2 3 4; do
v=$(python3 bench.py --steps 20 --warmup 5 --contexts $nc --no-extras --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1000,1))")
echo "rep $rep steps 20 warmup 5 contexts $nc: $v k pairs/s"
done; done
