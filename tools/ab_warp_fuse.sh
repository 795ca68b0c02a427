#!/bin/bash
# A/B of the in-kernel warp: contexts 1 and 4, three repetitions each, interleaved
mkdir -p gpurun_out
for rep in 1 2 3; do
 for wf in 1 0; do
  for nc in 1 4; do
   HNET_WARP_FUSE=$wf python bench.py --honour-env --no-extras --no-cpu-baseline --no-latency --no-verify --steps 200 --warmup 40 --contexts $nc 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
rf=r.get('roofline') or {}
print('wf=$wf nc=$nc rep=$rep value=%.0f ms_per_step=%.4f' % (r['value'], r['ms_per_step']), 'roofline kernel', rf.get('kernel'), rf.get('kernel_ms'))
"
  done
 done
done
