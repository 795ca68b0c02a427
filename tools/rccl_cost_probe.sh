#!/bin/bash
# cost of the per-step RCCL gather on a 1-rank communicator (gpurun box): the same step with and without --force-collective, at 256 and at 32 pairs
LOG=$GRAFT_REPO_ROOT/gpurun_out/r04_rccl_cost.log
B="python bench.py --no-cpu-baseline --no-latency --no-extras --no-verify --steps 200 --warmup 20"
for cfg in "" "--pairs-total 32 --variant prior3 --mc 16"; do
  for fc in "" "--force-collective"; do
    for rep in 1 2; do
      $B $cfg $fc 2>/dev/null | python -c "
import sys, json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cost_probe cfg=[$cfg] collective=[$fc]', r['value'], r['ms_per_step'])" | tee -a $LOG
    done
  done
done
