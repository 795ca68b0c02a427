#!/bin/bash
# cost of the per-step RCCL gather on a 1-rank communicator (gpurun box), round 5:
#  (1) the Python harness: the same step with and without --force-collective, at 256 and at 32 pairs per GPU
#  (2) the C++ caller (tests/cpp/rccl_gather_example.cpp --time): forward only / gather on the compute stream / gather on a side stream, at 32 and 256 pairs,
#      and BASELINE config 4's shape (one pair, N = 32: partial forward + ncclAllGather + finish on the gathered buffer)
LOG=$GRAFT_REPO_ROOT/gpurun_out/r05_rccl_gather_cost.log
B="python bench.py --no-cpu-baseline --no-latency --no-extras --no-verify --steps 200 --warmup 20"
for cfg in "" "--pairs-total 32 --variant prior3 --mc 16"; do
  for fc in "" "--force-collective"; do
    for rep in 1 2; do
      $B $cfg $fc 2>/dev/null | python -c "
import sys, json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cost_probe cfg=[$cfg] collective=[$fc]', r['value'], r['ms_per_step'])" | tee -a $LOG
    done
  done
done
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuahn_vio_amd import weights
weights.save_blob("gpurun_out/probe.hnw", weights.synthetic_state(0))
PY
for nb in 32 256; do ./tests/cpp/rccl_gather_example.bin gpurun_out/probe.hnw $nb --time 300 2>&1 | grep "RCCL_" | tee -a $LOG; done
rm -f gpurun_out/probe.hnw
