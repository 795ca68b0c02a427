for b in 1 5; do
HNET_FUSED_REDUCE=0 python tools/attic/pc_check.py gpurun_out/fr0.npz $b > /dev/null
HNET_FUSED_REDUCE=1 python tools/attic/pc_check.py gpurun_out/fr1.npz $b > /dev/null
python - <<PY
import numpy as np
a=np.load('gpurun_out/fr0.npz'); b=np.load('gpurun_out/fr1.npz')
print("batch $b: bitwise mean", np.array_equal(a['mean'], b['mean']), "cov", np.array_equal(a['cov'], b['cov']), float(np.abs(a['mean']-b['mean']).max()))
PY
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for f in 0 1; do HNET_FUSED_REDUCE=$f python bench.py --batch 1 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('fused_reduce=$f', r['latency_batch1_ms']['p50'], r['latency_batch1_ms']['end_to_end_p50'])"; done
python bench.py --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
