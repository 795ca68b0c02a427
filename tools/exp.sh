HNET_PRECISION=2 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
HNET_PRECISION=0 HNET_PREP_TILED=2 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
HNET_PRECISION=2 HNET_PREP_TILED=0 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print(r['value'], r['ms_per_step'], r['latency_batch1_ms']); print({k:v for k,v in r['forward']['stage_ms'].items() if 'prep' in k})"
