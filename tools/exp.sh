HNET_PRECISION=2 python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print(r['value'], r['ms_per_step'], r['latency_batch1_ms']['p50']); print({k:v for k,v in r['forward']['stage_ms'].items() if k.startswith('block') or k.startswith('heads')})"
