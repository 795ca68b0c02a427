HNET_PRECISION=2 python -m pytest tests -m gpu -q 2>&1 | tail -3
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print(r['value'], r['ms_per_step'], r['latency_batch1_ms'])"
HNET_GRAPH=0 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print('no graph', r['latency_batch1_ms'])"
python bench.py --no-cpu-baseline --steps 5 --precision fp32 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print('fp32', r['value'], r['latency_batch1_ms'])"
