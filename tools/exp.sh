HNET_PRECISION=2 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
HNET_PRECISION=2 python bench.py --no-cpu-baseline --no-latency --steps 10 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print(r['value'], r['ms_per_step']); print({k:v for k,v in r['forward']['stage_ms'].items() if k in ('block_3_1','block_4_2','block_4_0+4_1')})"
