python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print(r['value'], r['ms_per_step'], s['block_4_0+4_1'], s['block_3_1'], s['block_4_2'])"
