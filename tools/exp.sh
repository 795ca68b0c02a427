python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for t in 0 1; do HNET_B30_S3=$t python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print($t, r['value'], r['ms_per_step'], s['block_3_0'], s['block_3_1'])"; done
