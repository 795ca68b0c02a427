for t in 1; do HNET_S3_TILE=$t python bench.py --no-cpu-baseline --no-latency --steps 10 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print('TILE=$t', r['value'], r['ms_per_step']); print({k:v for k,v in r['forward']['stage_ms'].items() if k.startswith('block') or k.startswith('heads')})"; done
