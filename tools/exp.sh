python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for d in 0 1 0 1; do HNET_S3_MF16=$d python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print('mf16=$d', r['value'], r['ms_per_step'], {k:s[k] for k in ('block_3_2','block_4_3','block_2_2')})"; done
