for d in 0 12 25 40 0; do HNET_S3_STAGGER=$d python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print('stagger=$d', r['value'], r['ms_per_step'], {k:s[k] for k in ('block_2_2','block_2_3','block_3_3','block_2_4','heads_fc1','block_1_2')})"; done
