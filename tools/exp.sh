for d in 0 1 2 3; do HNET_DBG=$d HNET_PRECISION=2 python bench.py --no-cpu-baseline --no-latency --steps 5 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print('DBG=$d', {k:v for k,v in r['forward']['stage_ms'].items() if k in ('block_4_4','block_4_3','block_4_5','block_4_6','block_2_2','heads_fc1')})"; done
