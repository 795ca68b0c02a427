for d in 0 1 2 4 6 7; do HNET_B4_DBG=$d python bench.py --no-cpu-baseline --no-latency --steps 10 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print('dbg=$d', s['block_4_0+4_1'])"; done
