for d in 0 1 0 1; do HNET_XCD_REMAP=$d python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print('xcd=$d', r['value'], r['ms_per_step'], {k:s[k] for k in ('block_1_2','block_2_2','block_2_3','block_3_2','block_3_3','block_3_4','block_3_5','heads_fc1')})"; done
HNET_XCD_REMAP=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
