for cfg in "0 3" "8 1" "8 3" "0 3"; do set -- $cfg; HNET_B4_DBG=$1 HNET_PATCH_REV=$2 python bench.py --no-cpu-baseline --no-latency --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); s=r['forward']['stage_ms']; print('b4dbg=$1 patchrev=$2', r['value'], s['block_4_0+4_1'], s['block_3_1'], s['block_4_2'], s['block_4_3'])"; done
