for t in 0 4 1 0 4; do HNET_S3_TILE=$t python bench.py --no-cpu-baseline --no-latency --steps 20 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); s=r['forward']['stage_ms']; print('tile=$t', r['value'], s['block_1_2'], s['block_2_2'], s['block_2_3'])"; done
