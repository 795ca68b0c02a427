for d in 0 1 0 1; do HNET_SIDE_STREAM=$d python bench.py --no-cpu-baseline --no-latency --steps 30 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('side=$d', r['value'], r['ms_per_step'])"; done
