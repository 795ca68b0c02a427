python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for b in 1 4 16 64; do python bench.py --batch $b --no-cpu-baseline --no-latency --steps 30 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('batch $b', r['value'], r['ms_per_step'])"; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); l=r['latency_batch1_ms']; print(r['value'], l['p50'], l['end_to_end_p50'])"
