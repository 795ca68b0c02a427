python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); l=r['latency_batch1_ms']; print(r['value'], l['p50'], l['p95'], l['end_to_end_p50'], l['end_to_end_p95'])"
