python bench.py --mode mc --steps 5 --batch 64 2>&1 | tail -2
python bench.py --variant prior3 --batch 64 --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('prior3 b64:', r['value'], r['ms_per_step'], r['latency_batch1_ms'])"
