python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/stage_profile.py 1 | head -40
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/bench4.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench4.json')); print(r['value'], r['ms_per_step'], r['latency_batch1_ms'], r['roofline'])"
