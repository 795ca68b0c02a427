for sk in none; do HNET_STREAM_SKIP=$sk python bench.py --mode stream --steps 30 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('$sk', r['value'], r['ms_per_step'])"; done
python bench.py --mode stream --steps 30 --warmup 3 --variant prior3 --batch 64 2>&1 | tail -1 | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('prior3 b64', r['value'], r['ms_per_step'])"
