(python bench.py --no-cpu-baseline --no-latency --steps 12000 > /tmp/b.json 2>/dev/null &) 
sleep 18
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -6; sleep 1; done
wait
sleep 8
cat /tmp/b.json | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
