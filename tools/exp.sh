bash tools/profile_round.sh r01_v7 > /dev/null 2>&1
tools/pmc_pass.sh v7_mfma SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > gpurun_out/pmc_v7_mfma.txt 2>&1
python bench.py > gpurun_out/bench_v7.json 2>gpurun_out/bench_v7.err
python -c "
import json; r=json.load(open('gpurun_out/bench_v7.json')); print(r['value'], r['ms_per_step'], r['roofline'], r['latency_batch1_ms']['p50'], r['latency_batch1_ms']['end_to_end_p50'], r['cpu_baseline']['value'])"
python bench.py --variant prior3 --batch 64 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('prior3 b64', r['value'], r['ms_per_step'], r['latency_batch1_ms']['p50'])"
python bench.py --mode stream --steps 30 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read()); print('stream', r['value'], r['ms_per_step'])"
