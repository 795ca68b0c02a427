bash tools/profile_round.sh r01_v6 > /dev/null 2>&1
tools/pmc_pass.sh v6_mfma SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > gpurun_out/pmc_v6_mfma.txt 2>&1
python bench.py > gpurun_out/bench_v6.json 2>gpurun_out/bench_v6.err
python -c "
import json; r=json.load(open('gpurun_out/bench_v6.json')); print(r['value'], r['ms_per_step'], r['roofline'], r['latency_batch1_ms']['p50'], r['cpu_baseline']['value'])"
