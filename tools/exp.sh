bash tools/profile_round.sh r01_v5 > /dev/null 2>&1
tools/pmc_pass.sh v5_mfma SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE > gpurun_out/pmc_v5_mfma.txt 2>&1
python bench.py > gpurun_out/bench_v5.json 2>gpurun_out/bench_v5.err
tail -c 1500 gpurun_out/bench_v5.json
