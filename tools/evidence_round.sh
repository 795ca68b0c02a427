set -x
bash tools/profile_round.sh r04_v4 > /dev/null 2>&1
HNET_PARITY_TABLE=$PWD/gpurun_out/r04_parity_table.csv python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_mode.py -m gpu -q 2>&1 | tail -2
(python tools/full_batch_check.py 256 32 full 256 3; python tools/full_batch_check.py 256 16 prior3 256 3; python tools/full_batch_check.py 256 32 full 256 2; python tools/full_batch_check.py 256 32 full 256 0) > gpurun_out/r04_full_batch_check.log 2>&1
python bench.py 2>gpurun_out/r04_v4_bench.err | tail -1 > gpurun_out/r04_v4_bench.json
ls -la gpurun_out/r04_* gpurun_out/prof_r04_v4
