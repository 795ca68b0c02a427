HNET_PRECISION=2 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
HNET_PRECISION=0 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py > gpurun_out/bench_full.json 2>gpurun_out/bench_full.err; tail -c 3000 gpurun_out/bench_full.json
