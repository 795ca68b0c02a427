for i in 1 2 3 4 5; do timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 2977$i tests/dist_gpu_worker.py /tmp/v.json > /tmp/log.txt 2>&1 || tail -5 /tmp/log.txt; cat /tmp/v.json; echo; done
HNET_PRECISION=2 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench_s3.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/bench_s3.json')); print(r['value'], r['ms_per_step'], r['latency_batch1_ms'])"
