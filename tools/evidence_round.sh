# the round's evidence set on the GPU box (through gpurun):  bash tools/evidence_round.sh <tag>      e.g. r05_v1
TAG=${1:-r05_v1}
R=${TAG%%_*}
set -x
bash tools/profile_round.sh $TAG > /dev/null 2>&1
rm -f gpurun_out/${R}_parity_table.csv gpurun_out/${R}_full_batch_check.log
HNET_PARITY_TABLE=$PWD/gpurun_out/${R}_parity_table.csv python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_mode.py -m gpu -q 2>&1 | tail -2
(python tools/full_batch_check.py 256 32 full 256 3; python tools/full_batch_check.py 256 16 prior3 256 3; python tools/full_batch_check.py 256 32 full 256 2; python tools/full_batch_check.py 256 32 full 256 0) > gpurun_out/${R}_full_batch_check.log 2>&1
python tools/determinism_stress.py 2000 > gpurun_out/${R}_determinism.log 2>&1
# round 6: the one-XCD tail chains against the per-layer launches (outputs, per-layer maps, batch-1 / batch-8 device time), the chain kernel's in-kernel phase stamps
python tools/chain_check.py 200 > gpurun_out/${R}_chain_check.log 2>&1
(tools/trace_chain_plain.bin 1; tools/trace_chain_plain.bin 8; tools/trace_chain.bin 1) > gpurun_out/${R}_chain_trace.log 2>&1
python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
# round 6: block 4 sampling its own input (opt-in) against the two launches: A/B of the whole step, the kernel's phases (hipcc ... -DB4_WARPIN=0|1 -DHNET_B4_TRACE tools/trace_b4.hip)
bash tools/ab_warp_fuse.sh > gpurun_out/${R}_warp_fuse_ab.log 2>&1
(tools/trace_b4_w0.bin; tools/trace_b4_w1.bin) > gpurun_out/${R}_b4_trace.log 2>&1
# round 6, latency path: the FC partial sums of the chains and the copy-free hnet_infer graph (A/B), the chain soak, the batch-1 kernel trace
python tools/chain_fc_ab.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_chain_fc_ab.log
python tools/e2e_latency_ab.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_e2e_latency_ab.log
python tools/soak_chain.py 400000 2>&1 | grep "^soak" > gpurun_out/${R}_chain_soak_raw.log
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace -d gpurun_out/lat_tr --output-format csv -- python3 tools/lat_trace.py full 32 100 > /dev/null 2>&1; python3 tools/lat_trace_summary.py gpurun_out/lat_tr 16 > gpurun_out/${R}_lat_trace_summary.log 2>&1; rm -rf gpurun_out/lat_tr)
ls -la gpurun_out/${R}_* gpurun_out/prof_$TAG
