#!/bin/bash
# one rocprofv3 counter pass over bench.py on the GPU box:  tools/pmc_pass.sh <tag> <counter> [<counter> ...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-verify --no-extras --contexts 1 > $OUT.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
