#!/bin/bash
# Collects the round's rocprof evidence on the GPU box (run through gpurun):  tools/profile_round.sh <tag>
#   stats : rocprofv3 --kernel-trace --stats of the default bench command
#   fetch / write : FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots: 3 + 2 of 4)
#   mfma  : SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT + instruction counts
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
# (--contexts 1: per-kernel durations and counters of ONE stream of launches - with the default two contexts kernels of different steps overlap and stretch each other)
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency --no-verify --no-extras --contexts 1"
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/csrc_digest.py > $OUT/csrc_digest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 5 --warmup 2 > $OUT.stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B --steps 2 --warmup 1 > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B --steps 2 --warmup 1 > $OUT.write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- $B --steps 2 --warmup 1 > $OUT.mfma.log 2>&1
ls -R $OUT | head -40
# SURVEY 8(d) config 3 (prior-3, 64 pairs, N = 16): MFMA / LDS counters per kernel
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/cfg3 -- $B --steps 2 --warmup 1 --variant prior3 --batch 64 --mc 16 > $OUT.cfg3.log 2>&1
