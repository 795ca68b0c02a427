cd /tmp && export TMPDIR=/tmp
for t in 1 2; do
HNET_S3_PC=$t rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pc$t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > /dev/null 2>&1
python3 - <<PY
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_pc$t/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'pc_kernel' in r['Name'] or 'heads_prep' in r['Name']: print($t, r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
