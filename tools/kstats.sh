#!/bin/bash
# per-kernel durations of the default bench step on the GPU box:  tools/kstats.sh <tag> [extra bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-extras --contexts 1 "$@" > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*$", "", re.sub(r"hnet::", "", re.sub(r"^void ", "", r["Kernel_Name"])))
    acc[(k, int(r.get("Grid_Size") or r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
rows = sorted(((k, g, len(v), sum(v) / len(v)) for (k, g), v in acc.items() if len(v) >= 5), key=lambda r: -r[3] * r[2])
for k, g, n, a in rows[:45]:
    print(f"{a:9.1f} us x{n:4d}  grid {g:8d}  {k[:110]}")
PY
