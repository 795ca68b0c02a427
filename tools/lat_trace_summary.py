#!/usr/bin/env python3
"""per-kernel duration and the gap to the previous kernel, averaged over the last forwards of a rocprofv3 --kernel-trace of tools/lat_trace.py:  python tools/lat_trace_summary.py <dir> <kernels per forward>"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
npf = int(sys.argv[2])
rows = rows[-npf * 40:]          # the last 40 forwards
name = lambda r: re.sub(r"\(.*$", "", re.sub(r"hnet::", "", re.sub(r"^void ", "", r["Kernel_Name"])))[:60]
acc = collections.OrderedDict()
for i in range(1, len(rows)):
    k = (i % npf, name(rows[i]))
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
    acc.setdefault(k, []).append((d, g))
tot_d = tot_g = 0
for (pos, n), v in sorted(acc.items()):
    d = sum(x[0] for x in v) / len(v); g = sum(x[1] for x in v) / len(v)
    tot_d += d; tot_g += g
    print(f"{pos:3d} {n:62s} dur {d:6.2f} us  gap before {g:6.2f} us  (n={len(v)})")
print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, forward {tot_d + tot_g:.1f} us")
