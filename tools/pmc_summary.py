#!/usr/bin/env python3
"""per-kernel averages of a rocprofv3 --pmc pass (csv output dir given as argv[1]); launches with the full-batch grid only"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"hnet::", "", k)
    k = re.sub(r"\(.*$", "", k)
    acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
rows = []
for (k, g), cs in acc.items():
    names = sorted(cs)
    rows.append((k, g, {n: sum(v) / len(v) for n, v in cs.items()}, len(next(iter(cs.values())))))
names = sorted({n for _k, _g, c, _n in rows for n in c})
print("kernel".ljust(70), "grid".rjust(9), "n".rjust(4), " ".join(n[-22:].rjust(22) for n in names))
for k, g, c, n in sorted(rows, key=lambda r: -max(r[2].values())):
    if n < 2: continue
    print(k[:70].ljust(70), str(g).rjust(9), str(n).rjust(4), " ".join(f"{c.get(x, float('nan')):22.4g}" for x in names))
