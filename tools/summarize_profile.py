#!/usr/bin/env python3
"""Turns the raw rocprofv3 outputs of tools/profile_round.sh / tools/pmc_pass.sh (under gpurun_out/) into the small
tracked summaries under profiles/:   tools/summarize_profile.py <tag> [<pmc pass dir for MFMA/LDS counters>]
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats (verbatim)
  profiles/<tag>_pmc_hbm_traffic.csv   FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), full-batch launch = max over launches
  profiles/<tag>_pmc_mfma_lds.csv      per kernel: MFMA busy share of SIMD cycles, LDS busy share of CU cycles, bank-conflict share
"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


st = one("stats/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(dst, tag + "_kernel_stats.csv"))
# the --stats table averages over every launch of a kernel, including the batch-1 warm-up forward of hnet_create: the per-launch trace
# of the same run, restricted to the launches with the kernel's largest grid (= the batch-256 launches), is what bench.py's per-stage
# HIP events have to agree with
tr = one("stats/**/*kernel_trace.csv")
if tr:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(tr)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        per[r["Kernel_Name"]][g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for k, by_grid in per.items():
        g = max(by_grid)
        v = by_grid[g]
        rows.append((sum(v), k, g, len(v), sum(v) / len(v), min(v), max(v)))
    with open(os.path.join(dst, tag + "_kernel_stats_full_batch.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace of the same run as " + tag + "_kernel_stats.csv: launches with the kernel's largest grid only (batch 256)\n")
        f.write("kernel,grid_threads,launches,avg_us,min_us,max_us\n")
        for _t, k, g, n, a, lo, hi in sorted(rows, reverse=True):
            f.write('"%s",%d,%d,%.1f,%.1f,%.1f\n' % (k, g, n, a, lo, hi))


def per_kernel_max(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (max(v), len(v)) for k, v in acc.items()}


dg = one("csrc_digest.txt")
digest = open(dg).read().strip() if dg else "unknown"
fe, wr = one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv")
if fe and wr:
    f, w = per_kernel_max(fe, "FETCH_SIZE"), per_kernel_max(wr, "WRITE_SIZE")
    with open(os.path.join(dst, tag + "_pmc_hbm_traffic.csv"), "w") as out:
        out.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) of `python3 bench.py --steps 2 --warmup 1`, batch 256, default arithmetic mode of that build.\n")
        out.write("# Values are per launch at batch 256 (max over launches: the batch-1 warm-up launch of hnet_create is excluded). Raw counter values in KiB;\n")
        out.write("# gfx950 correction (MI355X_MICROARCH.md §HBM, re-calibrated for the access shapes of these kernels with tools/traffic_calib.hip on known\n")
        out.write("# byte counts - profiles/r02_traffic_calibration.log): FETCH_SIZE reports exactly half the bytes for 16-, 8- and 4-byte-per-lane loads and for\n")
        out.write("# LDS-DMA loads; WRITE_SIZE is exact for 16- and 8-byte-per-lane stores.  traffic_MB = (2 x FETCH_SIZE + WRITE_SIZE) / 1024: bytes that leave\n")
        out.write("# the XCD's L2 (Infinity-Cache hits are included).\n")
        out.write("# csrc_digest: " + digest + "   (tools/csrc_digest.py of the sources that were profiled; bench.py quotes this file only for the same sources)\n")
        cw = csv.writer(out)
        cw.writerow(["kernel", "launches", "FETCH_SIZE_KiB_full_batch_launch", "WRITE_SIZE_KiB_full_batch_launch", "traffic_MB_corrected"])
        for k in sorted(f):
            cw.writerow([k, f[k][1], f[k][0], w.get(k, (0, 0))[0], round((2 * f[k][0] + w.get(k, (0, 0))[0]) / 1024.0, 1)])

def mfma_lds_table(pass_dir, out_name, what):
    pm = glob.glob(os.path.join(ROOT, "gpurun_out", pass_dir, "**", "*counter_collection.csv"), recursive=True)
    if not pm:
        return
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pm[0])):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    best = {}
    for (k, g), cs in acc.items():          # keep the largest grid of each kernel (the full-batch launches)
        if k not in best or g > best[k][0]:
            best[k] = (g, {n: sum(v) / len(v) for n, v in cs.items()}, len(next(iter(cs.values()))))
    with open(os.path.join(dst, out_name), "w") as out:
        out.write("# rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (tools/profile_round.sh), " + what + ", averages over the full-batch launches.\n")
        out.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); lds_busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n")
        out.write("# csrc_digest: " + digest + "\n")
        cw = csv.writer(out)
        out.write("# valu_busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (4 SIMDs x SQ_BUSY_CU_CYCLES); valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA (wave instructions).\n")
        cw.writerow(["kernel", "grid_threads", "launches", "mfma_busy", "lds_busy", "lds_conflict_share", "valu_busy", "valu_per_mfma", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"])
        for k, (g, c, n) in sorted(best.items(), key=lambda kv: -kv[1][1].get("SQ_BUSY_CU_CYCLES", 0)):
            busy = c.get("SQ_BUSY_CU_CYCLES", 0)
            if busy <= 0 or "rocclr" in k:
                continue
            lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
            cw.writerow([k, g, n, round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * busy), 4), round(lds / busy, 4),
                         round(c.get("SQ_LDS_BANK_CONFLICT", 0) / lds, 4) if lds else 0,
                         round(c.get("SQ_ACTIVE_INST_VALU", 0) / busy, 4) if "SQ_ACTIVE_INST_VALU" in c else "",
                         round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2) if c.get("SQ_INSTS_MFMA") else "",
                         int(busy), int(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))])


mfma_lds_table(sys.argv[2] if len(sys.argv) > 2 else os.path.join("prof_" + tag, "mfma"), tag + "_pmc_mfma_lds.csv", "batch 256")
mfma_lds_table(os.path.join("prof_" + tag, "cfg3"), tag + "_cfg3_pmc_mfma_lds.csv", "SURVEY 8(d) config 3: prior-3 (three blocks + EKF prior), 64 pairs, N = 16")
print("profiles/ updated for", tag)
