#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            row["Name"] = row["Name"][:70]
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== kernel trace: per-dispatch resources of k_step ==")
for f in find("trace", "*kernel_trace.csv"):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if "k_step" in r.get("Kernel_Name", "")]
    if rows:
        r = rows[-1]
        keep = ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")
        print({k: r.get(k) for k in keep})
        d = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows]
        print(f"k_step dispatches={len(d)} avg_ns={sum(d)/len(d):.0f} min_ns={min(d)} max_ns={max(d)}")
print("== PMC (per k_step dispatch averages) ==")
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_step" not in row.get("Kernel_Name", ""):
                    continue
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[row["Counter_Name"]] += 1
        for k in sorted(acc):
            print(f"{sub}: {k} avg/dispatch = {acc[k]/cnt[k]:.1f}  (n={cnt[k]})")
