#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats: the kernel_stats.csv of a run as a short table (kernel, calls, avg / min / max us, share).  Usage: kernel_stats_table.py <csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print(f"{'kernel':70s} {'calls':>7s} {'avg us':>9s} {'min us':>8s} {'max us':>8s} {'% of GPU time':>14s}")
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    print(f"{name[:70]:70s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:9.2f} {float(r['MinNs']) / 1e3:8.2f} {float(r['MaxNs']) / 1e3:8.2f} {float(r['Percentage']):14.2f}")
