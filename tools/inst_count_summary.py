"""Reads the counter_collection.csv of one tools/inst_count_probe.py run: averages over the LAST 100 k_step dispatches."""
import csv
import glob
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
rows = defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_step" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)[-100:]
avg = {k: sum(rows[i].get(k, 0.0) for i in ids) / len(ids) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")}
w = avg["SQ_WAVES"] or 1
print(f"{tag:44s} VALU {avg['SQ_INSTS_VALU'] / w:7.0f}  SALU {avg['SQ_INSTS_SALU'] / w:7.0f}  LDS {avg['SQ_INSTS_LDS'] / w:6.0f}  total {(avg['SQ_INSTS_VALU'] + avg['SQ_INSTS_SALU'] + avg['SQ_INSTS_LDS']) / w:7.0f}   per market-step (n={len(ids)})")
