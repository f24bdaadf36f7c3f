#!/usr/bin/env python3
"""VERDICT r4 #4, part one: where does k_step's 40 % wait share come from?  Condenses three rocprofv3 --pmc passes over `bench.py --steps 200 --repeats 1`
(run by tools/r5_kstep_wait_attribution.sh) into one table: cycles of a market-wave by what the SQ says it was doing.  Usage: kstep_wait_attribution.py <dir with pmc_w1..3>"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
vals = {}
for sub in ("pmc_w1", "pmc_w2", "pmc_w3"):
    for f in sorted(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name", "")
                if "k_step" not in kn or not ("ILb1" in kn or "<true>" in kn):
                    continue
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
        for k in acc:
            vals[k] = acc[k] / cnt[k]
if "SQ_WAVE_CYCLES" not in vals:
    raise SystemExit("no k_step<true> rows found under " + out)
wc, waves = vals["SQ_WAVE_CYCLES"], vals.get("SQ_WAVES", 0.0)
print(f"k_step<true>, per launch averages: {waves:.0f} waves, SQ_WAVE_CYCLES {wc:.3e} (= {wc / max(waves, 1):.0f} cycles per market-wave, x4 on gfx950's counter = quad-cycles)")
print("share of the wave-cycles (a wave is in exactly one of: issuing an instruction of some type / waiting for an instruction to finish / waiting for anything else):")
for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC",
          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR"):
    if k in vals:
        print(f"  {k:26s} {vals[k]:.3e}   {100 * vals[k] / wc:6.2f} % of SQ_WAVE_CYCLES")
print("instructions per market-wave:")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"):
    if k in vals and waves:
        print(f"  {k:26s} {vals[k] / waves:8.1f}")
