#!/usr/bin/env python3
"""Condense the PMC passes of tools/profile_mlp.sh: per network kernel, averages per dispatch; HBM bytes with the calibration of
tools/pmc_calib.py (FETCH_SIZE: 2048 B per unit, WRITE_SIZE: 1024 B per unit on gfx950, profiles/r04/pmc_summary_4096x4_info1_g2.txt)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNELS = {"k_mlp_fb(": "k_mlp_fb", "k_mlp_wgrad": "k_mlp_wgrad", "k_grad_reduce": "k_grad_reduce", "k_adam": "k_adam", "k_mlp_fwd<1, 2>": "k_mlp_fwd<1, SAMPLE>",
           "k_mlp_fwd8": "k_mlp_fwd8", "k_mlp_bwd8": "k_mlp_bwd8", "k_ppo_loss32": "k_ppo_loss32", "k_prep_rows": "k_prep_rows"}
UNIT = {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0}
res = defaultdict(dict)
for sub in ("fetch", "write", "sq", "sq2"):
    for f in sorted(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                key = next((v for k, v in KERNELS.items() if k in name), None)
                if key is None:
                    continue
                acc[(key, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(key, row["Counter_Name"])] += 1
        for (k, c), v in acc.items():
            res[k][c] = v / cnt[(k, c)]
            res[k]["dispatches"] = cnt[(k, c)]
for k, d in res.items():
    if "FETCH_SIZE" in d: d["hbm_read_MB"] = d["FETCH_SIZE"] * UNIT["FETCH_SIZE"] / 1e6
    if "WRITE_SIZE" in d: d["hbm_write_MB"] = d["WRITE_SIZE"] * UNIT["WRITE_SIZE"] / 1e6
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        # SQ_BUSY_CYCLES counts per SE-level SQ; SQ_ACTIVE_INST_VALU sums over SIMDs (x4 cycles per instruction already): report the raw ratio and per-wave figures
        d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / max(1.0, d.get("SQ_WAVES", 1))
        d["mfma_insts_per_wave"] = d.get("SQ_INSTS_MFMA", 0) / max(1.0, d.get("SQ_WAVES", 1))
        d["active_inst_valu_over_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / max(1.0, d.get("SQ_WAVE_CYCLES", 1))
        d["mfma_busy_over_gpu_cycles"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, d.get("GRBM_GUI_ACTIVE", 1))
    print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in sorted(d.items())}))
with open(os.path.join(out, "mlp_pmc.json"), "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
