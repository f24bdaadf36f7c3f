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
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
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

print("== PMC calibration: 1 GiB per launch, 4 B per lane coalesced (tools/pmc_calib.py) ==")
calib = {}
for sub, kern, ctr in (("calib_fetch", "k_calib_read", "FETCH_SIZE"), ("calib_write", "k_calib_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        vals = []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kern in row.get("Kernel_Name", "") and row["Counter_Name"] == ctr:
                    vals.append(float(row["Counter_Value"]))
        if vals:
            avg = sum(vals) / len(vals)
            calib[ctr] = (1 << 30) / avg
            print(f"{ctr}: avg counter per 1 GiB launch = {avg:.1f} -> {calib[ctr]:.1f} bytes per counter unit (n={len(vals)})")
import json
res = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        vals = []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_step" in row.get("Kernel_Name", "") and row["Counter_Name"] == ctr:
                    vals.append(float(row["Counter_Value"]))
        if vals:
            res[ctr] = sum(vals) / len(vals)
import re
meta = {}
for f in find("", "trace_bench.json"):
    try:
        with open(f) as fh:
            line = [ln for ln in fh if ln.startswith("{")][-1]
        b = json.loads(line)
        meta = {"markets": b["config"]["markets_per_gpu"], "agents": b["config"]["agents"], "info": b["config"]["info_outputs"],
                "groups": b["config"]["groups"], "markets_per_launch": b["roofline"]["markets_per_launch"]}
    except Exception as ex:  # noqa: BLE001
        print("no bench line in the trace pass:", ex)
insts = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
    for f in find(sub, "*counter_collection.csv"):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_step" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
        for k in acc:
            insts[k] = acc[k] / cnt[k]
if res and calib:
    fetch_b = res.get("FETCH_SIZE", 0) * calib.get("FETCH_SIZE", 1024)
    write_b = res.get("WRITE_SIZE", 0) * calib.get("WRITE_SIZE", 1024)
    mpl = meta.get("markets_per_launch") or 4096
    issued = sum(insts.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
    out_json = dict(meta, kernel="k_step", workload=f"bench.py, {meta.get('markets')} markets x {meta.get('agents')} agents, {meta.get('groups')} group chain(s), info {meta.get('info')}",
                    fetch_counter=res.get("FETCH_SIZE"), write_counter=res.get("WRITE_SIZE"), bytes_per_fetch_unit=calib.get("FETCH_SIZE"),
                    bytes_per_write_unit=calib.get("WRITE_SIZE"), fetch_bytes_per_launch=fetch_b, write_bytes_per_launch=write_b,
                    hbm_bytes_per_launch=fetch_b + write_b, hbm_bytes_per_market_step=(fetch_b + write_b) / mpl,
                    wave_insts_per_market_step=issued / mpl, valu_insts_per_market_step=insts.get("SQ_INSTS_VALU", 0.0) / mpl,
                    salu_insts_per_market_step=insts.get("SQ_INSTS_SALU", 0.0) / mpl, lds_insts_per_market_step=insts.get("SQ_INSTS_LDS", 0.0) / mpl,
                    active_inst_any_over_wave_cycles=(insts.get("SQ_ACTIVE_INST_ANY", 0.0) / insts["SQ_WAVE_CYCLES"]) if insts.get("SQ_WAVE_CYCLES") else None,
                    wait_any_over_wave_cycles=(insts.get("SQ_WAIT_ANY", 0.0) / insts["SQ_WAVE_CYCLES"]) if insts.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in insts else None)
    print("k_step HBM traffic and issue picture per launch (calibrated):", json.dumps(out_json))
    # named by the shape it was taken on: bench.py reports roofline.traffic / issue_frac only for EXACTLY that shape (profiles/pmc/)
    name = f"{meta.get('markets')}x{meta.get('agents')}_info{int(bool(meta.get('info')))}_g{meta.get('groups')}.json"
    with open(os.path.join(out, name), "w") as fh:
        json.dump(out_json, fh, indent=1)
