#!/usr/bin/env python3
"""Static instruction count per source line of one function of the device assembly (build container, no GPU):
    hipcc <flags of __graft_entry__> -gline-tables-only -S --cuda-device-only -o build_tmp/cda_hip_g.s csrc/cda_hip.hip
    python tools/isa_line_profile.py build_tmp/cda_hip_g.s <mangled function name substring> [top]
Counts every machine instruction against the innermost .loc in force (inlined code is attributed to ITS source line)."""
import collections
import re
import sys

path, func = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
per_line = collections.Counter()
per_kind = collections.Counter()
inside = False
cur = None
total = 0
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if not inside:
        if re.match(r"^[A-Za-z_][\w$.]*:", ln) and func in ln and not ln.startswith(".L"):
            inside = True
        continue
    if s.startswith(".Lfunc_end"):
        break
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    total += 1
    per_line[cur] += 1
    per_kind[op.split("_")[0] if not op.startswith("s_waitcnt") else "s_waitcnt"] += 1
print(f"{func}: {total} instructions")
print("by class:", dict(per_kind.most_common(12)))
for (f, l), n in per_line.most_common(top):
    print(f"{n:6d}  {files.get(f, f)}:{l}")
