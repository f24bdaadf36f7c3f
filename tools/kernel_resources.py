#!/usr/bin/env python3
"""Register / spill / scratch figures of the kernels in a built libcda_hip.so, and optionally the gfx950 disassembly.
Runs without a GPU.  Usage: python tools/kernel_resources.py [lib.so] [--asm out.s] [--filter k_step]"""
import argparse
import os
import re
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so):
    """Every AMDGPU ELF embedded in the host library's .hip_fatbin section (one per translation unit)."""
    d = open(so, "rb").read()
    out, i = [], 0
    while True:
        i = d.find(b"\x7fELF", i)
        if i < 0:
            break
        if struct.unpack_from("<H", d, i + 18)[0] == 224:                       # EM_AMDGPU
            shoff = struct.unpack_from("<Q", d, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", d, i + 58)
            out.append(d[i:i + shoff + shentsize * shnum])
            i += shoff + shentsize * shnum
        else:
            i += 4
    if not out:
        raise SystemExit("no AMDGPU code object in " + so)
    return out


def code_object(so):
    """the first one (the env kernels: cda_hip.hip)"""
    return code_objects(so)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "gym_continuousdoubleauction_amd", "libcda_hip.so"))
    ap.add_argument("--asm", help="write the disassembly here")
    ap.add_argument("--filter", default="", help="only kernels whose mangled name contains this")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        notes, cos = "", []
        for k, blob in enumerate(code_objects(a.lib)):
            co = os.path.join(td, f"cda{k}.co")
            open(co, "wb").write(blob)
            cos.append(co)
            notes += subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or a.filter not in name.group(1):
                continue
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
            print("%-70s vgpr %s sgpr %s  spills v %s s %s  scratch %s B  lds %s B" % (
                name.group(1), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        if a.asm:
            with open(a.asm, "w") as f:
                for co in cos:
                    subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=f, check=True)


if __name__ == "__main__":
    main()
