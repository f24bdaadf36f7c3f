"""CPU: register budget of the built step kernel, read from the code object's metadata (no GPU needed).  k_step is pinned at four
market-waves per SIMD: 512 / 128 VGPRs; a change that pushes it over that, or into VGPR spills, halves the occupancy or adds
scratch traffic to every decimal operation - both cost more than any instruction they save (optimisation log)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernels():
    import __graft_entry__ as G
    from kernel_resources import code_object
    so = G.build_hip()
    co = os.path.join(ROOT, "gym_continuousdoubleauction_amd", "_kernel_resources.co")
    try:
        open(co, "wb").write(code_object(so))
        notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
    finally:
        if os.path.exists(co):
            os.remove(co)
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                     for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return out


@pytest.mark.skipif(not os.path.exists(READELF), reason="needs the ROCm LLVM tools")
def test_step_kernels_keep_four_waves_per_simd_and_spill_no_vgpr():
    ks = _kernels()
    steps = {n: v for n, v in ks.items() if "k_stepILb" in n}
    assert len(steps) == 4, sorted(ks)                     # two book capacities x with / without info tensors
    for n, v in steps.items():
        assert v["vgpr_count"] <= 128, (n, v)
        assert v["vgpr_spill_count"] == 0, (n, v)
        assert v["private_segment_fixed_size"] <= 128, (n, v)
    for cap in ("cap256", "cap512"):
        plain = next(v for n, v in steps.items() if cap in n and "ILb0" in n)
        info = next(v for n, v in steps.items() if cap in n and "ILb1" in n)
        assert plain["sgpr_spill_count"] < info["sgpr_spill_count"], (cap, plain, info)   # the info-less instance carries no info pointers
    for n, v in ks.items():
        if "k_run_random" in n or "k_reset" in n:                 # (the episode kernel keeps more state live and spills a few VGPRs)
            assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] <= 16, (n, v)
