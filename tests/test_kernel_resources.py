"""CPU: register budget of the built step kernel, read from the code object's metadata (no GPU needed).  k_step is pinned at four
market-waves per SIMD: 512 / 128 VGPRs; a change that pushes it over that, or into VGPR spills, halves the occupancy or adds
scratch traffic to every decimal operation - both cost more than any instruction they save (optimisation log)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _kernels():
    """metadata of every kernel + the disassembled body of every function of the built library"""
    import __graft_entry__ as G
    from kernel_resources import code_objects
    so = G.build_hip()
    co = os.path.join(ROOT, "gym_continuousdoubleauction_amd", "_kernel_resources.co")
    notes = asm = ""
    try:
        for blob in code_objects(so):                        # one code object per translation unit (env, PPO helpers, network)
            open(co, "wb").write(blob)
            notes += subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
            asm += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    finally:
        if os.path.exists(co):
            os.remove(co)
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                     for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    bodies, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = bodies.setdefault(m.group(1), [])
        elif cur is not None and line.strip():
            cur.append(line)
    return out, bodies


@pytest.mark.skipif(not (os.path.exists(READELF) and os.path.exists(OBJDUMP) and shutil.which(os.environ.get("HIPCC", "hipcc"))),
                    reason="needs hipcc and the ROCm LLVM tools")
def test_step_kernels_keep_four_waves_per_simd_and_spill_no_vgpr():
    ks, bodies = _kernels()
    steps = {n: v for n, v in ks.items() if "k_stepILb" in n}
    assert len(steps) == 4, sorted(ks)                     # two book tiles x with / without info tensors
    for n, v in steps.items():
        # the metadata is the maximum over the call graph: it includes the out-of-line general step (slow_step), which is free to
        # spill - what must hold is the occupancy (128 VGPRs = four waves per SIMD) and a HOT body without scratch traffic: at most
        # the one save / restore pair the compiler puts around the call of slow_step, on the branch a BASELINE market never takes
        assert v["vgpr_count"] <= 128, (n, v)
        body = bodies[n]
        scratch = [i for i, l in enumerate(body) if "scratch_" in l]
        assert len(scratch) <= 2, (n, len(scratch))
        calls = [i for i, l in enumerate(body) if "s_swappc" in l]
        for i in scratch:
            assert min(abs(i - c) for c in calls) <= 4, (n, body[i])
    for cap in ("cap256", "cap512"):
        slow = [n for n in bodies if cap in n and "slow_step" in n]
        assert len(slow) == 2, slow                          # the general build exists once per info variant, out of line
    for n, v in ks.items():
        if "k_run_random" in n or "k_reset" in n:                 # (the episode kernel keeps more state live and spills a few VGPRs)
            assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] <= 16, (n, v)


@pytest.mark.skipif(not (os.path.exists(READELF) and os.path.exists(OBJDUMP) and shutil.which(os.environ.get("HIPCC", "hipcc"))),
                    reason="needs hipcc and the ROCm LLVM tools")
def test_network_kernels_spill_nothing():
    """The MFMA kernels of csrc/cda_mlp.hip keep accumulators, operand rings and epilogue values in the 512-entry register file: a spill there
    is scratch traffic inside the k-loop (round 4: 137 spilled VGPRs made the training forward 1.5 x slower until its tail predicates went)."""
    ks, _ = _kernels()
    net = {n: v for n, v in ks.items() if any(k in n for k in ("k_mlp_fwd", "k_mlp_bwd", "k_mlp_wgrad"))}
    assert len(net) >= 10, sorted(ks)                      # forward: 3 tile sizes x 4 modes; backward: 3 tile sizes; one weight-gradient kernel
    for n, v in net.items():
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (n, v)
