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
        vals = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
        # the network kernels exist once per compiled history depth under the same name (one code object each): keep the worst figure of each kind
        out[name] = {k: max(v, out[name][k]) for k, v in vals.items()} if name in out else vals
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
    assert len(steps) == 8, sorted(ks)                     # two book tiles x with / without info tensors x with / without the episode-metric tallies
    for n, v in steps.items():
        # the metadata is the maximum over the call graph: it includes the out-of-line general step (slow_step), which is free to
        # spill - what must hold is the occupancy (128 VGPRs = four waves per SIMD) and a HOT body without scratch traffic: at most
        # the one save / restore pair the compiler puts around the call of slow_step, on the branch a BASELINE market never takes
        assert v["vgpr_count"] <= 128, (n, v)
        body = bodies[n]
        # ... around the calls in the kernel's cold tail (slow_step hands "the episode ended" back; the kernel then calls the episode-end routines from its own level)
        scratch = [i for i, l in enumerate(body) if "scratch_" in l]
        assert len(scratch) <= 6, (n, len(scratch))
        calls = [i for i, l in enumerate(body) if "s_swappc" in l]
        for i in scratch:
            assert min(abs(i - c) for c in calls) <= 24 and i > len(body) - 200, (n, i, len(body), body[i])
        # the instances launched while the metrics are off carry no tally: not one atomic in them
        if "k_stepILb0ELb0" in n or "k_stepILb1ELb0" in n:
            assert not any("atomic_add_f64" in l for l in body), n
        else:
            assert sum("atomic_add_f64" in l for l in body) == 11, n
    for cap in ("cap256", "cap512"):
        slow = [n for n in bodies if cap in n and "slow_step" in n]
        assert len(slow) == 2, slow                          # the general build exists once per info variant, out of line
    for n, v in ks.items():
        if "k_run_random" in n or "k_reset" in n:                 # (the episode kernel keeps more state live and spills a few VGPRs; more in the instance that tallies)
            assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] <= (32 if "k_run_randomILb1" in n else 16), (n, v)
    # the policy inside the step kernel: sixteen market-waves per workgroup = the same four waves per SIMD, and the same hot body as k_step<false> behind the forward pass
    pol = {n: v for n, v in ks.items() if "k_policy_step" in n}
    assert len(pol) == 14, sorted(pol)                           # one per compiled history depth (256-order tile only) x with / without the tallies
    for n, v in pol.items():
        assert v["vgpr_count"] <= 128 and v["vgpr_spill_count"] <= 2, (n, v)          # (the save / restore around the cold tail's calls)
        body = bodies[n]
        depth_k1 = {"k_policy_stepI": 11, "k_policy_step_h1I": 3, "k_policy_step_h2I": 6, "k_policy_step_h3I": 8, "k_policy_step_h6I": 16, "k_policy_step_h7I": 19, "k_policy_step_h8I": 21}
        k1 = depth_k1[[k for k in depth_k1 if k in n][0]]
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in body) == k1 + 16 + 16, n      # one tile per wave: layer 1 (KX / 16 k-steps), layer 2, heads
        scratch = [i for i, l in enumerate(body) if "scratch_" in l]
        calls = [i for i, l in enumerate(body) if "s_swappc" in l]
        assert len(scratch) <= 6 and all(min(abs(i - c) for c in calls) <= 24 and i > len(body) - 200 for i in scratch), (n, len(scratch))


@pytest.mark.skipif(not (os.path.exists(READELF) and os.path.exists(OBJDUMP) and shutil.which(os.environ.get("HIPCC", "hipcc"))),
                    reason="needs hipcc and the ROCm LLVM tools")
def test_network_kernels_spill_nothing():
    """The MFMA kernels of csrc/cda_mlp.hip keep accumulators, operand rings and epilogue values in the 512-entry register file: a spill there
    is scratch traffic inside the k-loop (round 4: 137 spilled VGPRs made the training forward 1.5 x slower until its tail predicates went)."""
    ks, _ = _kernels()
    net = {n: v for n, v in ks.items() if any(k in n for k in ("k_mlp_fwd", "k_mlp_bwd", "k_mlp_wgrad", "k_mlp_fb", "k_grad_reduce", "k_adam", "k_gae_records", "k_grad_norm",
                                                               "k_league_assign", "k_episode_returns"))}
    assert len(net) >= 20, sorted(ks)                      # forward: 3 tile sizes x 5 modes; backward: 3 tile sizes; the fused update kernel; weight gradients; reduce; Adam; GAE; ...
    for n, v in net.items():
        assert v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (n, v)
    # the update's dominant kernel is built for TWO workgroups per CU: 4 waves x 2 = 2 waves per SIMD = 256 registers per wave, and 160 KB / 2 of LDS
    fb = [v for n, v in ks.items() if "k_mlp_fb" in n]
    assert len(fb) == 1 and fb[0]["vgpr_count"] <= 256, fb


def test_fused_update_kernel_fits_two_workgroups_per_cu():
    """k_mlp_fb's dynamic LDS (csrc/cda_mlp.hip fb_lds: observation tile + activation tile + log-probabilities + the tile's sample records [+ the rollout policy's
    distributions for the KL term]) must stay under half of the CU's 160 KB for every shape the loops launch: one shared policy at 4 and 8 agents per row (the
    latter without the KL rows: DESIGN 4.2 states that cliff), a league update (one sample per row) with and without them."""
    ACT_LD, LPS_LD = 256 + 8, 23

    def fb_lds(agents, dist, n_hist=4):
        kx = (42 * n_hist + 15) // 16 * 16
        xs = max(64 * (kx + 8) * 2, 64 * 33 * 4 + 64 * 40 * 2)     # FB_XS_BYTES: the observation tile, whose bytes later hold the output tiles
        return xs + 64 * ACT_LD * 2 + 64 * LPS_LD * 4 + 64 * (agents * 32 + (28 * 4 if dist else 0))           # (a row's distribution: CDA_MLP_DIST_LD floats)
    src = open(os.path.join(ROOT, "gym_continuousdoubleauction_amd", "csrc", "cda_mlp.hip")).read()
    assert "size_t fb_lds(int agents, bool with_dist) { return (size_t)FB_XS_BYTES + (size_t)64 * ACT_LD * 2 + (size_t)64 * LPS_LD * 4 + (size_t)64 * (agents * 32 + (with_dist ? CDA_MLP_DIST_LD * 4 : 0)); }" in src
    assert "constexpr int FB_XS_BYTES = (64 * XS_LD * 2 > 64 * OUTS_LD * 4 + 64 * DO_LD * 2) ? 64 * XS_LD * 2 : 64 * OUTS_LD * 4 + 64 * DO_LD * 2;" in src
    assert fb_lds(4, False) == 71424
    for agents, dist in ((4, False), (4, True), (8, False), (1, False), (1, True)):
        for n_hist in (1, 2, 4):
            assert fb_lds(agents, dist, n_hist) <= 80 * 1024, (agents, dist, n_hist, fb_lds(agents, dist, n_hist))
    assert fb_lds(8, True) > 80 * 1024                       # the stated cliff: 8 agents per row AND the KL rows -> one workgroup per CU
    assert 80 * 1024 < fb_lds(4, False, 8) <= 160 * 1024     # ... and n_hist = 8 (a 44-KB observation tile): one workgroup per CU at every agent count
