#!/usr/bin/env python3
"""VERDICT r4 #3, measured with knobs instead of a rebuild: what would folding the W2 weight gradient into the update's forward / backward kernel buy?

The fused construction: a PERSISTENT workgroup keeps its network half's dW2 (256 x 256 f32 = 256 KB: 256 accumulator registers in each of its four waves) across its
row tiles, so h1 and dz2 never travel to HBM and k_mlp_wgrad loses its two dW2 jobs.  Its cost: 256 more registers per wave = ONE workgroup per CU (one wave per SIMD: no
second wave's MFMAs under a wave's tanh / loss phases), 64 more MFMAs per wave and tile, and one 256-KB partial per workgroup at the end.  Each term is measured with
the CDA_MLP_TIMING build's knobs (tools/libcda_tools.so, cda_tools_mlp_experiment):
    k_mlp_fb as built                                             (two workgroups per CU, all stores)
    k_mlp_fb forced to one workgroup per CU                       (16 KB of extra dynamic LDS: nothing else changes)
    k_mlp_fb without the h1p / dz2p stores                        (what the fused kernel would not write: 134 MB per 65 536 rows)
    ... + 64 MFMAs per wave and tile fed from LDS                 (the arithmetic of dW2 = dz2^T h1)
    k_mlp_wgrad as built / without its two dW2 jobs
An OPTIMISTIC estimate of the fused step = fb(one workgroup per CU, no h1p / dz2p stores, + the MFMAs) + wgrad(no dW2) + reduce + Adam - optimistic because the
persistent loop itself (round 4: +11 us, address hoisting against a full register file) and the 64-MB partial write at the end are not charged.

    python tools/fb_wgrad_fusion_probe.py [--rows 65536] [--agents 4] > profiles/r05/wgrad_fusion_experiment.txt
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import mlp  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--agents", type=int, default=4)
    a = ap.parse_args()
    T = C.CDLL(os.environ.get("CDA_TOOLS_LIB") or os.path.join(ROOT, "tools", "libcda_tools.so"))
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
    T.cda_tools_mlp_experiment.argtypes = [i32, i32, i32]; T.cda_tools_mlp_experiment.restype = None
    T.cda_mlp_forward_backward.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, i64, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    T.cda_mlp_wgrad.argtypes = [vp] * 6 + [i64, i32, vp, vp]
    T.cda_mlp_adam.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i64, f32, f32, f32, vp, f32, f32, f32, f32, f32, vp, vp, vp]
    dev = torch.device("cuda:0")
    R, A = a.rows, a.agents
    p = mlp.FusedPolicy(dev, seed=1)
    upd = mlp.FusedUpdate(p, R, R, A)
    obs = torch.randn(R, 168, device=dev)
    rec = torch.zeros(R, A, 8, device=dev)
    rec[..., 0:3] = torch.randint(0, 3, (R, A, 3), device=dev).int().view(torch.float32)
    rec[..., 3:5] = torch.randn(R, A, 2, device=dev); rec[..., 5] = -7.0; rec[..., 6:8] = torch.randn(R, A, 2, device=dev)
    torch.randperm(R, device=dev, out=upd.perm)
    st = torch.cuda.current_stream().cuda_stream
    chunks, tiles = upd.chunks, (R + 63) // 64

    def fb():
        rc = T.cda_mlp_forward_backward(p.wb.data_ptr(), p.theta.data_ptr(), obs.data_ptr(), upd.perm.data_ptr(), R, R, rec.data_ptr(), None, 0, A, 0.2, 0.5, 0.01, None,
                                        upd.x_pk_mb.data_ptr(), upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.dz1p.data_ptr(), upd.dz2p.data_ptr(), upd.doutp.data_ptr(),
                                        upd.bias_slab.data_ptr(), upd.sums5.data_ptr(), upd.out6.data_ptr(), 1, 0, None, None, st)
        assert rc == 0, rc

    def wg():
        rc = T.cda_mlp_wgrad(upd.x_pk_mb.data_ptr(), upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.dz1p.data_ptr(), upd.dz2p.data_ptr(), upd.doutp.data_ptr(), R, chunks, upd.slab.data_ptr(), st)
        assert rc == 0, rc

    def opt():
        rc = T.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), upd.slab.data_ptr(), chunks, upd.bias_slab.data_ptr(), tiles,
                            upd.sums5.data_ptr(), R * A, 0.5, 0.01, 0.0, upd.out6.data_ptr(), 0.0, 0.9, 0.999, 1e-8, 0.5, upd.grad.data_ptr(), upd.norm2.data_ptr(), st)
        assert rc == 0, rc

    def both():
        fb(); wg(); opt()
    out = {}
    PAD = 16 * 1024                  # 71 KB + 16 KB > half of the CU's 160 KB: one workgroup per CU
    for name, flags, pad, first_job in (("as built", 0, 0, 0), ("fb: one workgroup per CU", 0, PAD, 0), ("fb: no h1p / dz2p stores", 1, 0, 0),
                                        ("fb: no h1p / dz2p stores + 64 MFMAs per wave-tile", 3, 0, 0), ("fb: one workgroup per CU, no h1p / dz2p stores", 1, PAD, 0),
                                        ("fb: one workgroup per CU, no h1p / dz2p stores + 64 MFMAs per wave-tile; wgrad without dW2", 3, PAD, 2)):
        T.cda_tools_mlp_experiment(flags, pad, first_job)
        out[name] = (timed(fb), timed(wg), timed(opt), timed(both))
    T.cda_tools_mlp_experiment(0, 0, 0)
    print(f"k_mlp_fb / k_mlp_wgrad / reduce + Adam / the three back to back, us per launch; {R} rows x {A} agents per row, {chunks} weight-gradient chunks")
    for name, (t_fb, t_wg, t_opt, t_all) in out.items():
        print(f"  {name:100s} fb {t_fb:7.1f}   wgrad {t_wg:6.1f}   reduce+adam {t_opt:5.1f}   step {t_all:7.1f}")
    base, best = out["as built"][3], out["fb: one workgroup per CU, no h1p / dz2p stores + 64 MFMAs per wave-tile; wgrad without dW2"][3]
    print(f"optimistic fused-dW2 step: {best:.1f} us against {base:.1f} us as built: {100 * (base - best) / base:+.1f} % "
          "(not charged: the persistent loop, the 64-MB partial write of 256 workgroups x 256 KB, their reduction)")


if __name__ == "__main__":
    main()
