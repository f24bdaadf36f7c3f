#!/usr/bin/env python3
"""Cycle stamps of ONE workgroup of the network's forward kernel (the CDA_MLP_TIMING build inside tools/libcda_tools.so): where a tile's
time goes - observation load, the MFMA loops, the tanh / store epilogues, the barriers.  Usage: python tools/mlp_timing.py [--mt 4] [--sample]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import mlp  # noqa: E402

NAMES = {0: "start", 1: "x tile in LDS", 18: "outputs written", 19: "end"}
for h in (0, 1):
    NAMES.update({2 + 8 * h: f"half {h}: layer 1 MFMAs", 3 + 8 * h: f"half {h}: layer 1 epilogue", 4 + 8 * h: f"half {h}: barrier", 5 + 8 * h: f"half {h}: layer 2 MFMAs",
                  6 + 8 * h: f"half {h}: barrier + layer 2 epilogue", 7 + 8 * h: f"half {h}: barrier", 8 + 8 * h: f"half {h}: heads"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mt", type=int, default=4)
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--sample", action="store_true")
    ap.add_argument("--waves8", action="store_true", help="the 8-wave training forward (both halves in flight, 64 rows per workgroup)")
    ap.add_argument("--fused", action="store_true", help="the fused gather + forward + loss + backward kernel of the update")
    ap.add_argument("--block", type=int, default=7)
    a = ap.parse_args()
    T = C.CDLL(os.path.join(ROOT, "tools", "libcda_tools.so"))
    dev = torch.device("cuda:0")
    p = mlp.FusedPolicy(dev, seed=1)
    R = a.rows
    upd = mlp.FusedUpdate(p, R, R, 4)
    obs = torch.randn(R, 168, device=dev)
    from gym_continuousdoubleauction_amd._lib import check, lib
    check(lib().cda_mlp_prep_rows(obs.data_ptr(), None, R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")
    dbg = torch.zeros(8 * 32, dtype=torch.int64, device=dev)
    scratch = torch.zeros(R * 4 * 40, dtype=torch.uint8, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    vp = C.c_void_p
    if a.fused:
        i64, i32, f32 = C.c_int64, C.c_int32, C.c_float
        T.cda_tools_mlp_fb_dbg.argtypes = [vp, i32]; T.cda_tools_mlp_fb_dbg.restype = None
        T.cda_mlp_forward_backward.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, i64, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
        A = 4
        rec = torch.zeros(R, A, 8, device=dev)
        rec[..., 0:3] = torch.randint(0, 3, (R, A, 3), device=dev).int().view(torch.float32)
        rec[..., 3:5] = torch.randn(R, A, 2, device=dev); rec[..., 5] = -7.0; rec[..., 6:8] = torch.randn(R, A, 2, device=dev)
        perm = torch.randperm(R, device=dev)
        T.cda_tools_mlp_fb_dbg(dbg.data_ptr(), a.block)
        for rep in range(3):
            rc = T.cda_mlp_forward_backward(p.wb.data_ptr(), p.theta.data_ptr(), obs.data_ptr(), perm.data_ptr(), R, R, rec.data_ptr(), None, 0, A, 0.2, 0.5, 0.01, None,
                                            upd.x_pk_mb.data_ptr(), upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.dz1p.data_ptr(), upd.dz2p.data_ptr(), upd.doutp.data_ptr(),
                                            upd.bias_slab.data_ptr(), upd.sums5.data_ptr(), upd.out6.data_ptr(), 1, 0, None, None, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            torch.cuda.synchronize()
        d = dbg.cpu().view(8, 32)
        names = ["gather -> LDS", "x_pk", "M1", "E1", "barrier", "M2", "barrier", "E2", "barrier", "MH", "barrier", "loss", "barrier", "(doutp) MdH2 + E", "barrier", "MdH1", "E"]
        print(f"fused gather + forward + loss + backward, one network half of a 64-row tile per 4-wave workgroup, {R} rows; tile {a.block}; shader cycles per segment, "
              f"waves 0..3 of the policy workgroup, then of the value workgroup")
        for k, nm in enumerate(names):
            print(f"  {nm:20s} " + " ".join(f"{int(d[w, k + 1] - d[w, k]):7d}" for w in range(8)))
        print(f"  {'total':20s} " + " ".join(f"{int(d[w, 17] - d[w, 0]):7d}" for w in range(8)))
        print(f"  {'(MdH2 alone)':20s} " + " ".join(f"{int(d[w, 21] - d[w, 13]):7d}" for w in range(8)))
        t0 = int(d[:, 0].min())
        print(f"  {'(start, relative)':20s} " + " ".join(f"{int(d[w, 0]) - t0:7d}" for w in range(8)))
        return
    T.cda_tools_mlp_fwd_timing.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    for rep in range(3):
        rc = T.cda_tools_mlp_fwd_timing(p.wb.data_ptr(), p.theta.data_ptr(), upd.x_rm.data_ptr(), obs.data_ptr(), R, 4, upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.out.data_ptr(),
                                        scratch.data_ptr(), counter.data_ptr(), a.mt, 2 if a.waves8 else int(a.sample), dbg.data_ptr(), a.block, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
    if a.waves8:
        d = dbg.cpu().view(8, 32)
        names = ["x tile in LDS", "stagger barrier", "M1", "barrier", "E1", "barrier", "M2", "barrier", "E2", "barrier", "MH", "outputs", "final barrier"]
        print(f"8-wave training forward, 64 rows per workgroup, {R} rows; workgroup {a.block}; shader cycles per segment, waves 0..7 (0-3 policy, 4-7 value)")
        for k, nm in enumerate(names):
            print(f"  {nm:20s} " + " ".join(f"{int(d[w, k + 1] - d[w, k]):7d}" for w in range(8)))
        print(f"  {'total':20s} " + " ".join(f"{int(d[w, 13] - d[w, 0]):7d}" for w in range(8)))
        return
    d = dbg.cpu().view(8, 32)
    print(f"forward kernel, {'sampling' if a.sample else 'training'} mode, {32 * a.mt} rows per workgroup, {R} rows; workgroup {a.block}; shader cycles per segment, waves 0..3")
    keys = sorted(NAMES)
    for prev, k in zip(keys, keys[1:]):
        seg = [int(d[w, k] - d[w, prev]) for w in range(4)]
        print(f"  {NAMES[k]:40s} " + " ".join(f"{x:8d}" for x in seg))
    print(f"  {'total':40s} " + " ".join(f"{int(d[w, 19] - d[w, 0]):8d}" for w in range(4)))


if __name__ == "__main__":
    main()
