#!/usr/bin/env python3
"""Per-kernel timing of the network kernels (include/cda_mlp.h) on the shapes of BASELINE configs[4]: the update's minibatch (65536 rows =
262144 samples at 4 agents) and the rollout's chain (1024 rows).  HIP events around K back-to-back launches; useful FLOP counts exclude the
zero padding.  Usage: python tools/mlp_bench.py [--rows 65536] [--iters 20] [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import mlp  # noqa: E402
from gym_continuousdoubleauction_amd._lib import check, lib  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, A = a.rows, a.agents
    p = mlp.FusedPolicy(dev, seed=1)
    upd = mlp.FusedUpdate(p, R, R, A, chunks=a.chunks or None)
    g = torch.Generator().manual_seed(0)
    obs = (torch.randn(R, 168, generator=g)).to(dev)
    B = R * A
    acts = (torch.randint(0, 9, (B,), generator=g).int().to(dev), torch.randint(0, 10, (B,), generator=g).int().to(dev), torch.randint(0, 3, (B,), generator=g).int().to(dev),
            torch.randn(B, 2, generator=g).to(dev))
    lp_old, adv, ret = (torch.randn(B, generator=g) * 0.1 - 7).to(dev), torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    torch.randperm(R, device=dev, out=upd.perm)
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    tile = int(L.cda_mlp_tile_rows()); tiles = (R + tile - 1) // tile
    chunks = upd.chunks
    res = {"rows": R, "agents": A, "tile_rows": tile, "chunks": chunks}
    res["prep_rows_us"] = timed(lambda: check(L.cda_mlp_prep_rows(obs.data_ptr(), upd.perm.data_ptr(), R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), st), "prep"), a.iters)
    res["forward_train_us"] = timed(lambda: check(L.cda_mlp_forward_train(p.wb.data_ptr(), p.theta.data_ptr(), upd.x_rm.data_ptr(), R, upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.out.data_ptr(), st), "fwd"), a.iters)
    res["loss_us"] = timed(lambda: check(L.cda_ppo_loss32(upd.out.data_ptr(), p.theta.data_ptr() + mlp.OFF_LS * 4, acts[0].data_ptr(), acts[1].data_ptr(), acts[2].data_ptr(), acts[3].data_ptr(),
                                                         lp_old.data_ptr(), adv.data_ptr(), ret.data_ptr(), upd.perm.data_ptr(), R, A, 32, 0.2, 0.5, 0.01, upd.d_out.data_ptr(),
                                                         upd.sums5.data_ptr(), upd.out6.data_ptr(), 0, 0, 0, st), "loss"), a.iters)
    res["backward_us"] = timed(lambda: check(L.cda_mlp_backward(p.wb.data_ptr(), upd.d_out.data_ptr(), upd.h1p.data_ptr(), upd.h2p.data_ptr(), R, upd.dz1p.data_ptr(), upd.dz2p.data_ptr(),
                                                               upd.doutp.data_ptr(), upd.bias_slab.data_ptr(), st), "bwd"), a.iters)
    res["wgrad_us"] = timed(lambda: check(L.cda_mlp_wgrad(upd.x_pk.data_ptr(), upd.h1p.data_ptr(), upd.h2p.data_ptr(), upd.dz1p.data_ptr(), upd.dz2p.data_ptr(), upd.doutp.data_ptr(), R, chunks,
                                                         upd.slab.data_ptr(), st), "wgrad"), a.iters)
    res["adam_us"] = timed(lambda: check(L.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), upd.slab.data_ptr(), chunks,
                                                       upd.bias_slab.data_ptr(), tiles, upd.sums5.data_ptr(), R * A, 0.5, 0.01, 0.0, upd.out6.data_ptr(), 0.0, 0.9, 0.999, 1e-8, 0.5, upd.grad.data_ptr(), upd.norm2.data_ptr(), st), "adam"), a.iters)
    rec = torch.zeros(R, A, 8, device=dev)
    rec[..., 0] = acts[0].view(R, A).view(torch.float32); rec[..., 1] = acts[1].view(R, A).view(torch.float32); rec[..., 2] = acts[2].view(R, A).view(torch.float32)
    rec[..., 3:5] = acts[3].view(R, A, 2); rec[..., 5] = lp_old.view(R, A); rec[..., 6] = adv.view(R, A); rec[..., 7] = ret.view(R, A)
    res["forward_loss_backward_fused_us"] = timed(lambda: check(L.cda_mlp_forward_backward(
        p.wb.data_ptr(), p.theta.data_ptr(), obs.data_ptr(), upd.perm.data_ptr(), R, R, rec.data_ptr(), None, 0, A, 0.2, 0.5, 0.01, None, upd.x_pk_mb.data_ptr(), upd.h1p.data_ptr(),
        upd.h2p.data_ptr(), upd.dz1p.data_ptr(), upd.dz2p.data_ptr(), upd.doutp.data_ptr(), upd.bias_slab.data_ptr(), upd.sums5.data_ptr(), upd.out6.data_ptr(), 1, 0, None, None, st), "fb"), a.iters)
    res["minibatch_step_fused_us"] = timed(lambda: upd.minibatch_step(0, R, None, None, None, None, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5, records=(rec, None, 0), obs_rows=obs), a.iters)
    res["minibatch_step_us"] = timed(lambda: upd.minibatch_step(0, R, acts, lp_old, adv, ret, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5), a.iters)
    # useful work per row (MACs): layer 1 168 x 512, layer 2 two 256 x 256 blocks, heads 24 x 256 + 1 x 256
    fwd = 168 * 512 + 2 * 256 * 256 + 25 * 256
    bwd = 25 * 256 + 2 * 256 * 256               # no input gradient for layer 1
    wg = fwd
    res["useful_tflops"] = {"forward": 2 * fwd * R / res["forward_train_us"] * 1e-6, "backward": 2 * bwd * R / res["backward_us"] * 1e-6, "wgrad": 2 * wg * R / res["wgrad_us"] * 1e-6,
                            "minibatch_step": 2 * (fwd + bwd + wg) * R / res["minibatch_step_us"] * 1e-6,
                            "forward_loss_backward_fused": 2 * (fwd + bwd) * R / res["forward_loss_backward_fused_us"] * 1e-6,
                            "minibatch_step_fused": 2 * (fwd + bwd + wg) * R / res["minibatch_step_fused_us"] * 1e-6}
    # the rollout's policy step on one chain and on the whole batch
    for n in (1024, 4096):
        o = (torch.randn(n, 168, generator=g)).to(dev)
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        outs = p.policy_step(o, A, 1, counter, 0)
        res[f"policy_step_{n}_us"] = timed(lambda: p.policy_step(o, A, 1, counter, 0, outs=outs), a.iters)
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
