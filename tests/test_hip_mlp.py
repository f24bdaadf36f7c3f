"""GPU: the hand-written bf16 MFMA network (include/cda_mlp.h, csrc/cda_mlp.hip) against its plain PyTorch statement.

Numerics: the kernels multiply bfloat16 operands and accumulate in float32.  `mlp.reference_outputs` / `reference_gradients` restate the
same arithmetic on the CPU in float64 with the SAME roundings (inputs, weights and the activations between layers rounded to bfloat16),
so the comparison is tight: what is left is the accumulation order and an occasional one-ulp flip of a bfloat16 rounding (|h| <= 1:
one ulp = 2^-8 relative).  Tolerances are written at each assert.  The whole chain is also checked against float32 autograd through
`ppo.ActorCritic` - the network as PyTorch states it."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _policy(seed=3, scale=1.0):
    from gym_continuousdoubleauction_amd import mlp
    th = mlp.init_theta(generator=torch.Generator().manual_seed(seed))
    if scale != 1.0:                                    # larger weights: activations leave tanh's linear range
        th[:mlp.OFF_LS] *= scale
    return mlp.FusedPolicy(DEV, theta=th)


def _obs(n, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 168, generator=g) * 1.5
    x[:, ::7] = 0.0                                     # observations hold exact zeros (empty book levels)
    return x


def test_mfma_operand_and_accumulator_conventions():
    """D = A x B through one v_mfma_f32_32x32x16_bf16 with the operand / accumulator maps every kernel of cda_mlp.hip is built on;
    A and B are asymmetric random matrices (a transposed or row/column-swapped map cannot pass)."""
    from gym_continuousdoubleauction_amd._lib import lib, check
    g = torch.Generator().manual_seed(1)
    a = torch.randn(32, 16, generator=g).numpy().astype(np.float32)
    b = torch.randn(16, 32, generator=g).numpy().astype(np.float32)
    d = np.zeros((32, 32), np.float32)
    check(lib().cda_mlp_selftest_mfma(0, a.ctypes.data, b.ctypes.data, d.ctypes.data), "cda_mlp_selftest_mfma")
    ab = torch.from_numpy(a).to(torch.bfloat16).double().numpy()
    bb = torch.from_numpy(b).to(torch.bfloat16).double().numpy()
    want = ab @ bb
    assert np.abs(d - want).max() <= 1e-5 * np.abs(want).max() + 1e-6, np.abs(d - want).max()


@pytest.mark.parametrize("n", [32, 200, 1024 + 7])
@pytest.mark.parametrize("scale", [1.0, 3.0])
def test_forward_equals_the_rounded_reference(n, scale):
    from gym_continuousdoubleauction_amd import mlp
    p = _policy(scale=scale)
    x = _obs(n)
    out = p.forward(x.to(DEV)).cpu().double()
    ref = mlp.reference_outputs(p.theta, x)
    # same roundings, float32 vs float64 accumulation + rare one-ulp flips of an activation's bfloat16 rounding: 3e-3 of the output scale
    tol = 3e-3 * max(1.0, float(ref.abs().max()))
    assert (out[:, :25] - ref[:, :25]).abs().max() <= tol, float((out[:, :25] - ref[:, :25]).abs().max())
    assert (out[:, 25:] == 0).all()
    # and the network as PyTorch states it in float32 (no bfloat16 anywhere): bfloat16 operand precision, 2^-8 per product term
    plain = mlp.reference_outputs(p.theta, x, emulate_bf16=False)
    assert (out[:, :25] - plain[:, :25]).abs().max() <= 4e-2 * max(1.0, float(plain.abs().max()))
    # rows outside the requested range are left alone
    part = torch.full((n, 32), 7.0, device=DEV)
    p.forward(x.to(DEV), first_row=3, n_rows=min(n - 3, 40), out=part)
    assert (part[:3] == 7).all() and (part[3 + min(n - 3, 40):] == 7).all()
    assert torch.equal(part[3:3 + min(n - 3, 40)].cpu().double(), out[3:3 + min(n - 3, 40)])


def _train_forward(p, x, perm=None):
    from gym_continuousdoubleauction_amd import mlp
    from gym_continuousdoubleauction_amd._lib import lib, check
    n = x.shape[0]
    bf = torch.bfloat16
    tile = int(lib().cda_mlp_tile_rows())
    pad = (n + tile - 1) // tile * tile                    # the update's kernels write whole workgroup tiles
    ws = {"x_rm": torch.zeros(n * mlp.KX, dtype=bf, device=DEV), "x_pk": torch.zeros(n * 32 * mlp.XT, dtype=bf, device=DEV),
          "h1p": torch.zeros(pad * 512, dtype=bf, device=DEV), "h2p": torch.zeros(pad * 512, dtype=bf, device=DEV),
          "out": torch.zeros((pad, 32), dtype=torch.float32, device=DEV), "pad": pad}
    xd = x.to(DEV).contiguous()
    permd = None if perm is None else perm.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(lib().cda_mlp_prep_rows(xd.data_ptr(), permd.data_ptr() if permd is not None else None, n, ws["x_rm"].data_ptr(), ws["x_pk"].data_ptr(), st), "prep")
    check(lib().cda_mlp_forward_train(p.wb.data_ptr(), p.theta.data_ptr(), ws["x_rm"].data_ptr(), n, ws["h1p"].data_ptr(), ws["h2p"].data_ptr(), ws["out"].data_ptr(), st), "fwd")
    torch.cuda.synchronize()
    return ws


@pytest.mark.parametrize("n", [32, 160, 512])
def test_training_forward_and_its_packed_images(n):
    from gym_continuousdoubleauction_amd import mlp
    p = _policy(scale=2.0)
    x = _obs(n, seed=9)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(2))
    ws = _train_forward(p, x, perm)
    xs = x[perm]
    ref, xb, h1, h2 = mlp.reference_outputs(p.theta, xs, keep=True)
    # the two images of the (gathered, rounded) observation rows are exact
    x_rm = ws["x_rm"].cpu().float().view(n, mlp.KX)
    assert torch.equal(x_rm[:, :168].double(), xb) and (x_rm[:, 168:] == 0).all()
    x_pk = mlp.unpack_rows(ws["x_pk"], n, 192)
    assert torch.equal(x_pk[:, :168].double(), xb) and (x_pk[:, 168:] == 0).all()
    # activations: bfloat16 values in [-1, 1]; a float32-vs-float64 pre-activation can land on the other side of a rounding boundary: one ulp = 2^-8
    g1, g2 = mlp.unpack_rows(ws["h1p"][:n * 512], n, 512, paired=True).double(), mlp.unpack_rows(ws["h2p"][:n * 512], n, 512, paired=True).double()
    assert (g1 - h1).abs().max() <= 2 ** -8 and (g1 != h1).double().mean() < 0.02
    assert (g2 - h2).abs().max() <= 2 ** -7 and (g2 != h2).double().mean() < 0.05
    out = ws["out"][:n].cpu().double()
    assert (out[:, :25] - ref[:, :25]).abs().max() <= 3e-3 * max(1.0, float(ref.abs().max()))
    # the same rows through the rollout's forward (f32 observations converted in the kernel): the SAME outputs, bit for bit
    out2 = p.forward(xs.to(DEV)).cpu().double()
    assert torch.equal(out2, out)


def _full_backward(p, x, d_out, chunks):
    """forward_train -> backward -> wgrad -> adam(lr = 0): returns workspace + the dense gradient the optimiser saw"""
    from gym_continuousdoubleauction_amd import mlp
    from gym_continuousdoubleauction_amd._lib import lib, check
    n = x.shape[0]
    ws = _train_forward(p, x)
    bf = torch.bfloat16
    tile = int(lib().cda_mlp_tile_rows())
    tiles = (n + tile - 1) // tile
    pad = ws["pad"]
    ws.update(dz1p=torch.zeros(pad * 512, dtype=bf, device=DEV), dz2p=torch.zeros(pad * 512, dtype=bf, device=DEV), doutp=torch.zeros(pad * 32, dtype=bf, device=DEV),
              bias_slab=torch.zeros(tiles * mlp.BSLAB, dtype=torch.float32, device=DEV), slab=torch.zeros(chunks * mlp.SLAB, dtype=torch.float32, device=DEV),
              grad=torch.zeros(mlp.PARAMS, dtype=torch.float32, device=DEV), norm2=torch.zeros(512, dtype=torch.float64, device=DEV))
    dd = d_out.to(DEV).float().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    check(lib().cda_mlp_backward(p.wb.data_ptr(), dd.data_ptr(), ws["h1p"].data_ptr(), ws["h2p"].data_ptr(), n, ws["dz1p"].data_ptr(), ws["dz2p"].data_ptr(),
                                 ws["doutp"].data_ptr(), ws["bias_slab"].data_ptr(), st), "bwd")
    check(lib().cda_mlp_wgrad(ws["x_pk"].data_ptr(), ws["h1p"].data_ptr(), ws["h2p"].data_ptr(), ws["dz1p"].data_ptr(), ws["dz2p"].data_ptr(), ws["doutp"].data_ptr(), n, chunks,
                              ws["slab"].data_ptr(), st), "wgrad")
    theta0 = p.theta.clone()
    check(lib().cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), ws["slab"].data_ptr(), chunks,
                             ws["bias_slab"].data_ptr(), tiles, None, 0, 0.0, 0.0, 0.0, None, 0.0, 0.9, 0.999, 1e-8, 0.5, ws["grad"].data_ptr(), ws["norm2"].data_ptr(), st), "adam")
    torch.cuda.synchronize()
    assert torch.equal(p.theta, theta0)                  # lr = 0
    return ws


@pytest.mark.parametrize("n,chunks", [(32, 1), (160, 3), (768, 8)])
def test_backward_and_weight_gradients_equal_the_rounded_reference(n, chunks):
    from gym_continuousdoubleauction_amd import mlp
    p = _policy(scale=2.0)
    x = _obs(n, seed=11)
    g = torch.Generator().manual_seed(4)
    d_out = torch.zeros(n, 32)
    d_out[:, :25] = torch.randn(n, 25, generator=g) * 1e-3
    ws = _full_backward(p, x, d_out, chunks)
    # the reference is fed the kernel's own activations, so that only the backward arithmetic is compared
    h1, h2 = mlp.unpack_rows(ws["h1p"][:n * 512], n, 512, paired=True).double(), mlp.unpack_rows(ws["h2p"][:n * 512], n, 512, paired=True).double()
    xb = mlp.unpack_rows(ws["x_pk"], n, 192)[:, :168].double()
    gref, dz1, dz2 = mlp.reference_gradients(p.theta, xb, h1, h2, d_out)
    k2, k1 = mlp.unpack_rows(ws["dz2p"][:n * 512], n, 512, paired=True).double(), mlp.unpack_rows(ws["dz1p"][:n * 512], n, 512, paired=True).double()
    # pre-activation gradients: bfloat16 values; one ulp (2^-8 relative) where float32 and float64 round differently
    assert (k2 - dz2).abs().max() <= 2 ** -7 * dz2.abs().max() and (k1 - dz1).abs().max() <= 2 ** -6 * dz1.abs().max()
    assert torch.equal(mlp.unpack_rows(ws["doutp"][:n * 32], n, 32).double(), mlp._r(d_out.double()))
    # weight gradients from the kernel's own dz (isolates the product + reduction): float32 accumulation only
    gk, _, _ = mlp.reference_gradients(p.theta, xb, h1, h2, d_out)
    g_mine = torch.zeros(mlp.PARAMS, dtype=torch.float64)
    g_mine[mlp.OFF_W1:mlp.OFF_B1] = (k1.t() @ xb).reshape(-1); g_mine[mlp.OFF_B1:mlp.OFF_W2] = k1.sum(0)
    g_mine[mlp.OFF_W2:mlp.OFF_B2] = torch.stack([k2[:, :256].t() @ h1[:, :256], k2[:, 256:].t() @ h1[:, 256:]]).reshape(-1); g_mine[mlp.OFF_B2:mlp.OFF_WO] = k2.sum(0)
    g_mine[mlp.OFF_WO:mlp.OFF_LS] = gk[mlp.OFF_WO:mlp.OFF_LS]
    grad = ws["grad"].cpu().double()
    for lo, hi, name in ((mlp.OFF_W1, mlp.OFF_B1, "W1"), (mlp.OFF_B1, mlp.OFF_W2, "b1"), (mlp.OFF_W2, mlp.OFF_B2, "W2"), (mlp.OFF_B2, mlp.OFF_WO, "b2"),
                         (mlp.OFF_WO, mlp.OFF_BO, "Wo"), (mlp.OFF_BO, mlp.OFF_LS, "bo")):
        err = (grad[lo:hi] - g_mine[lo:hi]).abs().max()
        assert err <= 1e-4 * g_mine[lo:hi].abs().max() + 1e-12, (name, float(err), float(g_mine[lo:hi].abs().max()))
        err = (grad[lo:hi] - gref[lo:hi]).abs().max()
        assert err <= 2e-2 * gref[lo:hi].abs().max(), (name, float(err))
    assert (grad[mlp.OFF_LS:] == 0).all()                # no loss sums were handed over
    wo = grad[mlp.OFF_WO:mlp.OFF_BO].view(32, 256)
    assert (wo[25:] == 0).all() and (grad[mlp.OFF_BO + 25:mlp.OFF_LS] == 0).all()
    n2 = ws["norm2"].cpu()
    assert abs(float(n2[2]) - float((grad ** 2).sum())) <= 1e-6 * float((grad ** 2).sum())


def test_clip_and_adam_equal_torch():
    from gym_continuousdoubleauction_amd import mlp
    from gym_continuousdoubleauction_amd._lib import lib, check
    p = _policy()
    n, chunks = 256, 2
    x = _obs(n, seed=21)
    d_out = torch.zeros(n, 32); d_out[:, :25] = torch.randn(n, 25, generator=torch.Generator().manual_seed(8)) * 3e-3
    ws = _full_backward(p, x, d_out, chunks)
    grad = ws["grad"].clone()
    p.adam_m.zero_(); p.adam_v.zero_(); p.adam_step.zero_()      # (the lr = 0 step above moved the moments)
    th = torch.nn.Parameter(p.theta.clone())
    opt = torch.optim.Adam([th], lr=5e-5)
    tile = int(lib().cda_mlp_tile_rows()); tiles = (n + tile - 1) // tile
    st = torch.cuda.current_stream().cuda_stream
    scale = 1.0
    for step, f in enumerate((1.0, 40.0, 0.003)):         # the partial sums scaled between steps: clipped and unclipped steps, moments that disagree with the gradient
        ws["slab"].mul_(f); ws["bias_slab"].mul_(f); scale *= f
        th.grad = grad * scale
        torch.nn.utils.clip_grad_norm_([th], 0.5)
        opt.step()
        check(lib().cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), ws["slab"].data_ptr(), chunks,
                                 ws["bias_slab"].data_ptr(), tiles, None, 0, 0.0, 0.0, 0.0, None, 5e-5, 0.9, 0.999, 1e-8, 0.5, ws["grad"].data_ptr(), ws["norm2"].data_ptr(), st), "adam")
        torch.cuda.synchronize()
        # float32 update of magnitude ~lr = 5e-5: agreement to 1e-3 of a step
        assert (p.theta - th.detach()).abs().max() <= 5e-8, (step, float((p.theta - th.detach()).abs().max()))
    assert float(p.adam_step.item()) == 3.0
    # the operand blob follows theta
    fresh = mlp.FusedPolicy(DEV, theta=p.theta.cpu())
    assert torch.equal(fresh.wb, p.wb)


def test_whole_gradient_equals_float32_autograd_through_the_pytorch_network():
    """forward -> cda_ppo_loss32 -> backward -> wgrad -> reduce, against loss.backward() through ppo.ActorCritic in float32 on the same minibatch."""
    from gym_continuousdoubleauction_amd import mlp, ppo
    from gym_continuousdoubleauction_amd._lib import lib, check
    p = _policy(seed=13)
    R, A = 512, 4
    x = _obs(R, seed=17) * 0.5
    g = torch.Generator().manual_seed(6)
    B = R * A
    a_cat, a_price, a_off = torch.randint(0, 9, (B,), generator=g), torch.randint(0, 10, (B,), generator=g), torch.randint(0, 3, (B,), generator=g)
    a_cont = torch.randn(B, 2, generator=g)
    adv, ret, lp_old = torch.randn(B, generator=g), torch.randn(B, generator=g), torch.randn(B, generator=g) * 0.1 - 7.0
    upd = mlp.FusedUpdate(p, R, R, A, chunks=4)
    upd.perm.copy_(torch.arange(R))
    xd = x.to(DEV)
    check(lib().cda_mlp_prep_rows(xd.data_ptr(), None, R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")
    acts = (a_cat.int().to(DEV), a_price.int().to(DEV), a_off.int().to(DEV), a_cont.to(DEV))
    theta0 = p.theta.clone()
    lpd0, advd0, retd0 = lp_old.to(DEV), adv.to(DEV), ret.to(DEV)
    upd.minibatch_step(0, R, acts, lpd0, advd0, retd0, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5)
    torch.cuda.synchronize()
    assert torch.equal(p.theta, theta0)
    grad = upd.grad.cpu().double()
    # PyTorch statement
    m = mlp.actor_critic_from_theta(p.theta).float()
    logp, ent, v = m.evaluate(x, (a_cat, a_price, a_off, a_cont), agents_per_row=A)
    ratio = (logp - lp_old).exp()
    pg = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean()
    loss = pg + 0.5 * (v - ret).pow(2).mean() - 0.01 * ent.mean()
    loss.backward()
    gm = torch.zeros(mlp.PARAMS, dtype=torch.float64)
    H = 256
    gm[mlp.OFF_W1:mlp.OFF_B1] = m.l1.weight.grad.double().reshape(-1); gm[mlp.OFF_B1:mlp.OFF_W2] = m.l1.bias.grad.double()
    w2g = m.l2.weight.grad.double()
    gm[mlp.OFF_W2:mlp.OFF_B2] = torch.stack([w2g[:H, :H], w2g[H:, H:]]).reshape(-1); gm[mlp.OFF_B2:mlp.OFF_WO] = m.l2.bias.grad.double()
    wog = m.out.weight.grad.double(); blk = torch.zeros(32, H, dtype=torch.float64); blk[:24] = wog[:24, :H]; blk[24] = wog[24, H:]
    gm[mlp.OFF_WO:mlp.OFF_BO] = blk.reshape(-1)
    bog = m.out.bias.grad.double().clone(); bog[25:] = 0
    gm[mlp.OFF_BO:mlp.OFF_LS] = bog; gm[mlp.OFF_LS:] = m.log_std.grad.double()
    cos = float((grad * gm).sum() / (grad.norm() * gm.norm()))
    # bfloat16 operands against float32: direction within 1e-3, every block's magnitude within 3 %
    assert cos > 0.999, cos
    for lo, hi, name in ((mlp.OFF_W1, mlp.OFF_B1, "W1"), (mlp.OFF_B1, mlp.OFF_W2, "b1"), (mlp.OFF_W2, mlp.OFF_B2, "W2"), (mlp.OFF_B2, mlp.OFF_WO, "b2"),
                         (mlp.OFF_WO, mlp.OFF_BO, "Wo"), (mlp.OFF_BO, mlp.OFF_LS, "bo"), (mlp.OFF_LS, mlp.PARAMS, "log_std")):
        a, b = grad[lo:hi], gm[lo:hi]
        assert (a - b).norm() <= 3e-2 * b.norm() + 1e-9, (name, float((a - b).norm() / b.norm()))
    # loss statistics
    out6 = upd.out6.cpu()
    assert abs(float(out6[3]) - float(loss)) <= 2e-2 * abs(float(loss)) + 1e-3
    # the int32 loss kernel and cda_ppo_loss (int64 actions) agree bit for bit on the same outputs
    d64 = torch.zeros_like(upd.d_out); sums = torch.zeros(5, dtype=torch.float64, device=DEV); o6 = torch.zeros(6, device=DEV)
    c64, p64, f64, lpd, advd, retd = a_cat.to(DEV), a_price.to(DEV), a_off.to(DEV), lp_old.to(DEV), adv.to(DEV), ret.to(DEV)       # (kept alive: raw pointers below)
    check(lib().cda_ppo_loss(upd.out.data_ptr(), None, p.theta.data_ptr() + mlp.OFF_LS * 4, c64.data_ptr(), p64.data_ptr(), f64.data_ptr(),
                             acts[3].data_ptr(), lpd.data_ptr(), advd.data_ptr(), retd.data_ptr(), upd.perm.data_ptr(), R, A, 32, 0.2, 0.5, 0.01,
                             d64.data_ptr(), None, sums.data_ptr(), o6.data_ptr(), torch.cuda.current_stream().cuda_stream), "cda_ppo_loss")
    torch.cuda.synchronize()
    # (the same formulas; cda_mlp.hip is compiled with FMA contraction, cda_ppo.hip without: last-bit differences)
    assert torch.allclose(d64, upd.d_out, rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("n", [32, 1000, 262144, 300001])
def test_keyed_permutation_is_a_permutation(n):
    from gym_continuousdoubleauction_amd._lib import lib, check
    perm = torch.zeros(n, dtype=torch.int64, device=DEV)
    seen = []
    for key in (1, 0xDEADBEEFCAFEF00D, 7 << 40):
        check(lib().cda_mlp_permutation(key, n, perm.data_ptr(), torch.cuda.current_stream().cuda_stream), "cda_mlp_permutation")
        p = perm.cpu()
        assert torch.equal(torch.sort(p).values, torch.arange(n))                    # a bijection on 0 .. n-1
        if n >= 1000:                                                                 # ... that shuffles: neighbours land far apart, few fixed points
            assert float((p[1:] - p[:-1]).abs().float().mean()) > 0.25 * n and int((p == torch.arange(n)).sum()) < 10
        seen.append(p)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_sub_batched_minibatch_step_gives_the_same_gradient():
    """FusedUpdate(sub_batches=2): the same optimiser step from two half-size passes (loss sums and partial gradients accumulate)."""
    from gym_continuousdoubleauction_amd import mlp
    from gym_continuousdoubleauction_amd._lib import lib, check
    R, A = 512, 4
    x = _obs(R, seed=3) * 0.5
    g = torch.Generator().manual_seed(12)
    B = R * A
    acts = (torch.randint(0, 9, (B,), generator=g).int().to(DEV), torch.randint(0, 10, (B,), generator=g).int().to(DEV), torch.randint(0, 3, (B,), generator=g).int().to(DEV),
            torch.randn(B, 2, generator=g).to(DEV))
    lp, adv, ret = (torch.randn(B, generator=g) * 0.1 - 7).to(DEV), torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    grads, outs = [], []
    for sub in (1, 2):
        p = _policy(seed=13)
        upd = mlp.FusedUpdate(p, R, R, A, chunks=2, sub_batches=sub)
        upd.perm.copy_(torch.randperm(R, generator=torch.Generator().manual_seed(1)))
        xd = x.to(DEV)
        check(lib().cda_mlp_prep_rows(xd.data_ptr(), upd.perm.data_ptr(), R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")
        upd.minibatch_step(0, R, acts, lp, adv, ret, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5)
        torch.cuda.synchronize()
        grads.append(upd.grad.cpu().double()); outs.append(upd.out6.cpu().double())
    assert (grads[0] - grads[1]).abs().max() <= 1e-5 * grads[0].abs().max() and (outs[0] - outs[1]).abs().max() <= 1e-5 * outs[0].abs().max()


def test_policy_step_samples_what_it_reports():
    from gym_continuousdoubleauction_amd import mlp
    p = _policy(seed=23, scale=2.0)
    N, A = 300, 4
    x = _obs(N, seed=31).to(DEV)
    counter = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    o = p.policy_step(x, A, seed=77, counter=counter, draw=3)
    torch.cuda.synchronize()
    out = p.forward(x).cpu()
    cat, price, off = o["category"].cpu().long(), o["price"].cpu().long(), o["price_offset"].cpu().long()
    assert cat.min() >= 0 and cat.max() <= 8 and price.min() >= 0 and price.max() <= 9 and off.min() >= 0 and off.max() <= 2
    mean, sigma, cont = o["size_mean"].cpu(), o["size_sigma"].cpu(), o["a_cont"].cpu()
    assert mean.abs().max() <= 1 and sigma.min() >= 0 and sigma.max() <= 1
    assert torch.allclose(mean, torch.tanh(cont[..., 0]), atol=1e-6) and torch.allclose(sigma, torch.sigmoid(cont[..., 1]), atol=1e-6)
    assert torch.equal(o["value"].cpu(), out[:, 24])
    # the recorded log-probability is the log-probability of the recorded action under the kernel's own outputs
    lg = out[:, :24].unsqueeze(1).expand(N, A, 24)
    ls = p.theta[mlp.OFF_LS:].cpu()
    lp = (torch.log_softmax(lg[..., :9], -1).gather(-1, cat.unsqueeze(-1)).squeeze(-1) + torch.log_softmax(lg[..., 9:19], -1).gather(-1, price.unsqueeze(-1)).squeeze(-1)
          + torch.log_softmax(lg[..., 19:22], -1).gather(-1, off.unsqueeze(-1)).squeeze(-1))
    z = (cont - lg[..., 22:24]) * torch.exp(-ls)
    lp = lp + (-0.5 * z * z - ls - 0.5 * math.log(2 * math.pi)).sum(-1)
    assert (lp - o["logp"].cpu()).abs().max() <= 2e-4
    # same key -> same draw, other draw index / counter -> other actions; a sub-range leaves the rest alone
    o2 = p.policy_step(x, A, seed=77, counter=counter, draw=3)
    assert all(torch.equal(o[k], o2[k]) for k in o)
    o3 = p.policy_step(x, A, seed=77, counter=counter, draw=4)
    assert not torch.equal(o3["category"], o["category"])
    part = {k: torch.full_like(v, 5) for k, v in o.items()}
    p.policy_step(x, A, seed=77, counter=counter, draw=3, first_market=64, n_markets=100, outs=part)
    torch.cuda.synchronize()
    for k in o:
        assert torch.equal(part[k][64:164], o[k][64:164]) and (part[k][:64] == 5).all() and (part[k][164:] == 5).all(), k
    # the sampler follows the distribution: category frequencies over many draws against the mean softmax (4 sigma of a binomial)
    big = _obs(4096, seed=41).to(DEV)
    ob = p.policy_step(big, A, seed=1, counter=counter, draw=0)
    probs = torch.softmax(p.forward(big)[:, :9].double(), -1).mean(0).cpu()
    freq = torch.bincount(ob["category"].cpu().long().view(-1), minlength=9).double() / (4096 * A)
    assert ((freq - probs).abs() <= 4 * torch.sqrt(probs * (1 - probs) / (4096 * A)) + 1e-3).all(), (freq, probs)
    nz = ((ob["a_cont"].cpu() - p.forward(big)[:, 22:24].cpu().unsqueeze(1)) * torch.exp(-ls)).view(-1)
    assert abs(float(nz.mean())) < 0.02 and abs(float(nz.std()) - 1.0) < 0.02


@pytest.mark.parametrize("groups,use_graphs", [(1, False), (4, True)])
def test_rollout_chains_record_what_the_oracle_replays(groups, use_graphs):
    """A whole rollout as independent chains: the recorded actions replayed through the CPU oracle give the recorded observations and
    rewards bit for bit, and every step's record is what a single policy_step on that step's observation produces."""
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    import oracle_lib as O
    N, A, T = 192, 4, 12
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    p = _policy(seed=29)
    env.reset(seed=500)
    roll = mlp.RolloutChains(env, p, T, groups=groups, seed=99, use_graphs=use_graphs)
    for rnd in range(2):                                  # the second rollout continues where the first stopped (graph replay)
        buf = roll.run()
        torch.cuda.synchronize()
        b = {k: v.cpu() for k, v in buf.items()}
        if rnd == 0:
            ora = O.OracleEnv({k: v for k, v in cfg.items() if k != "auto_reset"}, N)
            o0 = ora.reset(seeds=(500 + np.arange(N)).astype(np.uint64))
            assert np.array_equal(b["obs"][0].numpy().view(np.uint32), o0.view(np.uint32))
        for t in range(T):
            oo, orw, ot, otr, _ = ora.step(b["category"][t].numpy(), b["size_mean"][t].numpy(), b["size_sigma"][t].numpy(), b["price"][t].numpy(), b["price_offset"][t].numpy())
            assert np.array_equal(b["reward"][t].numpy().view(np.uint64), orw.view(np.uint64)), (rnd, t)
            assert np.array_equal(b["obs"][t + 1].numpy().view(np.uint32), oo.view(np.uint32)), (rnd, t)
            assert not b["terminated"][t].any() and not b["truncated"][t].any()
        rec = b["record"]                                    # the sample records hold the same words as the per-sample arrays
        for w, k in enumerate(("category", "price", "price_offset")):
            assert torch.equal(rec[..., w].contiguous().view(torch.int32), b[k])
        assert torch.equal(rec[..., 3:5], b["a_cont"]) and torch.equal(rec[..., 5], b["logp"])
        cnt = roll.counter.clone()
        for t in (0, T // 2, T - 1):
            o = p.policy_step(buf["obs"][t], A, seed=99, counter=cnt, draw=t)
            torch.cuda.synchronize()
            for k in ("category", "size_mean", "size_sigma", "price", "price_offset", "a_cont", "logp"):
                assert torch.equal(o[k].cpu(), b[k][t]), (k, t)
            assert torch.equal(o["value"].cpu(), b["value"][t])
        assert torch.equal(p.forward(buf["obs"][T])[:, 24].cpu(), b["value"][T])
    assert (env.flags() == 0).all() and (env.check_invariants() == 0).all()
    env.close(); ora.close()


@pytest.mark.parametrize("T", [40, 13])                     # (the kernel walks eight steps at a time: a horizon that is not a multiple of 8 as well)
def test_gae_into_records_and_the_loss_on_records_equal_the_array_path(T):
    """cda_gae_records = ppo.gae on the rollout's buffers (episode ends included), its sums = the normalisation ppo_update applies; the loss
    reading records (normalising on the fly) = the loss reading the seven arrays with torch-normalised advantages."""
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp, ppo
    from gym_continuousdoubleauction_amd._lib import lib, check
    N, A = 96, 4
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 8, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    p = _policy(seed=31)
    env.reset(seed=7)
    roll = mlp.RolloutChains(env, p, T, groups=2, seed=5)
    buf = roll.run()
    rec, stats, count = roll.gae(gamma=0.97, lam=0.9, reward_scale=1e-3)
    torch.cuda.synchronize()
    assert count == T * N * A and bool((buf["terminated"] | buf["truncated"]).any())
    val = buf["value"][:T].unsqueeze(-1).expand(T, N, A).reshape(T, N * A)
    last_val = buf["value"][T].unsqueeze(-1).expand(N, A).reshape(N * A)
    rew = (buf["reward"].float() * 1e-3).view(T, N * A)
    dn = (buf["terminated"] | buf["truncated"]).unsqueeze(-1).expand(T, N, A).reshape(T, N * A).float()
    adv, ret = ppo.gae(rew, val, last_val, dn, gamma=0.97, lam=0.9)
    r4 = buf["record"]
    assert torch.allclose(r4[..., 6].reshape(T, N * A), adv, rtol=1e-5, atol=1e-6) and torch.allclose(r4[..., 7].reshape(T, N * A), ret, rtol=1e-5, atol=1e-6)
    a64 = r4[..., 6].double()
    assert abs(float(stats[0]) - float(a64.sum())) <= 1e-6 * float(a64.abs().sum()) and abs(float(stats[1]) - float((a64 * a64).sum())) <= 1e-9 * float((a64 * a64).sum())
    # the two loss kernels on the same rows
    R = T * N
    upd = mlp.FusedUpdate(p, R, R, A, chunks=2)
    upd.perm.copy_(torch.randperm(R, generator=torch.Generator().manual_seed(2)))
    check(lib().cda_mlp_prep_rows(buf["obs"][:T].view(R, -1).data_ptr(), upd.perm.data_ptr(), R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")
    adv_n = ((r4[..., 6] - r4[..., 6].mean()) / (r4[..., 6].std() + 1e-8)).reshape(-1).contiguous()
    acts = (buf["category"].view(-1), buf["price"].view(-1), buf["price_offset"].view(-1), buf["a_cont"].view(-1, 2))
    upd.minibatch_step(0, R, acts, buf["logp"].view(-1), adv_n, r4[..., 7].reshape(-1).contiguous(), 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5)
    torch.cuda.synchronize()
    g0, o0, d0 = upd.grad.clone(), upd.out6.clone(), upd.d_out.clone()
    upd.minibatch_step(0, R, None, None, None, None, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5, records=(rec, stats, count))
    torch.cuda.synchronize()
    assert torch.allclose(upd.d_out, d0, rtol=2e-4, atol=1e-9) and torch.allclose(upd.out6, o0, rtol=1e-4, atol=1e-7)
    assert (upd.grad - g0).abs().max() <= 2e-4 * g0.abs().max()
    assert abs(float(o0[0])) < 1e-3 + 1e-3                   # the rollout's own policy: ratio 1, normalised advantages -> policy loss ~ 0
    env.close()


@pytest.mark.parametrize("N,T,A", [(96, 40, 4), (100, 40, 4),    # 3840 rows = 60 whole tiles; 4000 rows: the last tile is half empty
                                   (64, 8, 8), (64, 8, 2), (64, 8, 1), (32, 9, 16), (40, 8, 5)])   # other agent counts: the quad deals agents round four lanes
def test_fused_forward_loss_backward_equals_the_separate_kernels(N, T, A):
    """cda_mlp_forward_backward (one launch: gather, forward, loss, back-propagation) against prep_rows + forward_train + loss_records + backward on the
    same minibatch: outputs bit for bit, gradients to float32 rounding of the loss arithmetic."""
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    from gym_continuousdoubleauction_amd._lib import lib, check
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 16, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    p = _policy(seed=37)
    env.reset(seed=11)
    roll = mlp.RolloutChains(env, p, T, groups=2, seed=6)
    buf = roll.run()
    records = roll.gae(gamma=0.99, lam=0.95, reward_scale=1e-3)
    R = T * N
    obs = buf["obs"][:T].view(R, -1)
    perm = torch.randperm(R, generator=torch.Generator().manual_seed(3))
    res = []
    for fused in (False, True):
        upd = mlp.FusedUpdate(p, R, R, A, chunks=3, fused=fused)
        upd.perm.copy_(perm)
        if not fused:
            check(lib().cda_mlp_prep_rows(obs.data_ptr(), upd.perm.data_ptr(), R, upd.x_rm.data_ptr(), upd.x_pk.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")
        upd.minibatch_step(0, R, None, None, None, None, 0.2, 0.5, 0.01, 0.0, (0.9, 0.999), 1e-8, 0.5, records=records, obs_rows=obs if fused else None, debug_outputs=True)
        torch.cuda.synchronize()
        xpk = (upd.x_pk_mb if fused else upd.x_pk)[:R * 192].clone()
        res.append(dict(out=upd.out[:R].clone(), d_out=upd.d_out[:R].clone(), grad=upd.grad.clone(), out6=upd.out6.clone(), xpk=xpk,
                        h1=upd.h1p[:R * 512].clone(), h2=upd.h2p[:R * 512].clone(), dz2=upd.dz2p[:R * 512].float().clone(), dz1=upd.dz1p[:R * 512].float().clone()))
    a, b = res
    assert torch.equal(a["xpk"].view(torch.int16), b["xpk"].view(torch.int16))
    assert torch.equal(a["h1"].view(torch.int16), b["h1"].view(torch.int16)) and torch.equal(a["h2"].view(torch.int16), b["h2"].view(torch.int16))
    assert torch.equal(a["out"][:, :25], b["out"][:, :25])
    # (the two loss kernels order a row's float32 sums differently: entries that are differences of much larger terms agree to the terms' rounding)
    assert torch.allclose(a["d_out"], b["d_out"], rtol=2e-4, atol=2e-5 * float(a["d_out"].abs().max()))
    for k in ("dz2", "dz1"):
        assert (a[k] - b[k]).abs().max() <= 2e-2 * a[k].abs().max()          # (a float32 ulp in d_out now and then moves a bf16 rounding)
    assert torch.allclose(a["out6"], b["out6"], rtol=1e-4, atol=1e-7)
    assert (a["grad"] - b["grad"]).abs().max() <= 1e-3 * a["grad"].abs().max()
    env.close()


def test_fused_training_loop_runs_end_to_end():
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 48, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=256, with_info=False)
    logs, keep = [], {}
    pol, hist = ppo.train_fused(env, iters=3, horizon=32, log=logs.append, minibatch=256 * 32 * 4 // 2, keep=keep)
    assert len(hist) == 3 and all(math.isfinite(h[k]) for h in hist for k in ("pg_loss", "v_loss", "entropy", "mean_reward"))
    assert float(pol.adam_step.item()) == 3 * 4 * 2       # iterations x epochs x minibatches
    assert torch.isfinite(pol.theta).all()
    b = keep["buffers"]
    assert bool((b["terminated"] | b["truncated"]).any())  # max_step 48 < 3 x 32: episodes ended and were reset on the device
    assert (env.flags() == 0).all() and (env.check_invariants() == 0).all()
    _, bad = env.nav_conservation()
    assert not bad.any()
    # the first minibatch step of an update recomputes the rollout's own log-probabilities: ratio == 1 up to float32 rounding
    env.close()


@pytest.mark.parametrize("hidden,sd", [((128, 64), False), ((64, 256), True), ((200, 100), False)])
def test_narrow_networks_train_inside_the_wide_kernels(hidden, sd):
    """The reference's `fcnet_hiddens` is configuration (config/train_config.json:49); the kernels are compiled for 256-wide layers.  A narrower network is the same
    parameter vector with the units beyond (h1, h2) dead (mlp.init_theta(hidden=...)): their tanh(0) is an exact zero on the device, so every gradient entry that touches a
    dead unit is an exact zero and Adam never moves it - after optimiser steps on the fused kernels the dead entries are still bit-zero, the live ones moved, and the
    network's outputs are those of the narrow network (the float64 statement of the live blocks alone)."""
    import math
    from gym_continuousdoubleauction_amd import mlp
    g = torch.Generator().manual_seed(21)
    th0 = mlp.init_theta(generator=torch.Generator().manual_seed(4), hidden=hidden, state_dependent_log_std=sd)
    assert mlp.hidden_widths(th0) == hidden
    p = mlp.FusedPolicy(DEV, theta=th0)
    R, A = 512, 4
    x = (torch.randn(R, mlp.OBS, generator=g) * 0.5)
    rec = torch.zeros(R, A, 8)
    rec[..., 0] = torch.randint(0, 9, (R, A), generator=g).int().view(torch.float32)
    rec[..., 1] = torch.randint(0, 10, (R, A), generator=g).int().view(torch.float32)
    rec[..., 2] = torch.randint(0, 3, (R, A), generator=g).int().view(torch.float32)
    rec[..., 3:5] = torch.randn(R, A, 2, generator=g)
    rec[..., 5] = torch.randn(R, A, generator=g) * 0.1 - 7.0
    rec[..., 6] = torch.randn(R, A, generator=g)
    rec[..., 7] = torch.randn(R, A, generator=g)
    upd = mlp.FusedUpdate(p, R, R, A)
    recd, xd = rec.to(DEV), x.to(DEV)
    for step in range(6):
        upd.perm.copy_(torch.randperm(R, generator=g))
        upd.minibatch_step(0, R, None, None, None, None, 0.3, 1.0, 0.01, 1e-3, (0.9, 0.999), 1e-8, 0.5, records=(recd.data_ptr(), None, 0), obs_rows=xd)
    torch.cuda.synchronize()
    th1 = p.theta.cpu()
    dead = th0 == 0
    dead[mlp.OFF_LS:] = False
    assert int(dead.sum()) > 50000 and bool((th1[dead] == 0).all())              # bit-zero, not small
    assert mlp.hidden_widths(th1) == hidden and mlp.has_log_std_head(th1) == sd
    live = ~dead
    live[mlp.OFF_LS:] = False
    assert float((th1[live] - th0[live]).abs().max()) > 1e-4 and float(((th1 - th0)[live] != 0).float().mean()) > 0.95      # ... while the live network trained
    # the outputs are the narrow network's: layer by layer over the live blocks only (float64, bf16-rounded operands like the kernels)
    h1, h2 = hidden
    t = th1.double()
    rd = lambda v: v.to(torch.bfloat16).to(torch.float64)                          # noqa: E731
    W1, b1 = t[mlp.OFF_W1:mlp.OFF_B1].view(2, 256, mlp.OBS), t[mlp.OFF_B1:mlp.OFF_W2].view(2, 256)
    W2, b2 = t[mlp.OFF_W2:mlp.OFF_B2].view(2, 256, 256), t[mlp.OFF_B2:mlp.OFF_WO].view(2, 256)
    Wo, bo = t[mlp.OFF_WO:mlp.OFF_BO].view(32, 256), t[mlp.OFF_BO:mlp.OFF_LS]
    xb = rd(x.double())
    hid = [rd(torch.tanh(rd(torch.tanh(xb @ rd(W1[k, :h1]).t() + b1[k, :h1])) @ rd(W2[k, :h2, :h1]).t() + b2[k, :h2])) for k in range(2)]
    want = torch.zeros(R, 32, dtype=torch.float64)
    rows = mlp.POLICY_ROWS
    want[:, rows] = hid[0] @ rd(Wo[rows, :h2]).t() + bo[rows]
    want[:, 24] = hid[1] @ rd(Wo[24, :h2]) + bo[24]
    got = p.forward(xd).cpu().double()
    assert float((got - want).abs().max()) <= 3e-3 * max(1.0, float(want.abs().max()))
    assert math.isfinite(float(upd.out6.cpu()[3]))
