"""GPU: the rollout's policy evaluated INSIDE the step kernel (include/cda.h cda_policy_step_range, csrc/cda_kernels.inc k_policy_step) against the two launches
it replaces - cda_mlp_policy_step (k_mlp_fwd<SAMPLE>) followed by cda_step_range_capture (k_step) - bit for bit: sampled actions, log-probabilities, values,
sample records, the step's observations / rewards / flags, the auto reset and the episode-end capture, markets on the general build (HBM tier of the book)
included.  What the reference does between two env.step calls (RLlib's policy forward + action sampling, train/train.py:453-541) lives in the step itself."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
KEYS = ("category", "size_mean", "size_sigma", "price", "price_offset")


def _bufs(N, A, cap, obs_dim=168):
    e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=DEV)          # noqa: E731
    return {"category": e((N, A), torch.int32), "size_mean": e((N, A), torch.float32), "size_sigma": e((N, A), torch.float32), "price": e((N, A), torch.int32),
            "price_offset": e((N, A), torch.int32), "a_cont": e((N, A, 2), torch.float32), "logp": e((N, A), torch.float32), "value": e((N,), torch.float32),
            "rec": e((N, A, 8), torch.float32), "dist": e((N, 28), torch.float32), "obs": e((N, obs_dim), torch.float32), "reward": e((N, A), torch.float64),
            "term": e((N,), torch.uint8), "trunc": e((N,), torch.uint8), "fin_obs": e((cap, obs_dim), torch.float32), "fin_count": e((1,), torch.int32),
            "fin_index": torch.full((N,), -1, dtype=torch.int32, device=DEV)}


@pytest.mark.parametrize("A,N,max_step,cash,deep,H,sd", [(4, 150, 7, 1000000, False, 4, False), (8, 70, 5, 20000, False, 4, False), (2, 33, 40, 1000000, False, 4, False),
                                                          (4, 40, 4096, 1000000, True, 4, False), (4, 70, 6, 1000000, False, 1, False), (8, 50, 9, 20000, False, 2, False),
                                                          (3, 45, 5, 1000000, False, 8, False), (4, 40, 4096, 1000000, True, 8, False),
                                                          (4, 70, 6, 1000000, False, 3, False), (8, 50, 9, 20000, False, 6, False), (4, 45, 5, 1000000, False, 7, False),
                                                          (4, 150, 7, 1000000, False, 4, True), (8, 50, 9, 20000, False, 2, True)])     # sd: the state-dependent log-std head
def test_policy_inside_the_step_kernel_equals_the_policy_launch_followed_by_the_step_launch(A, N, max_step, cash, deep, H, sd):
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    from gym_continuousdoubleauction_amd._lib import check, lib
    L = lib()
    T, seed = 14, 77
    cfg = {"num_of_agents": A, "init_cash": cash, "max_step": max_step, "is_render": False, "auto_reset": True, "n_hist": H}
    envs = [CDAVecEnv(cfg, n_markets=N, with_info=False) for _ in range(2)]
    OBSD = 42 * H
    assert L.cda_policy_step_supported(envs[1]._h) == 1
    th = mlp.init_theta(OBSD, generator=torch.Generator().manual_seed(5), state_dependent_log_std=sd); th[:mlp.layout(H).OFF_LS] *= 1.5
    p = mlp.FusedPolicy(DEV, theta=th, n_hist=H)
    assert p.state_dependent_log_std == sd
    obs = [e.reset(seed=900).clone() for e in envs]
    if deep:                                                      # four markets far beyond the LDS tile: the general build (HBM tier) steps them, inside both kernels
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from fuzz_cases import prefill_book
        for i in (0, 7, 16, N - 1):
            for e in envs:
                prefill_book(e, i, np.random.default_rng(50 + i), A, 400, 400)
    counter = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    cap = N * (T // max_step + 2)
    b1, b2 = _bufs(N, A, cap, OBSD), _bufs(N, A, cap, OBSD)
    st = torch.cuda.current_stream().cuda_stream
    ends = 0
    for t in range(T):
        for b in (b1, b2):
            b["fin_index"].fill_(-1); b["fin_count"].zero_()
        # the two launches
        p.policy_step(obs[0], A, seed, counter, t, outs={k: b1[k] for k in (*KEYS, "a_cont", "logp", "value")})
        check(L.cda_step_range_capture(envs[0]._h, 0, N, *[b1[k].data_ptr() for k in KEYS], None, b1["obs"].data_ptr(), b1["reward"].data_ptr(), b1["term"].data_ptr(),
                                       b1["trunc"].data_ptr(), None, b1["fin_obs"].data_ptr(), cap, b1["fin_count"].data_ptr(), b1["fin_index"].data_ptr(), st), "step")
        # one launch - as two ranges whose boundary is no multiple of the workgroup's sixteen markets
        cut = N // 2 + 3
        for first, n in ((0, cut), (cut, N - cut)):
            check(L.cda_policy_step_range(envs[1]._h, first, n, p.wb.data_ptr(), p.theta.data_ptr(), obs[1].data_ptr(), seed, counter.data_ptr(), t,
                                          *[b2[k].data_ptr() for k in KEYS], b2["a_cont"].data_ptr(), b2["logp"].data_ptr(), b2["value"].data_ptr(), b2["rec"].data_ptr(),
                                          b2["dist"].data_ptr(), b2["obs"].data_ptr(), b2["reward"].data_ptr(), b2["term"].data_ptr(), b2["trunc"].data_ptr(),
                                          b2["fin_obs"].data_ptr(), cap, b2["fin_count"].data_ptr(), b2["fin_index"].data_ptr(), st), "policy step")
        torch.cuda.synchronize()
        for k in (*KEYS, "a_cont", "logp", "value", "obs", "reward", "term", "trunc"):
            assert torch.equal(b1[k], b2[k]), (k, t)
        # the sample record is the sampled action as the update's loss reads it; the distribution row is the row's normalised log-probabilities | means
        rec = b2["rec"].cpu()
        assert torch.equal(rec[..., 0].contiguous().view(torch.int32), b2["category"].cpu()) and torch.equal(rec[..., 1].contiguous().view(torch.int32), b2["price"].cpu())
        assert torch.equal(rec[..., 2].contiguous().view(torch.int32), b2["price_offset"].cpu()) and torch.equal(rec[..., 3:5], b2["a_cont"].cpu()) and torch.equal(rec[..., 5], b2["logp"].cpu())
        out = p.forward(obs[1]).cpu()
        ls_rows = p.log_std.cpu().double() + out[:, 25:27].double()       # the log-stds every row was sampled with: the free vector + the head's offsets (zero without the head)
        assert (float(out[:, 25:27].abs().max()) > 0.05) == sd and float(out[:, 27:].abs().max()) == 0.0
        want = torch.cat([torch.log_softmax(out[:, :9].double(), -1), torch.log_softmax(out[:, 9:19].double(), -1), torch.log_softmax(out[:, 19:22].double(), -1), out[:, 22:24].double(),
                          ls_rows, torch.zeros(N, 2, dtype=torch.float64)], 1)
        assert (b2["dist"].cpu().double() - want).abs().max() < 1e-5
        # ... and the recorded log-probability is the sampled action's under exactly those log-stds
        a_cont, lp = b2["a_cont"].cpu().double(), b2["logp"].cpu().double()
        z = (a_cont - out[:, None, 22:24].double()) * torch.exp(-ls_rows)[:, None, :]
        lp_want = (-0.5 * z * z - ls_rows[:, None, :] - 0.5 * math.log(2 * math.pi)).sum(-1)
        for (lo, hi), key in zip(((0, 9), (9, 19), (19, 22)), ("category", "price", "price_offset")):
            lp_want = lp_want + want[:, lo:hi].gather(1, b2[key].cpu().long())
        assert (lp - lp_want).abs().max() < 2e-4
        # episode ends: the same markets captured, the same last observations (slots are handed out by an atomic: compared through the index)
        f1, f2 = b1["fin_index"].cpu(), b2["fin_index"].cpu()
        assert torch.equal(f1 >= 0, f2 >= 0) and int(b1["fin_count"]) == int(b2["fin_count"]) == int((f1 >= 0).sum())
        done = (b1["term"] | b1["trunc"]).cpu().bool()
        assert torch.equal(f1 >= 0, done)
        for j in torch.nonzero(done).flatten().tolist():
            assert torch.equal(b1["fin_obs"][f1[j]], b2["fin_obs"][f2[j]]), (t, j)
        ends += int(done.sum())
        obs = [b1["obs"].clone(), b2["obs"].clone()]
    assert ends > 0 or max_step > T
    for e in envs:
        assert (e.flags() == 0).all() and (e.check_invariants() == 0).all()
    for i in (0, N // 2, N - 1):                                  # and the markets themselves: record for record
        assert bytes(envs[0].get_state(i)) == bytes(envs[1].get_state(i)), i
        for side in (0, 1):
            assert np.array_equal(envs[0].get_book(i, side), envs[1].get_book(i, side)), (i, side)
    if deep:
        assert int(envs[1].book_peak().max()) > 256
    for e in envs:
        e.close()


def test_unsupported_envs_say_so_and_the_rollout_chain_falls_back():
    """512-order tiles, history depths the network is not compiled for and more than 8 agents are not built into the one-launch kernel: cda_policy_step_range returns CDA_ERR_UNSUPPORTED
    and the rollout chain launches the two kernels (the rollout still replays: tests/test_hip_hist.py, tools/rollout_soak.py at 12 / 16 agents)."""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd._lib import lib
    L = lib()
    for cfg, want in (({"num_of_agents": 4}, 1), ({"num_of_agents": 4, "n_hist": 8}, 1), ({"num_of_agents": 4, "n_hist": 3}, 1), ({"num_of_agents": 12}, 0), ({"num_of_agents": 4, "n_hist": 5}, 0), ({"num_of_agents": 4, "n_hist": 10}, 0),
                      ({"num_of_agents": 4, "book_capacity": 512}, 0)):
        env = CDAVecEnv(dict(cfg, init_cash=1000000, max_step=64, is_render=False, auto_reset=True), n_markets=32, with_info=False)
        assert L.cda_policy_step_supported(env._h) == want, cfg
        if not want:
            one = C.c_void_p(64)
            assert L.cda_policy_step_range(env._h, 0, 32, one, one, one, 1, one, 0, *([one] * 5), *([one] * 5), *([one] * 4), None, 0, None, None, None) == -4
        env.close()
    # supported is not advised: every workgroup streams the whole network for its sixteen rows - beyond one resident round of workgroups (N > 16 x CUs) the batched
    # policy kernel is the faster one and the rollout chain keeps the two launches
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    small = CDAVecEnv(dict(num_of_agents=4, init_cash=1000000, max_step=64, is_render=False, auto_reset=True), n_markets=16 * cus, with_info=False)
    big = CDAVecEnv(dict(num_of_agents=4, init_cash=1000000, max_step=64, is_render=False, auto_reset=True), n_markets=16 * cus + 16, with_info=False)
    assert (L.cda_policy_step_supported(small._h), L.cda_policy_step_advised(small._h)) == (1, 1)
    assert (L.cda_policy_step_supported(big._h), L.cda_policy_step_advised(big._h)) == (1, 0)
    small.close(); big.close()
