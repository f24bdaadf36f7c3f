"""A small PyTorch-ROCm PPO loop on the batched env (BASELINE config #5).

The reference trains through RLlib's PPO (train/train.py:453-541, config/train_config.json `ppo` group:
256x256 tanh MLP, lr 5e-5, 4 epochs, separate value network).  This is NOT a port of that harness - it is
the consumer-side counterpart of the vectorised env: rollouts never leave the GPU (obs/reward tensors come
straight from `CDAVecEnv.step`), one shared policy plays every agent slot (self-play), and the Dict action
is produced by three categorical heads (category 9, price 10, price_offset 3) and two bounded Gaussian heads
(size_mean in [-1,1], size_sigma in [0,1]).

Shared observations: every agent of a market is handed the SAME observation vector (exchg/state_helper.py:76,109), and with one
shared policy the network outputs (logits, value) are therefore the same for all A agents of a (step, market) pair.  By default
(`shared_obs=True`) the network runs once per market-step - on N rows in the rollout, on the T x N unique observations in the
update - and the A samples of a row differ only in their drawn action, advantage and return: the loss of a minibatch is the same
sum over the same samples, its gradient with respect to a row's outputs the sum over the row's samples (csrc/cda_ppo.hip), and a
minibatch holds whole market-steps (rows are shuffled, not single samples).  A quarter of the matrix work at 4 agents, an eighth
at 8, for the same numbers; `shared_obs=False` is the plain one-forward-per-sample loop (the rows are then replicated A times).

    python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 4
"""
import argparse
import json
import math
import time

import torch
import torch.nn as nn

CAT_N, PRICE_N, OFF_N = 9, 10, 3


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T + b whose WEIGHT gradient is computed as a split-K batched product.  dW = g^T x has a tiny output (256 x 168 ...) and
    a huge reduction (the minibatch, 262 144): as ONE GEMM it is a dozen workgroups on a 256-CU chip (hipBLASLt picked MT64x64x256
    without split-K: 229 us per call, a third of the round-2 update, profiles/r02/kernel_stats_ppo.csv).  Cut into SPLIT
    slices of the batch it is SPLIT x a dozen workgroups (torch.bmm), summed in float32 afterwards."""
    SPLIT = 64

    @staticmethod
    def forward(ctx, x, w, b, mask=None):
        """mask (optional, same shape as w): a block structure of w - entries where mask == 0 ARE zero in w and must stay so; their
        gradient is dropped here (what multiplying w by the mask in the forward pass would do in its backward, without the two
        elementwise passes over w per call)."""
        ctx.save_for_backward(x, w, mask) if mask is not None else ctx.save_for_backward(x, w)
        ctx.has_mask = mask is not None
        return torch.nn.functional.linear(x, w.to(x.dtype), b.to(x.dtype))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors[:2]
        g = g.contiguous()
        gx = g @ w.to(g.dtype) if ctx.needs_input_grad[0] else None
        n, s = g.shape[0], _SplitKLinear.SPLIT
        acc = torch.float32 if g.dtype in (torch.bfloat16, torch.float16) else g.dtype       # the slices are summed in float32
        if n % s == 0 and n >= 64 * s:
            gw = torch.bmm(g.view(s, n // s, -1).transpose(1, 2), x.reshape(s, n // s, -1)).sum(0, dtype=acc)
        else:
            gw = (g.t() @ x).to(acc)
        if ctx.has_mask:
            gw = gw * ctx.saved_tensors[2]
        return gx, gw.to(w.dtype), g.sum(0, dtype=acc).to(w.dtype), None


class _Linear(nn.Linear):
    def forward(self, x):
        return _SplitKLinear.apply(x, self.weight, self.bias)


class ActorCritic(nn.Module):
    """Separate policy and value MLPs (256x256 tanh), as in config/train_config.json:49 - stored and multiplied as ONE set of block
    matrices: both read the same observation, so the first layers are one 168 -> 512 product; the second layers are the two
    diagonal blocks of a 512 x 512 matrix; the heads are rows 0..23 (policy, reading the first 256 units) and row 24 (value, reading
    the last 256) of a 32 x 512 matrix.  The off-block entries start at zero and their gradient is masked out (_SplitKLinear), so they stay
    zero and the two halves never mix - mathematically two independent networks.  Why: three well-shaped GEMMs per direction instead of five, two of them with 24 and
    ONE output column - shapes for which the bfloat16 GEMM libraries take a slow path (the 1-column value head alone made a
    backward pass take 12 ms of host time, tools/ppo_probe.py)."""
    N_OUT, N_PAD = CAT_N + PRICE_N + OFF_N + 2, 32

    def __init__(self, obs_dim, hidden=256, state_dependent_log_std=False):
        """state_dependent_log_std: RLlib's default module for Box actions (what the reference's PPO modules are, train/policy/policy_handler.py:69-76): rows 25, 26 of
        the output matrix read the policy half and are per-row log-std OFFSETS on top of the free `log_std` vector (which then stays where it is)."""
        super().__init__()
        self.hidden = hidden
        self.state_dependent_log_std = bool(state_dependent_log_std)
        H = hidden
        self.l1 = _Linear(obs_dim, 2 * H)                         # [policy | value] first layers
        self.l2 = _Linear(2 * H, 2 * H)
        self.out = _Linear(2 * H, self.N_PAD)
        m2 = torch.zeros(2 * H, 2 * H)
        m2[:H, :H] = 1; m2[H:, H:] = 1
        mo = torch.zeros(self.N_PAD, 2 * H)
        mo[:self.N_OUT, :H] = 1; mo[self.N_OUT, H:] = 1
        if self.state_dependent_log_std:
            mo[self.N_OUT + 1:self.N_OUT + 3, :H] = 1
        self.register_buffer("mask2", m2)
        self.register_buffer("mask_out", mo)
        with torch.no_grad():                                      # every block initialised as the nn.Linear(256, .) it stands for
            for w, b, rows in ((self.l2.weight, self.l2.bias, 2 * H), (self.out.weight, self.out.bias, self.N_PAD)):
                bound = 1.0 / math.sqrt(H)
                w.uniform_(-bound, bound); b.uniform_(-bound, bound)
            self.l2.weight.mul_(m2); self.out.weight.mul_(mo)
            self.out.bias[self.N_OUT + (3 if self.state_dependent_log_std else 1):] = 0
        self.log_std = nn.Parameter(torch.full((2,), -0.5), requires_grad=not self.state_dependent_log_std)

    def trunk(self, obs):
        """-> (policy outputs [B, 24], value [B])"""
        o = self.trunk_packed(obs)
        return o[:, :self.N_OUT], o[:, self.N_OUT]

    def trunk_ls(self, obs):
        """-> (policy outputs [B, 24], value [B], log-stds [B, 2]: the free vector, plus the row's offsets with the state-dependent head)"""
        o = self.trunk_packed(obs)
        ls = self.log_std.float().expand(o.shape[0], 2)
        if self.state_dependent_log_std:
            ls = ls + o[:, self.N_OUT + 1:self.N_OUT + 3].float()
        return o[:, :self.N_OUT], o[:, self.N_OUT], ls

    def trunk_packed(self, obs):
        """-> the padded output matrix [B, 32] as the last product leaves it: policy outputs in columns 0..23, the value in column 24
        (what cda_ppo_loss / cda_policy_sample read in place, `out_stride` / `logits_stride` = 32)"""
        h = torch.tanh(self.l1(obs))
        h = torch.tanh(_SplitKLinear.apply(h, self.l2.weight, self.l2.bias, self.mask2))     # (off-block entries: zero at init, zero gradient)
        return _SplitKLinear.apply(h, self.out.weight, self.out.bias, self.mask_out)

    def pi(self, obs):
        return self.trunk(obs)[0]

    def v(self, obs):
        return self.trunk(obs)[1].unsqueeze(-1)

    def _dists(self, o, log_std=None):
        # validate_args=False: the argument checks read a flag back to the host, which neither a captured HIP graph
        # nor an asynchronous rollout can afford
        o = o.float()
        cat = torch.distributions.Categorical(logits=o[:, :CAT_N], validate_args=False)
        price = torch.distributions.Categorical(logits=o[:, CAT_N:CAT_N + PRICE_N], validate_args=False)
        off = torch.distributions.Categorical(logits=o[:, CAT_N + PRICE_N:CAT_N + PRICE_N + OFF_N], validate_args=False)
        mu = o[:, -2:]
        cont = torch.distributions.Normal(mu, (self.log_std if log_std is None else log_std).exp().expand_as(mu), validate_args=False)
        return cat, price, off, cont

    def dists(self, obs):
        o, _, ls = self.trunk_ls(obs)
        return self._dists(o, ls)

    def act(self, obs):
        o, val, ls = self.trunk_ls(obs)
        cat, price, off, cont = self._dists(o, ls)
        # Normal.sample() checks std >= 0 on the host (a sync, illegal inside a captured graph): draw the noise directly
        a_cat, a_price, a_off = cat.sample(), price.sample(), off.sample()
        a_cont = (cont.loc + cont.scale * torch.randn_like(cont.loc)).detach()
        logp = cat.log_prob(a_cat) + price.log_prob(a_price) + off.log_prob(a_off) + cont.log_prob(a_cont).sum(-1)
        return (a_cat, a_price, a_off, a_cont), logp, val.float()

    def act_fused(self, obs, n, a, state, shared=False):
        """act() + to_env_actions() for HIP tensors with ONE sampling launch (cda_policy_sample) behind the network: returns
        (actions, logp, value, env_actions).  `state` = (seed, counter tensor i64[1]) from new_sampler_state().
        shared=False: `obs` holds one row per (market, agent) pair, [n * a, obs_dim].  shared=True: one row per market, [n, obs_dim] -
        the network runs once per market and its outputs serve the market's `a` agents; `value` is then per market, [n]."""
        from ._lib import check, lib
        if self.state_dependent_log_std:
            raise NotImplementedError("cda_policy_sample takes ONE log_std pair: the state-dependent head samples through act() or the fused network kernels (mlp.FusedPolicy)")
        o = self.trunk_packed(obs).float().contiguous()           # [rows, 32]: read in place, the value made contiguous by the sampler
        rows, dev = o.shape[0], o.device
        val = torch.empty(rows, dtype=torch.float32, device=dev)
        per_row = a if shared else 1
        B = rows * per_row
        a_cat, a_price, a_off = (torch.empty(B, dtype=torch.int64, device=dev) for _ in range(3))
        a_cont, logp = torch.empty((B, 2), dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
        e_cat, e_price, e_off = (torch.empty((n, a), dtype=torch.int32, device=dev) for _ in range(3))
        e_mean, e_sigma = (torch.empty((n, a), dtype=torch.float32, device=dev) for _ in range(2))
        seed, counter = state
        check(lib().cda_policy_sample(o.data_ptr(), self.N_PAD, val.data_ptr(), self.log_std.detach().float().contiguous().data_ptr(), rows, per_row,
                                      int(seed) & (2 ** 64 - 1), counter.data_ptr(),
                                      a_cat.data_ptr(), a_price.data_ptr(), a_off.data_ptr(), a_cont.data_ptr(), logp.data_ptr(),
                                      e_cat.data_ptr(), e_mean.data_ptr(), e_sigma.data_ptr(), e_price.data_ptr(), e_off.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "cda_policy_sample")
        return (a_cat, a_price, a_off, a_cont), logp, val, (e_cat, e_mean, e_sigma, e_price, e_off)

    def evaluate(self, obs, actions, agents_per_row=1):
        """log-probability of `actions`, entropy and value for a batch - the three discrete heads through ONE log-softmax pass each
        on the float32 logits, the Gaussian heads in closed form (no distribution objects: a quarter of the elementwise launches).
        agents_per_row > 1: `obs` holds one row per market-step and serves that many consecutive samples of `actions`; the outputs
        are per sample (the row's outputs repeated)."""
        a_cat, a_price, a_off, a_cont = actions
        o, val, log_std = self.trunk_ls(obs)
        o = o.float()
        if agents_per_row > 1:
            o, val, log_std = o.repeat_interleave(agents_per_row, dim=0), val.repeat_interleave(agents_per_row, dim=0), log_std.repeat_interleave(agents_per_row, dim=0)
        logp = ent = 0.0
        for lo, hi, a in ((0, CAT_N, a_cat), (CAT_N, CAT_N + PRICE_N, a_price), (CAT_N + PRICE_N, CAT_N + PRICE_N + OFF_N, a_off)):
            ls = torch.log_softmax(o[:, lo:hi], dim=-1)
            logp = logp + ls.gather(1, a.view(-1, 1)).squeeze(1)
            ent = ent - (ls.exp() * ls).sum(-1)
        mu = o[:, -2:]
        z = (a_cont - mu) * torch.exp(-log_std)
        logp = logp + (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        ent = ent + (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1)
        return logp, ent, val.float()


class _FusedPPOLoss(torch.autograd.Function):
    """loss(minibatch) and its gradient with respect to the network outputs in ONE launch of the HIP kernel behind cda_ppo_loss
    (csrc/cda_ppo.hip) instead of ~100 elementwise / reduction launches; `ActorCritic.evaluate` + the formulas of ppo_update
    are the plain PyTorch statement of the same op (the numerics reference, and the path of non-HIP tensors)."""

    @staticmethod
    def forward(ctx, logits, value, log_std, a_cat, a_price, a_off, a_cont, logp_old, adv, ret, clip, vf_coef, ent_coef, agents_per_row=1, row_index=None):
        from ._lib import check, lib
        rows = logits.shape[0]                                     # a row serves agents_per_row consecutive samples (cda.h cda_ppo_loss)
        if row_index is None:
            assert a_cat.numel() == rows * agents_per_row and adv.numel() == rows * agents_per_row
        else:                                                      # the rows are a shuffled minibatch; the per-sample arrays hold the whole batch
            assert row_index.numel() == rows and row_index.dtype == torch.int64 and row_index.is_contiguous()
        logits, value = logits.contiguous(), value.contiguous()
        d_logits, d_value = torch.empty_like(logits), torch.empty_like(value)
        sums = torch.empty(5, dtype=torch.float64, device=logits.device)
        out = torch.empty(6, dtype=torch.float32, device=logits.device)
        check(lib().cda_ppo_loss(logits.data_ptr(), value.data_ptr(), log_std.detach().float().contiguous().data_ptr(), a_cat.data_ptr(), a_price.data_ptr(),
                                 a_off.data_ptr(), a_cont.contiguous().data_ptr(), logp_old.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                 row_index.data_ptr() if row_index is not None else None, rows, int(agents_per_row), 0,
                                 float(clip), float(vf_coef), float(ent_coef), d_logits.data_ptr(), d_value.data_ptr(), sums.data_ptr(), out.data_ptr(),
                                 torch.cuda.current_stream(logits.device).cuda_stream), "cda_ppo_loss")
        ctx.save_for_backward(d_logits, d_value, out)
        ctx.mark_non_differentiable(out)
        return out[3], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        d_logits, d_value, out = ctx.saved_tensors
        return (d_logits * g_loss, d_value * g_loss, out[4:6] * g_loss) + (None,) * 12


class _FusedPPOLossPacked(torch.autograd.Function):
    """_FusedPPOLoss on the network's padded output matrix [rows, 32] as it stands (ActorCritic.trunk_packed): no slicing of logits
    and value before the kernel, no zero-filled reassembly of their gradients behind it - one tensor in, its gradient out."""

    @staticmethod
    def forward(ctx, outputs, log_std, a_cat, a_price, a_off, a_cont, logp_old, adv, ret, clip, vf_coef, ent_coef, agents_per_row=1, row_index=None):
        from ._lib import check, lib
        outputs = outputs.contiguous()
        rows, stride = outputs.shape
        if row_index is None:
            assert a_cat.numel() == rows * agents_per_row and adv.numel() == rows * agents_per_row
        else:
            assert row_index.numel() == rows and row_index.dtype == torch.int64 and row_index.is_contiguous()
        d_out = torch.empty_like(outputs)
        sums = torch.empty(5, dtype=torch.float64, device=outputs.device)
        out = torch.empty(6, dtype=torch.float32, device=outputs.device)
        check(lib().cda_ppo_loss(outputs.data_ptr(), None, log_std.detach().float().contiguous().data_ptr(), a_cat.data_ptr(), a_price.data_ptr(),
                                 a_off.data_ptr(), a_cont.contiguous().data_ptr(), logp_old.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                 row_index.data_ptr() if row_index is not None else None, rows, int(agents_per_row), int(stride),
                                 float(clip), float(vf_coef), float(ent_coef), d_out.data_ptr(), None, sums.data_ptr(), out.data_ptr(),
                                 torch.cuda.current_stream(outputs.device).cuda_stream), "cda_ppo_loss")
        ctx.save_for_backward(d_out, out)
        ctx.mark_non_differentiable(out)
        return out[3], out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        d_out, out = ctx.saved_tensors
        return (d_out * g_loss, out[4:6] * g_loss) + (None,) * 12


def to_env_actions(actions, n, a):
    """Policy sample -> the env's five [N,A] tensors (raw Gaussian heads squashed into the Box bounds)."""
    a_cat, a_price, a_off, a_cont = actions
    mean = torch.tanh(a_cont[:, 0])
    sigma = torch.sigmoid(a_cont[:, 1])
    return (a_cat.view(n, a).to(torch.int32), mean.view(n, a).float(), sigma.view(n, a).float(),
            a_price.view(n, a).to(torch.int32), a_off.view(n, a).to(torch.int32))


def gae(rew, val, last_val, done, gamma=0.99, lam=0.95, fused=None):
    """rew/val/done: [T, B]; returns advantages and returns [T, B].  HIP tensors take ONE launch (cda_gae, csrc/cda_ppo.hip: a thread
    per column walks its T steps backwards); the loop below is the plain statement of the same recursion (and the path elsewhere)."""
    fused = (rew.is_cuda and rew.dtype == torch.float32) if fused is None else bool(fused)
    if fused:
        from ._lib import check, lib
        T, B = rew.shape
        rew, val, done, last_val = rew.contiguous(), val.float().contiguous(), done.float().contiguous(), last_val.float().contiguous()
        adv, ret = torch.empty_like(rew), torch.empty_like(rew)
        check(lib().cda_gae(rew.data_ptr(), val.data_ptr(), last_val.data_ptr(), done.data_ptr(), T, B, float(gamma), float(lam), adv.data_ptr(), ret.data_ptr(),
                            torch.cuda.current_stream(rew.device).cuda_stream), "cda_gae")
        return adv, ret
    T = rew.shape[0]
    adv = torch.zeros_like(rew)
    nxt, run = last_val, torch.zeros_like(last_val)
    for t in range(T - 1, -1, -1):
        nd = 1.0 - done[t]
        delta = rew[t] + gamma * nxt * nd - val[t]
        run = delta + gamma * lam * nd * run
        adv[t] = run
        nxt = val[t]
    return adv, adv + val


class _GraphedUpdate:
    """The minibatch steps of ppo_update as captured HIP graphs: one graph per minibatch slot (forward, fused loss, backward, gradient
    clipping, Adam step - ~70 launches that cost more host time than device time once the network runs once per market-step), reading
    STATIC buffers: the epoch's shuffled observations and its permutation (the loss kernel finds a row's samples through it, so the
    per-sample tensors are copied in once per update and never shuffled).  Needs a capturable optimizer (Adam(capturable=True)) and
    parameters that were stepped eagerly at least once (allocator, GEMM workspaces, Adam state)."""

    def __init__(self, model, opt, R, A, obs_dim, x_dtype, rows_mb, clip, vf_coef, ent_coef, device):
        self.R, self.A, self.rows_mb = R, A, rows_mb
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)          # noqa: E731
        self.xs, self.perm = e((R, obs_dim), x_dtype), e((R,), torch.int64)
        self.acts = (e((R * A,), torch.int64), e((R * A,), torch.int64), e((R * A,), torch.int64), e((R * A, 2), torch.float32))
        self.lp_old, self.adv, self.ret = e((R * A,), torch.float32), e((R * A,), torch.float32), e((R * A,), torch.float32)
        self.graphs, self.out = [], None
        pool = None
        for s in range(0, R, rows_mb):
            t = min(R, s + rows_mb)
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)                            # the graph's backward allocates (and from then on overwrites) the gradients
            with torch.cuda.graph(g, pool=pool):
                loss, out = _FusedPPOLossPacked.apply(model.trunk_packed(self.xs[s:t]).float(), model.log_std, self.acts[0], self.acts[1], self.acts[2],
                                                      self.acts[3], self.lp_old, self.adv, self.ret, clip, vf_coef, ent_coef, A, self.perm[s:t])
                loss.backward()
                nn.utils.clip_grad_norm_(model.parameters(), 0.5, foreach=True)
                opt.step()
            pool = pool or g.pool()                                     # replayed one after the other, in capture order: one pool serves all
            self.graphs.append(g)
            self.out = out

    def run(self, x_all, actions, logp_old, adv, ret, epochs):
        for src, dst in zip(actions + (logp_old, adv, ret), self.acts + (self.lp_old, self.adv, self.ret)):
            dst.copy_(src.view(dst.shape))
        for _ in range(epochs):
            torch.randperm(self.R, device=x_all.device, out=self.perm)
            torch.index_select(x_all, 0, self.perm, out=self.xs)
            for g in self.graphs:
                g.replay()
        return {"pg_loss": self.out[0], "v_loss": self.out[1], "entropy": self.out[2]}


def ppo_update(model, opt, obs, actions, logp_old, adv, ret, epochs=4, minibatch=262144, clip=0.2, vf_coef=0.5, ent_coef=0.01, amp=False, fused=None,
               agents_per_row=1, graphs=None):
    """amp: the MLPs' matrix products of the update run in bfloat16 on the MFMA units (the observation batch is cast ONCE per
    update, activations are kept in bfloat16; log-softmax / log-prob / losses, parameters and Adam state stay float32) - the
    update is the learner-bound part of an iteration.  Per epoch the whole batch is shuffled with ONE gather per tensor and the
    minibatches are contiguous slices of the shuffled copy (no per-minibatch index kernels); nothing is read back to the host
    until the update is over.
    agents_per_row = A > 1: `obs` holds the R unique observations (one row per market-step) and the per-sample tensors hold R * A
    entries, sample r * A + a belonging to row r (module docstring, "Shared observations").  Rows are shuffled - a minibatch of
    `minibatch` samples is minibatch / A whole rows - and the network sees every row once per epoch.
    graphs: a dict the caller keeps between calls; when given (HIP tensors, fused loss, a capturable optimizer) the first call runs
    eagerly and then captures the minibatch steps as HIP graphs (_GraphedUpdate); later calls of the same shape replay them."""
    A = int(agents_per_row)
    R = obs.shape[0]
    B = R * A
    assert adv.numel() == B and logp_old.numel() == B and actions[0].numel() == B, "per-sample tensors must hold rows * agents_per_row entries"
    fused = obs.is_cuda if fused is None else bool(fused)         # the HIP loss kernel (cda_ppo_loss); the PyTorch statement elsewhere
    if getattr(model, "state_dependent_log_std", False):
        fused = False                                             # (cda_ppo_loss takes ONE log_std pair: the state-dependent head goes through the PyTorch statement)
    adv = ((adv - adv.mean()) / (adv.std() + 1e-8)).float().contiguous()
    logp_old, ret = logp_old.float().contiguous(), ret.float().contiguous()
    actions = (actions[0].contiguous(), actions[1].contiguous(), actions[2].contiguous(), actions[3].float().contiguous())
    by_row = lambda t: t.view(R, A, *t.shape[1:])                 # noqa: E731 - [R * A, ...] -> [R, A, ...]
    flat = lambda t: t.reshape(-1, *t.shape[2:])                  # noqa: E731 - back to one entry per sample
    x_all = obs.to(torch.bfloat16) if amp else obs
    rows_mb = max(1, minibatch // A)
    stats = {}
    graph_key = (R, A, obs.shape[1], x_all.dtype, rows_mb, clip, vf_coef, ent_coef, id(model), id(opt))     # (the captured graphs hold raw references to THIS model and optimiser)
    use_graphs = graphs is not None and fused and obs.is_cuda
    if use_graphs and graphs.get("key") == graph_key:
        stats = graphs["update"].run(x_all, actions, logp_old, adv, ret, epochs)
        return {k: float(v) for k, v in stats.items()}
    def eager_epochs():
        stats = {}
        for _ in range(epochs):
            perm = torch.randperm(R, device=obs.device)
            xs = x_all[perm]                                          # the shuffle moves the observations; a row's samples are found through `perm`
            for s in range(0, R, rows_mb):
                e = min(R, s + rows_mb)
                if fused:
                    loss, out = _FusedPPOLossPacked.apply(model.trunk_packed(xs[s:e]).float(), model.log_std, actions[0], actions[1], actions[2], actions[3],
                                                          logp_old, adv, ret, clip, vf_coef, ent_coef, A, perm[s:e])
                    pg, vl, ent_m = out[0], out[1], out[2]
                else:
                    rows = perm[s:e]
                    pick = lambda t: flat(by_row(t)[rows])           # noqa: E731 - the minibatch's samples, row by row
                    logp, ent, v = model.evaluate(xs[s:e], tuple(pick(a) for a in actions), agents_per_row=A)
                    ratio = (logp - pick(logp_old)).exp()
                    a_mb = pick(adv)
                    pg = -torch.min(ratio * a_mb, ratio.clamp(1 - clip, 1 + clip) * a_mb).mean()
                    vl = (v - pick(ret)).pow(2).mean()
                    ent_m = ent.mean()
                    loss = pg + vf_coef * vl - ent_coef * ent_m
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(model.parameters(), 0.5, foreach=True)
                opt.step()
                stats = {"pg_loss": pg.detach(), "v_loss": vl.detach(), "entropy": ent_m.detach()}
        return stats

    stats = eager_epochs()                                   # (its locals - the last autograd graph among them - are gone before any capture)
    if use_graphs:                                            # this call ran eagerly (it warmed everything up): capture for the next ones
        graphs.clear()
        graphs["update"] = _GraphedUpdate(model, opt, R, A, obs.shape[1], x_all.dtype, rows_mb, clip, vf_coef, ent_coef, obs.device)
        graphs["key"] = graph_key
    return {k: float(v) for k, v in stats.items()}          # one host sync per update, not one per minibatch


def _join(env):
    """A groups > 1 env leaves each group's outputs on that group's stream: this loop consumes them on the current one."""
    if getattr(env, "groups", 1) > 1:
        env.join()


def new_sampler_state(seed, device):
    return int(seed), torch.zeros(1, dtype=torch.int64, device=device)


def _capture_policy_step(model, env, N, A, seed=0, shared=True):
    """HIP graph of: (observation broadcast ->) policy/value forward -> ONE sampling launch (cda_policy_sample: the three categorical and
    two Gaussian heads, log-probability, the env's action tensors).  The sampler's draw counter lives on the device and is bumped
    inside the graph, so every replay draws fresh numbers.  shared: the network reads the env's [N, obs_dim] buffer as it is."""
    state = new_sampler_state(seed, env.obs.device)
    rows = (lambda: env.obs) if shared else (lambda: env.obs.repeat_interleave(A, dim=0))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):                                           # warm-up outside capture (allocator, lazy init)
            model.act_fused(rows(), N, A, state, shared=shared)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        pobs = rows()
        actions, logp, val, env_acts = model.act_fused(pobs, N, A, state, shared=shared)
    # `state` is returned because the graph holds the RAW address of its draw counter (and bumps it on every replay): the tensor
    # must live as long as the graph does
    return g, (pobs, actions, logp, val, env_acts), state


def _capture_rollout_step(model, env, N, A, T, seed=0, shared=True):
    """HIP graph of ONE WHOLE rollout step - policy step (as _capture_policy_step), the env step itself (cda_step / cda_step_groups are
    plain kernel launches on the capturing stream, or forked from it; the device-side auto reset included) and the writes of everything
    the update needs into [T, ...] rollout buffers at a step index that lives on the device and is bumped inside the graph.  A rollout
    is then T replays and nothing else: no per-step host work beyond one graph launch.  Needs an auto_reset env (episode ends are
    handled on the device) and no per-step host callback."""
    dev = env.obs.device
    state = new_sampler_state(seed, dev)
    rows, B = (N if shared else N * A), N * A
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)    # noqa: E731
    buf = {"obs": e((T, rows, env.obs_dim), torch.float32), "val": e((T, rows), torch.float32), "logp": e((T, B), torch.float32),
           "a_cat": e((T, B), torch.int64), "a_price": e((T, B), torch.int64), "a_off": e((T, B), torch.int64), "a_cont": e((T, B, 2), torch.float32),
           "rew": e((T, N, A), torch.float64), "term": e((T, N), torch.bool), "trunc": e((T, N), torch.bool)}
    t_dev = torch.zeros(1, dtype=torch.int64, device=dev)

    def put(pairs, bump):
        """slot *t_dev of every named buffer <- its tensor, ONE launch (cda_store_slots; t read on the device)"""
        import ctypes as C
        from ._lib import check, lib
        n = len(pairs)
        xs = [x.contiguous() for _, x in pairs]
        src = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        dst = (C.c_void_p * n)(*[buf[name].data_ptr() for name, _ in pairs])
        nb = (C.c_int64 * n)(*[x.numel() * x.element_size() for x in xs])
        for (name, _), x in zip(pairs, xs):
            assert buf[name][0].numel() * buf[name].element_size() == x.numel() * x.element_size(), name
        check(lib().cda_store_slots(n, src, dst, nb, t_dev.data_ptr(), int(bump), torch.cuda.current_stream(dev).cuda_stream), "cda_store_slots")

    def policy_part():
        pobs = env.obs if shared else env.obs.repeat_interleave(A, dim=0)
        actions, logp, val, env_acts = model.act_fused(pobs, N, A, state, shared=shared)
        put([("obs", pobs), ("val", val), ("logp", logp)] + list(zip(("a_cat", "a_price", "a_off", "a_cont"), actions)), bump=False)
        return env_acts

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):                                           # warm-up outside capture (allocator, lazy init); the env is not stepped
            policy_part()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        env_acts = policy_part()
        _, r, term, trunc, _ = env.step(*env_acts)                    # groups > 1: forks from / joins into the capturing stream
        put([("rew", r), ("term", term), ("trunc", trunc)], bump=True)
    buf["_sampler_state"] = state            # the graph holds the RAW address of the draw counter: it must live as long as the graph
    return g, buf, t_dev


#: the objective of this module's loops (the round-2 choice: PPO as commonly stated)
PPO_DEFAULTS = dict(gamma=0.99, lam=0.95, clip=0.2, vf_coef=0.5, ent_coef=0.01, max_norm=0.5, kl_coef=0.0, kl_target=0.01, vf_clip=0.0, bootstrap_truncation=False,
                    reward_scale=1e-3)
#: the objective the REFERENCE optimises: train/train.py:519-526 sets only the batch size, the epochs and the learning rate on RLlib's PPOConfig, so everything else
#: is RLlib's default - clip_param 0.3, lambda_ 1.0, gamma 0.99, vf_loss_coeff 1.0, entropy_coeff 0, vf_clip_param 10 (the squared value error clamped), use_kl_loss
#: with kl_coeff 0.2 adapted towards kl_target 0.01 (x 1.5 above twice the target, x 0.5 below half of it, after every update), grad_clip None, rewards
#: unscaled, advantages standardised per module batch, time-limit truncations bootstrapped with V(last observation).
RLLIB_DEFAULTS = dict(gamma=0.99, lam=1.0, clip=0.3, vf_coef=1.0, ent_coef=0.0, max_norm=math.inf, kl_coef=0.2, kl_target=0.01, vf_clip=10.0, bootstrap_truncation=True,
                      reward_scale=1.0)


def adapt_kl_coef(kl_coef, sampled_kl, kl_target):
    """RLlib's rule (PPOLearner._update_module_kl_coeff): the coefficient follows the KL the last minibatch step measured"""
    if kl_coef <= 0.0 or not math.isfinite(sampled_kl):
        return kl_coef
    if sampled_kl > 2.0 * kl_target:
        return kl_coef * 1.5
    if sampled_kl < 0.5 * kl_target:
        return kl_coef * 0.5
    return kl_coef


def train_fused(env, iters=4, horizon=64, lr=5e-5, epochs=4, reward_scale=None, seed=0, log=print, use_graph=True, chains=4, minibatch=262144,
                gamma=None, lam=None, clip=None, vf_coef=None, ent_coef=None, policy=None, keep=None, sub_batches=None, objective=None, recorder=None, info_markets=0,
                allreduce=None, world=1, first_market=0, episode_metrics=True, strict_nav_check=True, state_dependent_log_std=False, hidden=(256, 256)):
    """The PPO loop on the hand-written network kernels (mlp.py, include/cda_mlp.h): rollouts as `chains` independent per-chain launch
    sequences (policy forward + sampling -> env step -> auto reset, one HIP graph per chain, no cross-stream edge inside the horizon), the
    sample records completed by one GAE launch, the update as {gather + forward + loss + back-propagation, weight gradients, reduce, clip + Adam}
    per minibatch step - no autograd, no GEMM library.
    objective: a dict over PPO_DEFAULTS' keys (RLLIB_DEFAULTS = what the reference's RLlib run optimises); the explicit keyword arguments override it.
    recorder + info_markets = S: the last S markets run as a chain of their own with the info tensors of every step (RolloutChains.info) and the recorder
    (episode_record.BatchedEpisodeRecorder over those markets) is fed from the rollout buffers after the horizon - no per-step host call.
    allreduce / world / first_market: a DATA-PARALLEL learner (one process per GPU): every rank runs this loop on its own env shard (its markets are the global
    markets [first_market, first_market + N): seeds and sampling keys follow the global index), rolls out and back-propagates locally, and the ranks sum the
    gradient (parallel.make_grad_allreduce: one all-reduce of 0.9 MB per minibatch step, mlp.FusedUpdate) and the two advantage sums of a rollout - no
    observation ever crosses the fabric.  Every rank starts from the same parameters (same `seed`) and applies the same steps.
    episode_metrics (default on): every episode that ends anywhere inside a rollout is checked (exact sum of NAV) and tallied ON THE DEVICE, in the cold path of the
    in-kernel auto reset (CDAVecEnv.enable_episode_metrics); history[i]["episode_metrics"] holds what the reference's callback logs per episode
    (episode_metrics.summarise: pass / rejection fractions, reward-term means and variance shares, NAV / drawdown / inventory at the episode's end, the most maker-like
    agent's passive share) and a violation raises NavConservationError like the reference's strict_nav_check run (train/train.py:1125-1164); strict_nav_check=False logs it.
    state_dependent_log_std: a fresh policy gets RLlib's default head for Box actions (two log-stds per row from the policy network: mlp.FusedPolicy); a given `policy`
    brings its own.  hidden: `fcnet_hiddens` of a fresh policy, <= 256 each (mlp.init_theta).
    Needs a HIP CDAVecEnv with auto_reset and 168-float observations.  Returns (FusedPolicy, history); `keep` (a dict) receives the last
    rollout's buffers and the RolloutChains object.  history[i]: losses, `mean_reward` (of the rollout's slice of the episodes - it depends on WHICH part of
    the episodes the slice covers) and `episode_return` (mean return of the episodes that were COMPLETED during the iteration, None if none was)."""
    from .mlp import EpisodeReturns, FusedPolicy, FusedUpdate, RolloutChains
    obj = dict(PPO_DEFAULTS)
    obj.update(objective or {})
    for k_, v_ in (("gamma", gamma), ("lam", lam), ("clip", clip), ("vf_coef", vf_coef), ("ent_coef", ent_coef), ("reward_scale", reward_scale)):
        if v_ is not None:
            obj[k_] = v_
    dev = env.obs.device
    N, A, T = env.n_markets, env.num_agents, int(horizon)
    if policy is None:
        policy = FusedPolicy(dev, seed=seed, n_hist=env.n_hist, state_dependent_log_std=state_dependent_log_std, hidden=hidden)
    env.reset(seed=seed + int(first_market))
    if episode_metrics:
        env.enable_episode_metrics(True)
    use_kl = obj["kl_coef"] > 0.0
    roll = RolloutChains(env, policy, T, groups=chains, seed=seed + 7919 * int(first_market), use_graphs=use_graph, with_dist=use_kl,
                         capture_ends=bool(obj["bootstrap_truncation"]), info_markets=info_markets if recorder is not None else 0)
    R = T * N
    rows_mb = max(32, min(R, (max(1, minibatch // A) // 32) * 32))
    import os
    sub_batches = int(os.environ.get("CDA_PPO_SUB_BATCHES", "1")) if sub_batches is None else int(sub_batches)
    upd = FusedUpdate(policy, R, rows_mb, A, sub_batches=sub_batches, allreduce=allreduce, world=world)
    returns = EpisodeReturns(N, A, dev)
    kl_coef = float(obj["kl_coef"])
    history = []
    for it in range(iters):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        buf = roll.run()
        records = roll.gae(gamma=obj["gamma"], lam=obj["lam"], reward_scale=obj["reward_scale"])     # advantages + returns into the sample records, one launch
        if allreduce is not None and world > 1:                              # the advantages are standardised over the GLOBAL batch
            allreduce(records[1])
            records = (records[0], records[1], records[2] * world)
        torch.cuda.synchronize(dev)
        t_roll = time.perf_counter()
        upd.set_extra(kl_coef=kl_coef, vf_clip=obj["vf_clip"], dist_old=buf.get("dist"), log_std_old=roll.log_std_old if use_kl else None)
        stats = upd.run(buf["obs"][:T].view(R, -1), epochs=epochs, clip=obj["clip"], vf_coef=obj["vf_coef"], ent_coef=obj["ent_coef"], lr=lr, max_norm=obj["max_norm"],
                        records=records)
        em = env.collect_episode_metrics() if episode_metrics else None      # (two small launches, inside the timed region: the episodes that ended during this rollout)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        acc = returns.update(buf, T).cpu()                                   # (outside the timed region, like everything below: logging only)
        stats = {k: float(v) for k, v in stats.items()}
        done = float(acc[1].sum())
        stats.update(iter=it, mean_reward=float(buf["reward"].mean()), episode_return=(float(acc[0].sum()) / done) if done else None, episodes=done / A,
                     kl_coef=kl_coef, agent_steps=N * A * T, agent_steps_per_s=N * A * T / (t1 - t0), rollout_s=t_roll - t0, update_s=t1 - t_roll)
        kl_coef = adapt_kl_coef(kl_coef, stats["kl"], obj["kl_target"])
        roll.check_capture_overflow()                                       # (outside the timed region; warns)
        if recorder is not None and roll.info is not None:
            recorder.record_rollout(roll, iteration=it)
        if em is not None:
            from . import episode_metrics as EM
            summ = EM.summarise(*em, module_names=["policy_0"])
            stats["episode_metrics"] = dict(summ.get("all", {}), episodes=summ["episodes"], nav_conservation_violations=summ["nav_conservation_violations"],
                                            nav_conservation_error=summ["nav_conservation_error"], maker_fill_ratio_max=summ.get("maker_fill_ratio_max"),
                                            episode_len_mean=summ.get("episode_len_mean"))
        history.append(stats)
        log(json.dumps(stats))
        if em is not None:
            EM.check_nav_conservation(it, summ, strict=strict_nav_check, log=log)
    if keep is not None:
        keep.update(buffers=roll.buf, rollout=roll, update=upd)
    return policy, history


def train(env, iters=4, horizon=64, lr=5e-5, epochs=4, reward_scale=1e-3, seed=0, log=print, use_graph=True, rollout_hook=None, amp=None,
          shared_obs=True):
    """On-device PPO over a CDAVecEnv-shaped env. Returns per-iteration stats (incl. agent-steps/s).
    rollout_hook(iteration, step, env_actions, obs, reward, terminated, truncated): called after every env step with the
    five [N,A] action tensors the policy produced and the step's output tensors (device tensors; clone what you keep).
    shared_obs: run the network once per market-step instead of once per agent (module docstring); False = one row per sample.
    use_graph (HIP devices): the rollout step is replayed from a captured HIP graph - the whole step (policy, env step, buffer writes:
    _capture_rollout_step) on an auto_reset env without a rollout_hook, else the policy step alone (_capture_policy_step) - and from
    the second iteration on the update's minibatch steps are too (_GraphedUpdate)."""
    torch.manual_seed(seed)
    dev = env.obs.device
    amp = (dev.type == "cuda") if amp is None else bool(amp)
    N, A = env.n_markets, env.num_agents
    per_row = A if shared_obs else 1
    model = ActorCritic(env.obs_dim).to(dev)
    hip = dev.type == "cuda"
    opt = torch.optim.Adam(model.parameters(), lr=lr, fused=hip, capturable=hip and bool(use_graph))     # one kernel per step instead of one per tensor
    env.reset(seed=seed)
    auto_reset = bool(getattr(env, "config", {}).get("auto_reset", False))
    history = []
    # The policy step of the rollout (MLP forward, five samplers, log-probabilities: ~60 small kernels) is launch bound
    # next to a 40-us env step, so it is captured ONCE in a HIP graph that reads the env's own observation buffer and
    # replayed every step; with a device-side auto reset and no host callback the env step and the buffer writes are in it too.
    policy_step = rollout_step = None
    if hip and use_graph:
        try:
            if auto_reset and rollout_hook is None:
                rollout_step = _capture_rollout_step(model, env, N, A, horizon, seed=seed, shared=shared_obs)
            else:
                policy_step = _capture_policy_step(model, env, N, A, seed=seed, shared=shared_obs)
        except Exception as e:  # noqa: BLE001 - eager rollouts are always available
            log(json.dumps({"hip_graph": f"capture failed, eager rollout: {e}"}))
            policy_step = rollout_step = None
    update_graphs = {} if (hip and use_graph) else None
    for it in range(iters):
        t0 = time.perf_counter()
        if rollout_step is not None:
            g, buf, t_dev = rollout_step
            t_dev.zero_()
            for _ in range(horizon):
                g.replay()
            obs_rows = buf["obs"].view(-1, env.obs_dim)
            acts = tuple(buf[k].view(horizon * N * A, *buf[k].shape[2:]) for k in ("a_cat", "a_price", "a_off", "a_cont"))
            logp_all = buf["logp"].view(-1)
            val = buf["val"].repeat_interleave(A, dim=1) if shared_obs else buf["val"]
            rew = (buf["rew"].float() * reward_scale).view(horizon, N * A)
            dn = (buf["term"] | buf["trunc"]).repeat_interleave(A, dim=1).float()
        else:
            buf_obs, buf_act, buf_logp, buf_val, buf_rew, buf_done = [], [], [], [], [], []
            for _ in range(horizon):
                if policy_step is not None:
                    g, (pobs_s, actions_s, logp_s, val_s, env_acts_s), _state = policy_step
                    g.replay()                                           # reads env.obs (this step's observation)
                    pobs, actions, logp, val = pobs_s.clone(), tuple(x.clone() for x in actions_s), logp_s.clone(), val_s.clone()
                    o, r, term, trunc, _ = env.step(*env_acts_s)
                    _join(env)
                    if rollout_hook is not None:
                        rollout_hook(it, len(buf_obs), env_acts_s, o, r, term, trunc)
                else:
                    pobs = env.obs.clone() if shared_obs else env.obs.repeat_interleave(A, dim=0)   # every agent of a market sees the same vector
                    with torch.no_grad():
                        if shared_obs:                                   # one forward per market; its outputs serve the market's A draws
                            o_rows, val = model.trunk(pobs)
                            o_rep = o_rows.float().repeat_interleave(A, dim=0)
                            cat, price, off, cont = model._dists(o_rep)
                            a_cat, a_price, a_off = cat.sample(), price.sample(), off.sample()
                            a_cont = (cont.loc + cont.scale * torch.randn_like(cont.loc)).detach()
                            logp = cat.log_prob(a_cat) + price.log_prob(a_price) + off.log_prob(a_off) + cont.log_prob(a_cont).sum(-1)
                            actions, val = (a_cat, a_price, a_off, a_cont), val.float()
                        else:
                            actions, logp, val = model.act(pobs)
                    env_acts = to_env_actions(actions, N, A)
                    o, r, term, trunc, _ = env.step(*env_acts)
                    _join(env)
                    if rollout_hook is not None:
                        rollout_hook(it, len(buf_obs), env_acts, o, r, term, trunc)
                done = (term | trunc)
                if shared_obs:
                    val = val.repeat_interleave(A)                       # the market's value, once per agent (returns differ per agent)
                buf_obs.append(pobs); buf_act.append(actions); buf_logp.append(logp); buf_val.append(val)
                buf_rew.append((r.float() * reward_scale).reshape(-1)); buf_done.append(done.repeat_interleave(A).float())
                if not auto_reset and bool(done.any()):                # (a host sync per step; auto_reset envs reset on the device)
                    env.reset(mask=done)                               # seed=None semantics: streams continue
            flat = lambda xs: torch.cat(xs, 0)                         # noqa: E731
            obs_rows, logp_all = flat(buf_obs), flat(buf_logp)
            acts = tuple(flat([b[i] for b in buf_act]) for i in range(4))
            rew, val, dn = torch.stack(buf_rew), torch.stack(buf_val), torch.stack(buf_done)
        with torch.no_grad():
            last_val = model.v(env.obs).squeeze(-1).float().repeat_interleave(A)
        adv, ret = gae(rew, val, last_val, dn)
        if hip:
            torch.cuda.synchronize()                                   # (so that rollout_s / update_s are device times, not enqueue times)
        t_roll = time.perf_counter()
        stats = ppo_update(model, opt, obs_rows, acts, logp_all, adv.reshape(-1), ret.reshape(-1), epochs=epochs, amp=amp,
                           agents_per_row=per_row, graphs=update_graphs)       # (the first update runs eagerly and captures: see _GraphedUpdate)
        if hip:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        stats.update(iter=it, mean_reward=float(rew.mean()) / reward_scale, agent_steps=N * A * horizon,
                     agent_steps_per_s=N * A * horizon / (t1 - t0), rollout_s=t_roll - t0, update_s=t1 - t_roll)
        history.append(stats)
        log(json.dumps(stats))
    return model, history


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--markets", type=int, default=4096)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--horizon", type=int, default=64)
    p.add_argument("--iters", type=int, default=4)
    p.add_argument("--max-step", type=int, default=4096)
    p.add_argument("--n-hist", type=int, default=4, help="history depth of the observation (the reference trains with 4; the fused kernels are compiled for 1, 2, 4, 8)")
    p.add_argument("--fp32-update", action="store_true", help="PPO update in float32 instead of bfloat16 autocast")
    p.add_argument("--groups", type=int, default=1, help="market groups of the env step (independent launch chains, vec_env.CDAVecEnv)")
    p.add_argument("--no-graphs", action="store_true", help="eager rollout and update (no HIP graphs)")
    p.add_argument("--per-sample-forward", action="store_true", help="run the network once per (market, agent) sample instead of once per market-step (shared_obs=False)")
    p.add_argument("--out", default=None, help="write a JSON summary (config, per-iteration stats, end-of-run env checks) to this file")
    p.add_argument("--legacy", action="store_true", help="the round-3 loop: PyTorch network (library GEMMs, autograd), one graph per rollout step")
    p.add_argument("--chains", type=int, default=4, help="fused loop: independent rollout chains (market groups on their own streams)")
    p.add_argument("--fcnet-hiddens", type=int, nargs=2, default=(256, 256), metavar=("H1", "H2"), help="fused loop: the two hidden widths (config/train_config.json:49), <= 256 each")
    p.add_argument("--log-std-head", action="store_true", help="fused loop: the state-dependent log-std head (RLlib's default module for Box actions) instead of a free log_std vector")
    p.add_argument("--objective", choices=("ppo", "rllib"), default="ppo", help="fused loop: PPO_DEFAULTS, or RLLIB_DEFAULTS = the objective the reference's RLlib run optimises "
                                                                              "(clip 0.3, lambda 1, vf coeff 1, entropy 0, vf clip 10, adaptive KL penalty, no gradient clipping, truncation bootstrap)")
    args = p.parse_args(argv)
    from .vec_env import CDAVecEnv
    p_groups = max(1, min(args.groups, args.markets))
    env = CDAVecEnv({"num_of_agents": args.agents, "init_cash": 1000000, "max_step": args.max_step, "is_render": False, "auto_reset": True, "n_hist": args.n_hist},
                    n_markets=args.markets, device="cuda:0", with_info=False, groups=p_groups)
    from .mlp import HIST_VARIANTS
    if not args.legacy and ((args.horizon * args.markets) % 32 or env.n_hist not in HIST_VARIANTS):
        # the fused kernels step whole 32-row tiles and are compiled per history depth: other shapes run the PyTorch statement of the same loop
        print(json.dumps({"note": f"fused loop needs horizon * markets % 32 == 0 and n_hist in {HIST_VARIANTS} (got {args.horizon} x {args.markets}, n_hist {env.n_hist}): running the legacy loop"}))
        args.legacy = True
    if args.legacy:
        _, hist = train(env, iters=args.iters, horizon=args.horizon, amp=not args.fp32_update, shared_obs=not args.per_sample_forward, use_graph=not args.no_graphs)
    else:
        _, hist = train_fused(env, iters=args.iters, horizon=args.horizon, use_graph=not args.no_graphs, chains=args.chains,
                              objective=RLLIB_DEFAULTS if args.objective == "rllib" else None, state_dependent_log_std=args.log_std_head, hidden=tuple(args.fcnet_hiddens))
    flags = env.flags()
    _, bad = env.nav_conservation()
    tail = hist[2:] if len(hist) >= 4 else (hist[1:] or hist)
    summary = {"metric": "agent-steps/sec end to end (rollout + PPO update), BASELINE configs[4]",
               "config": {"workload": f"{args.markets} markets x {args.agents} agents, " + ("PyTorch-ROCm PPO policy (library GEMMs, autograd)" if args.legacy else
                                      "PPO policy on the hand-written bf16 MFMA network kernels") + " in the loop (256x256 tanh actor and critic, "
                                      f"4 epochs, 262144-sample minibatches), horizon {args.horizon}, {args.iters} iterations, auto_reset on",
                          "objective": "legacy loop's own" if args.legacy else args.objective,
                          "markets": args.markets, "agents": args.agents, "horizon": args.horizon, "iters": args.iters, "env_groups": p_groups,
                          "loop": "legacy (PyTorch network)" if args.legacy else (f"fused: hand-written bf16 MFMA network (csrc/cda_mlp.hip), {args.chains} rollout chains; GAE straight into the sample records (one launch); "
                                                                                      "a minibatch step = {gather + forward + loss + backward in one launch, weight gradients, reduce, Adam}"),
                          "hip_graphs": "none" if args.no_graphs else ("one graph per rollout step (policy + env step + buffer writes), one per minibatch step of the update"
                                                                      if args.legacy else "one graph per rollout chain (the whole horizon); the update is plain launches"),
                          "update_dtype": "float32" if args.fp32_update else "bfloat16 operands, float32 accumulation (float32 parameters, Adam state, softmax and losses)",
                          "network_forward": "once per (market, agent) sample" if args.per_sample_forward else
                                             "once per market-step: the market's agents share the observation, the policy is shared, so their logits and value are one row "
                                             "(same samples, same loss, same gradients; ppo.py module docstring)"},
               "iterations": hist,
               # the first two iterations are warm-up (graph capture in the first; the second now and then pays a one-off 40-ms stall on a fresh box)
               "timed_iterations": len(tail),
               "value": sum(h["agent_steps"] for h in tail) / sum(h["rollout_s"] + h["update_s"] for h in tail),
               "rollout_agent_steps_per_s": sum(h["agent_steps"] for h in tail) / sum(h["rollout_s"] for h in tail),
               "unit": "agent-steps/s", "flagged_markets": int((flags != 0).sum().item()), "nav_conservation_violations": int(bad.sum().item()),
               "invariant_violations": int((env.check_invariants() != 0).sum().item())}
    print(json.dumps(summary))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(summary, fh, indent=1)
    env.close()


if __name__ == "__main__":
    main()
