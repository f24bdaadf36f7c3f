"""The policy / value network of the PPO loop as hand-written bf16 MFMA kernels (include/cda_mlp.h, csrc/cda_mlp.hip): host side.

`FusedPolicy` owns the f32 master parameters (one vector `theta`), the Adam state and the bf16 operand copies the kernels multiply
with; `RolloutChains` runs whole rollouts as independent per-chain launch sequences (policy forward + sampling -> env step -> auto
reset, no cross-stream edge until the end of the horizon); `FusedUpdate` is the PPO update (forward, loss, back-propagation, weight
gradients, clipping + Adam) on those kernels.  `ppo.ActorCritic` stays the plain PyTorch statement of the same network - the numerics
reference of the tests (`FusedPolicy.to_actor_critic`, `reference_outputs`) and the path of non-HIP tensors.

Reference: the network of config/train_config.json:45-53 (separate policy and value MLPs, 256 x 256, tanh), trained through RLlib's PPO
at train/train.py:453-541; every agent of a market is handed the same observation (envs/exchg/state_helper.py:76,109), so the network
runs once per market-step.
"""
import ctypes as C
import math

import torch

OBS, KX, XT, HID, FEAT, NOUT, N_LOGITS = 168, 176, 6, 256, 512, 32, 24
OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_WO, OFF_BO, OFF_LS, PARAMS = 0, 86016, 86528, 217600, 218112, 226304, 226336, 226338
WB_ELEMS, SLAB, BSLAB = 385024, 245760, 1056


def _lib():
    from ._lib import lib
    return lib()


def _check(rc, what):
    from ._lib import check
    check(rc, what)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def init_theta(obs_dim=OBS, generator=None):
    """A fresh parameter vector with nn.Linear's default initialisation per block (uniform +-1/sqrt(fan_in)), log_std = -0.5."""
    assert obs_dim == OBS, "the fused network is built for the reference's 4 x 42 observation"
    u = lambda n, fan_in: (torch.rand(n, generator=generator) * 2 - 1) / math.sqrt(fan_in)       # noqa: E731
    th = torch.zeros(PARAMS)
    th[OFF_W1:OFF_B1] = u(FEAT * OBS, OBS); th[OFF_B1:OFF_W2] = u(FEAT, OBS)
    th[OFF_W2:OFF_B2] = u(2 * HID * HID, HID); th[OFF_B2:OFF_WO] = u(FEAT, HID)
    wo = u(NOUT * HID, HID).view(NOUT, HID); wo[N_LOGITS + 1:] = 0
    bo = u(NOUT, HID); bo[N_LOGITS + 1:] = 0
    th[OFF_WO:OFF_BO] = wo.reshape(-1); th[OFF_BO:OFF_LS] = bo
    th[OFF_LS:] = -0.5
    return th


def theta_from_actor_critic(model):
    """ppo.ActorCritic (block matrices with masks) -> the fused layout."""
    H = model.hidden
    assert H == HID and model.l1.weight.shape[1] == OBS
    th = torch.zeros(PARAMS, dtype=torch.float32)
    with torch.no_grad():
        th[OFF_W1:OFF_B1] = model.l1.weight.detach().float().cpu().reshape(-1)
        th[OFF_B1:OFF_W2] = model.l1.bias.detach().float().cpu()
        w2 = model.l2.weight.detach().float().cpu()
        th[OFF_W2:OFF_B2] = torch.stack([w2[:H, :H], w2[H:, H:]]).reshape(-1)
        th[OFF_B2:OFF_WO] = model.l2.bias.detach().float().cpu()
        wo = model.out.weight.detach().float().cpu()
        blk = torch.zeros(NOUT, H)
        blk[:N_LOGITS] = wo[:N_LOGITS, :H]; blk[N_LOGITS] = wo[N_LOGITS, H:]
        th[OFF_WO:OFF_BO] = blk.reshape(-1)
        bo = model.out.bias.detach().float().cpu().clone(); bo[N_LOGITS + 1:] = 0
        th[OFF_BO:OFF_LS] = bo
        th[OFF_LS:] = model.log_std.detach().float().cpu()
    return th


def actor_critic_from_theta(theta, dtype=torch.float32):
    from .ppo import ActorCritic
    th = theta.detach().float().cpu()
    m = ActorCritic(OBS).to(dtype)
    H = HID
    with torch.no_grad():
        m.l1.weight.copy_(th[OFF_W1:OFF_B1].view(FEAT, OBS)); m.l1.bias.copy_(th[OFF_B1:OFF_W2])
        w2 = th[OFF_W2:OFF_B2].view(2, H, H)
        m.l2.weight.zero_(); m.l2.weight[:H, :H] = w2[0]; m.l2.weight[H:, H:] = w2[1]; m.l2.bias.copy_(th[OFF_B2:OFF_WO])
        wo = th[OFF_WO:OFF_BO].view(NOUT, H)
        m.out.weight.zero_(); m.out.weight[:N_LOGITS, :H] = wo[:N_LOGITS]; m.out.weight[N_LOGITS, H:] = wo[N_LOGITS]
        m.out.bias.copy_(th[OFF_BO:OFF_LS]); m.log_std.copy_(th[OFF_LS:])
    return m


def _r(t):
    """round to bfloat16 and back (what an MFMA operand sees)"""
    return t.to(torch.bfloat16).to(t.dtype)


def reference_outputs(theta, x, emulate_bf16=True, dtype=torch.float64, keep=False):
    """The network in plain PyTorch on the CPU: out [n, 32].  emulate_bf16: operands (inputs, weights, activations between layers) rounded
    to bfloat16 as the kernels do, products and sums in `dtype`.  keep: also return (xb, h1, h2) as the kernels store them."""
    th = theta.detach().cpu().to(dtype)
    x = x.detach().cpu().to(dtype)
    rd = _r if emulate_bf16 else (lambda t: t)
    W1, b1 = rd(th[OFF_W1:OFF_B1].view(FEAT, OBS)), th[OFF_B1:OFF_W2]
    W2, b2 = rd(th[OFF_W2:OFF_B2].view(2, HID, HID)), th[OFF_B2:OFF_WO]
    Wo, bo = rd(th[OFF_WO:OFF_BO].view(NOUT, HID)), th[OFF_BO:OFF_LS]
    xb = rd(x)
    h1 = rd(torch.tanh(xb @ W1.t() + b1))
    h2 = torch.cat([rd(torch.tanh(h1[:, :HID] @ W2[0].t() + b2[:HID])), rd(torch.tanh(h1[:, HID:] @ W2[1].t() + b2[HID:]))], dim=1)
    out = torch.zeros(x.shape[0], NOUT, dtype=dtype)
    out[:, :N_LOGITS] = h2[:, :HID] @ Wo[:N_LOGITS].t() + bo[:N_LOGITS]
    out[:, N_LOGITS] = h2[:, HID:] @ Wo[N_LOGITS] + bo[N_LOGITS]
    return (out, xb, h1, h2) if keep else out


def reference_gradients(theta, xb, h1, h2, d_out, dtype=torch.float64):
    """The back-propagation the kernels perform, in plain PyTorch: gradient of theta (dense vector, log_std entries zero) for given
    d_out [n, 32], with the kernels' roundings (d_out, dz2, dz1 rounded to bfloat16 where they become operands)."""
    th = theta.detach().cpu().to(dtype)
    W2 = _r(th[OFF_W2:OFF_B2].view(2, HID, HID)); Wo = _r(th[OFF_WO:OFF_BO].view(NOUT, HID))
    d_out = d_out.detach().cpu().to(dtype)
    dob = _r(d_out)
    dh2 = torch.cat([dob[:, :N_LOGITS] @ Wo[:N_LOGITS], dob[:, N_LOGITS:N_LOGITS + 1] @ Wo[N_LOGITS:N_LOGITS + 1]], dim=1)
    dz2 = _r(dh2 * (1 - h2 * h2))
    dh1 = torch.cat([dz2[:, :HID] @ W2[0], dz2[:, HID:] @ W2[1]], dim=1)
    dz1 = _r(dh1 * (1 - h1 * h1))
    g = torch.zeros(PARAMS, dtype=dtype)
    g[OFF_W1:OFF_B1] = (dz1.t() @ xb).reshape(-1); g[OFF_B1:OFF_W2] = dz1.sum(0)
    g[OFF_W2:OFF_B2] = torch.stack([dz2[:, :HID].t() @ h1[:, :HID], dz2[:, HID:].t() @ h1[:, HID:]]).reshape(-1); g[OFF_B2:OFF_WO] = dz2.sum(0)
    gwo = torch.zeros(NOUT, HID, dtype=dtype)
    gwo[:N_LOGITS] = dob[:, :N_LOGITS].t() @ h2[:, :HID]; gwo[N_LOGITS] = dob[:, N_LOGITS] @ h2[:, HID:]
    g[OFF_WO:OFF_BO] = gwo.reshape(-1)
    gbo = torch.zeros(NOUT, dtype=dtype); gbo[:N_LOGITS + 1] = d_out[:, :N_LOGITS + 1].sum(0)
    g[OFF_BO:OFF_LS] = gbo
    return g, dz1, dz2


def unpack_rows(packed, n_rows, n_feat, paired=False):
    """packed bf16 image [n_rows/32][n_feat/32][2][64][8] -> [n_rows, n_feat] (host; tests and diagnostics).  paired: the feature tiles of
    the hidden activations (h1 / h2 / dz1 / dz2): position q of tile ft is feature 64 (ft // 2) + 2 q + (ft & 1) (csrc/cda_mlp.hip feature_of)."""
    p = packed.detach().cpu().float().view(n_rows // 32, n_feat // 32, 2, 64, 8)
    out = torch.zeros(n_rows, n_feat)
    lane = torch.arange(64)
    j, h = lane & 31, lane >> 5
    for ks in range(2):
        for e in range(8):
            r = 8 * ks + e
            row = (r & 3) + 8 * (r >> 2) + 4 * h                       # [64]
            for rt in range(n_rows // 32):
                for ft in range(n_feat // 32):
                    col = (64 * (ft // 2) + 2 * j + (ft & 1)) if paired else ft * 32 + j
                    out[rt * 32 + row, col] = p[rt, ft, ks, :, e]
    return out


class FusedPolicy:
    """theta (f32 master copy), Adam state and the bf16 operand blob on one HIP device."""

    def __init__(self, device, theta=None, seed=0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FusedPolicy needs a HIP device; the PyTorch statement of the network is ppo.ActorCritic")
        if theta is None:
            g = torch.Generator().manual_seed(int(seed))
            theta = init_theta(generator=g)
        assert theta.numel() == PARAMS
        self.theta = theta.detach().float().to(self.device).contiguous()
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_step = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.wb = torch.zeros(WB_ELEMS, dtype=torch.bfloat16, device=self.device)
        self.pack()

    @classmethod
    def from_actor_critic(cls, model, device):
        return cls(device, theta=theta_from_actor_critic(model))

    def to_actor_critic(self, dtype=torch.float32):
        return actor_critic_from_theta(self.theta, dtype)

    @property
    def log_std(self):
        return self.theta[OFF_LS:]

    def pack(self):
        _check(_lib().cda_mlp_pack(self.theta.data_ptr(), self.wb.data_ptr(), _stream(self.device)), "cda_mlp_pack")

    def forward(self, obs, first_row=0, n_rows=None, out=None):
        """network outputs f32 [rows, 32] (columns 0..23 policy, 24 value) for rows [first_row, first_row + n_rows) of obs f32[*, 168]"""
        obs = obs.contiguous()
        assert obs.dtype == torch.float32 and obs.shape[-1] == OBS and obs.device == self.device
        n_rows = obs.shape[0] - first_row if n_rows is None else n_rows
        if out is None:
            out = torch.zeros((obs.shape[0], NOUT), dtype=torch.float32, device=self.device)
        _check(_lib().cda_mlp_forward(self.wb.data_ptr(), self.theta.data_ptr(), obs.data_ptr(), int(first_row), int(n_rows), out.data_ptr(),
                                      _stream(self.device)), "cda_mlp_forward")
        return out

    def policy_step(self, obs, num_agents, seed, counter, draw, first_market=0, n_markets=None, outs=None):
        """one launch: forward + sampling for the markets [first_market, first_market + n_markets).  Returns the dict of output tensors
        ([N, A] each; a_cont [N, A, 2]; value [N])."""
        obs = obs.contiguous()
        N = obs.shape[0]
        n_markets = N - first_market if n_markets is None else n_markets
        A, dev = int(num_agents), self.device
        if outs is None:
            e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)         # noqa: E731
            outs = {"category": e((N, A), torch.int32), "size_mean": e((N, A), torch.float32), "size_sigma": e((N, A), torch.float32),
                    "price": e((N, A), torch.int32), "price_offset": e((N, A), torch.int32), "a_cont": e((N, A, 2), torch.float32),
                    "logp": e((N, A), torch.float32), "value": e((N,), torch.float32)}
        _check(_lib().cda_mlp_policy_step(self.wb.data_ptr(), self.theta.data_ptr(), obs.data_ptr(), int(first_market), int(n_markets), A,
                                          int(seed) & (2 ** 64 - 1), counter.data_ptr(), int(draw),
                                          *[outs[k].data_ptr() for k in ("category", "size_mean", "size_sigma", "price", "price_offset", "a_cont", "logp", "value")],
                                          _stream(dev)), "cda_mlp_policy_step")
        return outs


class RolloutChains:
    """Whole rollouts of a CDAVecEnv (auto_reset on) under a FusedPolicy as G independent chains: chain g = the markets of group g, on
    its own stream: for every step {policy forward + sampling -> env step -> auto reset}, then the bootstrap value of the last
    observation.  Weights are frozen during a rollout and markets never interact, so no chain ever waits for another; the caller's
    stream forks into the chains before the first step and joins them after the last.  Buffers are [T(+1), N, ...] and every step's
    kernels read / write their own slot directly - nothing is copied between steps.  Each chain's launch sequence is captured into a HIP
    graph once and replayed (use_graphs), or enqueued by one native call per chain (cda_mlp_rollout_chain)."""

    def __init__(self, env, policy, horizon, groups=4, seed=0, use_graphs=True):
        from ._lib import RolloutBufs
        self.env, self.policy, self.T = env, policy, int(horizon)
        N, A, dev, T = env.n_markets, env.num_agents, env.device, int(horizon)
        if env.obs_dim != OBS:
            raise ValueError("the fused network is built for n_hist = 4 (168-float observations)")
        if not bool(env.config.get("auto_reset", False)):
            raise ValueError("RolloutChains needs an auto_reset env (episode ends are handled on the device)")
        self.N, self.A, self.device = N, A, dev
        e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)             # noqa: E731
        self.buf = {"obs": e((T + 1, N, OBS), torch.float32), "category": e((T, N, A), torch.int32), "size_mean": e((T, N, A), torch.float32),
                    "size_sigma": e((T, N, A), torch.float32), "price": e((T, N, A), torch.int32), "price_offset": e((T, N, A), torch.int32),
                    "a_cont": e((T, N, A, 2), torch.float32), "logp": e((T, N, A), torch.float32), "value": e((T + 1, N), torch.float32),
                    "reward": e((T, N, A), torch.float64), "terminated": e((T, N), torch.uint8), "truncated": e((T, N), torch.uint8),
                    "record": e((T, N, A, 8), torch.float32)}          # include/cda_mlp.h CDA_REC_*: what the update's loss reads, one line per row
        self.adv_stats = torch.zeros(2, dtype=torch.float64, device=dev)
        self._cbufs = RolloutBufs(**{k: v.data_ptr() for k, v in self.buf.items()})
        self.seed = int(seed) & (2 ** 64 - 1)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        G = max(1, min(int(groups), N))
        self.ranges = []
        for g in range(G):
            first, cnt = C.c_int32(), C.c_int32()
            _lib().cda_group_range(N, G, g, C.byref(first), C.byref(cnt))
            self.ranges.append((first.value, cnt.value))
        if G > 1:
            from .streams import concurrent_streams
            self.streams = list(concurrent_streams(dev, G))
        else:
            self.streams = [torch.cuda.current_stream(dev)]
        self._fork = torch.cuda.Event()
        self._joins = [torch.cuda.Event() for _ in range(G)]
        self.graphs = None
        self._have_obs = False
        self.use_graphs = bool(use_graphs)
        self.join_mode = "events"

    def _enqueue(self, g, copy_first_obs):
        first, cnt = self.ranges[g]
        with torch.cuda.device(self.device):
            _check(_lib().cda_mlp_rollout_chain(self.env._h, self.policy.wb.data_ptr(), self.policy.theta.data_ptr(), first, cnt, self.T, self.seed,
                                                self.counter.data_ptr(), C.byref(self._cbufs), int(copy_first_obs), torch.cuda.current_stream(self.device).cuda_stream),
                   "cda_mlp_rollout_chain")

    def _capture(self):
        graphs = []
        for g, s in enumerate(self.streams):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                self._enqueue(g, True)
            graphs.append(gr)
        self.graphs = graphs

    def run(self):
        """one rollout of `horizon` steps; returns the buffer dict (views stay valid; the next run() overwrites them)"""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if not self._have_obs:                                   # the very first rollout starts from the env's current observation
            self.env.join()
            self.buf["obs"][self.T].copy_(self.env.obs)
            self._have_obs = True
        self.counter.add_(1)                                     # fresh draws for this rollout (on the caller's stream, before the fork)
        if self.use_graphs and self.graphs is None and len(self.streams) >= 1:
            torch.cuda.synchronize(dev)
            try:
                self._capture()
            except Exception as ex:  # noqa: BLE001 - the native loop is always available
                import warnings
                warnings.warn(f"HIP graph capture of the rollout chains failed ({ex}); using direct launches")
                self.graphs, self.use_graphs = None, False
        self._fork.record(cur)
        for g, s in enumerate(self.streams):
            if s.cuda_stream != cur.cuda_stream:
                s.wait_event(self._fork)
            with torch.cuda.stream(s):
                if self.graphs is not None:
                    self.graphs[g].replay()
                else:
                    self._enqueue(g, True)
            if s.cuda_stream != cur.cuda_stream:
                self._joins[g].record(s)
        # the joins only after EVERY chain is enqueued: a wait placed on the caller's stream between two chains' launches delays the later
        # chain's start when the caller's stream is the legacy default stream (tools/rollout_probe.py)
        if self.join_mode != "none":
            for g, s in enumerate(self.streams):
                if s.cuda_stream != cur.cuda_stream:
                    cur.wait_event(self._joins[g])
        return self.buf

    def gae(self, gamma=0.99, lam=0.95, reward_scale=1.0):
        """advantages and returns of the last run() straight into the sample records (one launch; ppo.gae's recursion); returns
        (records [T * N, A, 8], the sums the update normalises the advantages with, their count)"""
        _check(_lib().cda_gae_records(self.buf["reward"].data_ptr(), self.buf["value"].data_ptr(), self.buf["terminated"].data_ptr(), self.buf["truncated"].data_ptr(),
                                      self.T, self.N, self.A, float(reward_scale), float(gamma), float(lam), self.buf["record"].data_ptr(), self.adv_stats.data_ptr(),
                                      _stream(self.device)), "cda_gae_records")
        return self.buf["record"].view(self.T * self.N, self.A, 8), self.adv_stats, self.T * self.N * self.A


class FusedUpdate:
    """The PPO update on the kernels of include/cda_mlp.h: per epoch a keyed permutation of the R unique observation rows, per minibatch
    {gather + forward + loss + back-propagation (one launch), weight gradients, reduce, clip + Adam}: four launches, no autograd, no GEMM
    library.  fused=False: the same step as separate kernels (a gather / convert pass per epoch, then forward, loss, backward, ...)."""

    def __init__(self, policy, n_rows, rows_mb, num_agents, chunks=None, sub_batches=1, fused=True):
        """fused (default; used when run() is given sample records): gather, forward, loss and back-propagation of a minibatch are ONE launch
        (cda_mlp_forward_backward) - a minibatch step is {that, weight gradients, reduce, Adam}; otherwise the separate kernels run (prep_rows per
        epoch, forward, loss, backward).
        sub_batches > 1 (separate kernels only): a minibatch step runs {forward, loss, backward, weight gradients} once per sub-batch of rows_mb / sub_batches rows and
        reduces all their partial sums in one optimiser step - the same step, with activations of a sub-batch small enough to stay in the
        256-MB Infinity Cache between the kernel that writes them and the ones that read them."""
        self.p, self.R, self.rows_mb, self.A = policy, int(n_rows), int(rows_mb), int(num_agents)
        self.sub = max(1, int(sub_batches))
        self.fused = bool(fused) and self.sub == 1
        if self.R % 32 or self.rows_mb % 32 or self.rows_mb > self.R:
            raise ValueError("rows and minibatch rows must be multiples of 32")
        dev = policy.device
        self.tile_rows = int(_lib().cda_mlp_tile_rows())
        self.n_tiles = (self.rows_mb + self.tile_rows - 1) // self.tile_rows
        pad = self.n_tiles * self.tile_rows                            # the kernels write whole workgroup tiles
        pad = ((pad + 63) // 64) * 64                                  # (the fused kernel's are 64 rows whatever CDA_MLP_MT says)
        self.chunks = int(chunks) if chunks else max(1, min(51, self.rows_mb // 512))      # 5 jobs x 51 chunks = 255 workgroups: one wave of the 256 CUs
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)                      # noqa: E731
        self.x_rm, self.x_pk = e(self.R * KX, bf), e(self.R * 32 * XT, bf)
        self.h1p, self.h2p, self.dz1p, self.dz2p = (e(pad * FEAT, bf) for _ in range(4))
        self.doutp = e(pad * NOUT, bf)
        self.out, self.d_out = e(pad * NOUT, f32).view(-1, NOUT), e(pad * NOUT, f32).view(-1, NOUT)
        self.slab, self.bias_slab = e(self.sub * self.chunks * SLAB, f32), e((max(self.n_tiles, pad // 64) + self.sub) * BSLAB, f32)
        self.grad, self.norm2 = e(PARAMS, f32), e(512, torch.float64)        # (norm2[2] = the squared gradient norm of the last step)
        self.sums5, self.out6 = e(64 * 8, torch.float64), e(6, f32)       # CDA_MLP_LOSS_SLOTS x 8: the loss sums (slot 0, words 0..4 for the separate loss kernels)
        self.perm = torch.zeros(self.R, dtype=torch.int64, device=dev)
        pad64 = ((self.rows_mb + 63) // 64) * 64
        self.x_pk_mb = e(pad64 * 32 * XT, bf) if self.fused else None         # the fused kernel's packed image of the minibatch's observations
        self.shuffle_seed, self._epochs_done = 0x5DEECE66D, 0

    def minibatch_step(self, s, rows, acts, logp_old, adv, ret, clip, vf_coef, ent_coef, lr, betas, eps, max_norm, apply=True, records=None, obs_rows=None,
                       debug_outputs=False):
        """rows [s, s + rows) of the prepared (shuffled) observations: one optimiser step.  obs_rows (with records, fused=True): the unshuffled f32
        observation rows - the fused kernel gathers rows perm[s .. s + rows) itself, no prepared images needed.  records = (rec f32 [R, A, 8], adv sums f64[2] or
        None, their count): the loss reads sample records (RolloutChains.gae) instead of the seven per-sample arrays."""
        L, p, dev = _lib(), self.p, self.p.device
        st = _stream(dev)
        if obs_rows is not None and records is not None and self.fused:
            rec, stats, count = records
            chunks, tiles = max(1, min(self.chunks, rows // 32)), (rows + 63) // 64
            _check(L.cda_mlp_forward_backward(p.wb.data_ptr(), p.theta.data_ptr(), obs_rows.data_ptr(), self.perm.data_ptr() + s * 8, rows, rows, rec.data_ptr(),
                                              stats.data_ptr() if stats is not None else None, int(count), self.A, float(clip), float(vf_coef), float(ent_coef),
                                              self.x_pk_mb.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(),
                                              self.bias_slab.data_ptr(), self.sums5.data_ptr(), self.out6.data_ptr(), 0 if apply else 1, 0 if apply else 1,
                                              self.out.data_ptr() if debug_outputs else None, self.d_out.data_ptr() if debug_outputs else None, st), "cda_mlp_forward_backward")
            _check(L.cda_mlp_wgrad(self.x_pk_mb.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(), rows, chunks,
                                   self.slab.data_ptr(), st), "cda_mlp_wgrad")
            if apply:
                _check(L.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), self.slab.data_ptr(), chunks,
                                      self.bias_slab.data_ptr(), tiles, self.sums5.data_ptr(), rows * self.A, float(vf_coef), float(ent_coef), self.out6.data_ptr(),
                                      float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm),
                                      self.grad.data_ptr(), self.norm2.data_ptr(), st), "cda_mlp_adam")
            return chunks, tiles
        sub = self.sub if (rows % (32 * self.sub) == 0 and rows // self.sub >= 32) else 1
        rs = rows // sub                                           # rows per sub-batch
        tiles_sub = (rs + self.tile_rows - 1) // self.tile_rows
        chunks = max(1, min(self.chunks, rs // 32))
        for k in range(sub):
            o = s + k * rs
            x_rm = self.x_rm.data_ptr() + o * KX * 2
            x_pk = self.x_pk.data_ptr() + o * 32 * XT * 2
            _check(L.cda_mlp_forward_train(p.wb.data_ptr(), p.theta.data_ptr(), x_rm, rs, self.h1p.data_ptr(), self.h2p.data_ptr(), self.out.data_ptr(), st), "cda_mlp_forward_train")
            last = k == sub - 1
            if records is not None:
                rec, stats, count = records
                _check(L.cda_ppo_loss_records(self.out.data_ptr(), p.theta.data_ptr() + OFF_LS * 4, rec.data_ptr(), stats.data_ptr() if stats is not None else None, int(count),
                                              self.perm.data_ptr() + o * 8, rs, self.A, NOUT, float(clip), float(vf_coef), float(ent_coef), self.d_out.data_ptr(),
                                              self.sums5.data_ptr(), self.out6.data_ptr(), rows, 0 if (apply or k) else 1, 0 if (apply or not last) else 1, st),
                       "cda_ppo_loss_records")
            else:
              _check(L.cda_ppo_loss32(self.out.data_ptr(), p.theta.data_ptr() + OFF_LS * 4, acts[0].data_ptr(), acts[1].data_ptr(), acts[2].data_ptr(), acts[3].data_ptr(),
                                    logp_old.data_ptr(), adv.data_ptr(), ret.data_ptr(), self.perm.data_ptr() + o * 8, rs, self.A, NOUT,
                                    float(clip), float(vf_coef), float(ent_coef), self.d_out.data_ptr(), self.sums5.data_ptr(), self.out6.data_ptr(), rows,
                                    0 if (apply or k) else 1, 0 if (apply or not last) else 1, st),
                     "cda_ppo_loss32")          # (apply: the sums are finished and cleared by cda_mlp_adam; they accumulate over the sub-batches)
            _check(L.cda_mlp_backward(p.wb.data_ptr(), self.d_out.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), rs, self.dz1p.data_ptr(), self.dz2p.data_ptr(),
                                      self.doutp.data_ptr(), self.bias_slab.data_ptr() + k * tiles_sub * BSLAB * 4, st), "cda_mlp_backward")
            _check(L.cda_mlp_wgrad(x_pk, self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(), rs, chunks,
                                   self.slab.data_ptr() + k * chunks * SLAB * 4, st), "cda_mlp_wgrad")
        chunks, tiles = chunks * sub, tiles_sub * sub
        if apply:
            _check(L.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), self.slab.data_ptr(), chunks,
                                  self.bias_slab.data_ptr(), tiles, self.sums5.data_ptr(), rows * self.A, float(vf_coef), float(ent_coef), self.out6.data_ptr(),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm),
                                  self.grad.data_ptr(), self.norm2.data_ptr(), st), "cda_mlp_adam")
        return chunks, tiles

    def run(self, obs_rows, acts=None, logp_old=None, adv=None, ret=None, epochs=4, clip=0.2, vf_coef=0.5, ent_coef=0.01, lr=5e-5, betas=(0.9, 0.999), eps=1e-8,
            max_norm=0.5, perms=None, records=None):
        """obs_rows f32 [R, 168] (one row per market-step); acts = (category i32, price i32, price_offset i32, a_cont f32[.., 2]) and logp_old / adv
        / ret f32, R * A entries each, sample r * A + a belonging to row r.  adv is expected normalised.  perms: optional i64 [epochs, R]
        (tests); default a keyed permutation per epoch.  records = (rec, adv sums or None, count) replaces acts / logp_old / adv / ret (see
        minibatch_step); with the sums given the advantages are normalised inside the loss."""
        L, dev = _lib(), self.p.device
        assert obs_rows.shape == (self.R, OBS) and obs_rows.dtype == torch.float32 and obs_rows.is_contiguous()
        if records is not None:
            assert records[0].dtype == torch.float32 and records[0].is_contiguous() and records[0].numel() == self.R * self.A * 8
        else:
            for t in (*acts, logp_old, adv, ret):
                assert t.is_contiguous()
            assert acts[0].dtype == torch.int32 and acts[3].dtype == torch.float32 and adv.dtype == torch.float32
        for ep in range(epochs):
            if perms is None:                                   # a keyed bijection per epoch (one launch; torch.randperm is a device sort)
                self._epochs_done += 1
                key = (self.shuffle_seed * 0x9E3779B97F4A7C15 + self._epochs_done * 0xD1342543DE82EF95) & (2 ** 64 - 1)
                _check(L.cda_mlp_permutation(key, self.R, self.perm.data_ptr(), _stream(dev)), "cda_mlp_permutation")
            else:
                self.perm.copy_(perms[ep])
            fused = self.fused and records is not None
            if not fused:
                _check(L.cda_mlp_prep_rows(obs_rows.data_ptr(), self.perm.data_ptr(), self.R, self.x_rm.data_ptr(), self.x_pk.data_ptr(), _stream(dev)), "cda_mlp_prep_rows")
            for s in range(0, self.R, self.rows_mb):
                rows = min(self.rows_mb, self.R - s)
                self.minibatch_step(s, rows, acts, logp_old, adv, ret, clip, vf_coef, ent_coef, lr, betas, eps, max_norm, records=records,
                                    obs_rows=obs_rows if fused else None)
        return {"pg_loss": self.out6[0], "v_loss": self.out6[1], "entropy": self.out6[2]}
