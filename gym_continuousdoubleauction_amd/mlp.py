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

HID, FEAT, NOUT, N_LOGITS, BSLAB = 256, 512, 32, 24, 1056
#: floats per row of a rollout's distribution record (include/cda_mlp.h CDA_MLP_DIST_LD): 22 normalised log-probabilities | 2 means | the 2 log-stds sampled with | 2 zeros
DIST_LD = 28
#: output rows of the policy half: the 24 logits / means and (rows 25, 26) the state-dependent log-std head's offsets - zero rows in a network built without the head
LS_ROWS = (N_LOGITS + 1, N_LOGITS + 2)
POLICY_ROWS = list(range(N_LOGITS)) + list(LS_ROWS)
#: history depths the network kernels are compiled for (include/cda_mlp.h CDA_MLP_HIST + CDA_MLP_HIST_VARIANTS): 4 = the reference's n_hist, the unsuffixed entry points
HIST_VARIANTS = (1, 2, 3, 4, 6, 7, 8)          # (5: the update kernel's gather of 210-float rows spills 14 registers at that width - not built; 9 .. 16: the 128-row forward tile
#:  no longer fits the LDS - both run the PyTorch loops)


def _lib():
    from ._lib import lib
    return lib()


class Layout:
    """The constants of include/cda_mlp.h at one history depth (observation = n_hist frames of 42 floats) and the entry points compiled for it
    (`fn("cda_mlp_policy_step")` -> the library's cda_mlp_policy_step[_h<H>])."""

    def __init__(self, n_hist):
        h = int(n_hist)
        if h not in HIST_VARIANTS:
            raise ValueError(f"the network kernels are compiled for n_hist in {HIST_VARIANTS} (got {n_hist}); other depths run the PyTorch loops (ppo.train)")
        self.hist, self.OBS = h, 42 * h
        self.KX = (self.OBS + 15) // 16 * 16
        self.XT = (self.KX + 31) // 32
        self.OFF_W1 = 0
        self.OFF_B1 = FEAT * self.OBS
        self.OFF_W2 = self.OFF_B1 + FEAT
        self.OFF_B2 = self.OFF_W2 + 2 * HID * HID
        self.OFF_WO = self.OFF_B2 + FEAT
        self.OFF_BO = self.OFF_WO + NOUT * HID
        self.OFF_LS = self.OFF_BO + NOUT
        self.PARAMS = self.OFF_LS + 2
        self.WB_ELEMS = FEAT * self.KX + 2 * HID * HID + 2 * NOUT * HID + 2 * HID * HID + 2 * HID * NOUT
        self.SLAB = FEAT * 32 * self.XT + 2 * HID * HID + NOUT * FEAT
        self.suffix = "" if h == 4 else f"_h{h}"

    def fn(self, name):
        return getattr(_lib(), name + self.suffix)


_LAYOUTS = {}


def layout(n_hist=4):
    if n_hist not in _LAYOUTS:
        _LAYOUTS[n_hist] = Layout(n_hist)
    return _LAYOUTS[n_hist]


def layout_of_params(n_params):
    """the layout whose parameter vector has `n_params` entries"""
    for h in HIST_VARIANTS:
        if layout(h).PARAMS == int(n_params):
            return layout(h)
    raise ValueError(f"{n_params} parameters match no compiled history depth {HIST_VARIANTS}")


# the reference's depth as module constants (what most callers and the tests use)
_L4 = layout(4)
OBS, KX, XT = _L4.OBS, _L4.KX, _L4.XT
OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_WO, OFF_BO, OFF_LS, PARAMS = _L4.OFF_W1, _L4.OFF_B1, _L4.OFF_W2, _L4.OFF_B2, _L4.OFF_WO, _L4.OFF_BO, _L4.OFF_LS, _L4.PARAMS
WB_ELEMS, SLAB = _L4.WB_ELEMS, _L4.SLAB
assert (OBS, KX, XT, OFF_B1, OFF_W2, OFF_B2, OFF_WO, OFF_BO, OFF_LS, PARAMS, WB_ELEMS, SLAB) == (168, 176, 6, 86016, 86528, 217600, 218112, 226304, 226336, 226338, 385024, 245760)


def _check(rc, what):
    from ._lib import check
    check(rc, what)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def init_theta(obs_dim=OBS, generator=None, state_dependent_log_std=False, hidden=(HID, HID)):
    """A fresh parameter vector with nn.Linear's default initialisation per block (uniform +-1/sqrt(fan_in)), log_std = -0.5; obs_dim = 42 n_hist.
    state_dependent_log_std: rows 25, 26 of the output layer - the log-std head's offsets on top of the free vector, RLlib's default module for Box actions
    (train/policy/policy_handler.py:69-76) - are initialised like every other output row instead of zero (`has_log_std_head(theta)` tells the two apart).
    hidden = (h1, h2), each 1 .. 256: the reference's `fcnet_hiddens` (config/train_config.json:49; 256 x 256 there).  The kernels are compiled for 256-wide layers; a
    NARROWER network is the same parameter vector with the units beyond h1 / h2 of both halves dead - incoming and outgoing weights and biases exactly zero.  A dead
    unit outputs tanh(0) = 0 exactly, so every gradient that touches it is an exact zero and Adam never moves it (tests/test_hip_mlp.py): the narrow network trains
    inside the wide kernels at the wide kernels' cost.  `hidden_widths(theta)` reads the widths back."""
    if obs_dim % 42:
        raise ValueError("an observation is n_hist frames of 42 floats")
    L = layout(obs_dim // 42)
    u = lambda n, fan_in: (torch.rand(n, generator=generator) * 2 - 1) / math.sqrt(fan_in)       # noqa: E731
    th = torch.zeros(L.PARAMS)
    th[L.OFF_W1:L.OFF_B1] = u(FEAT * L.OBS, L.OBS); th[L.OFF_B1:L.OFF_W2] = u(FEAT, L.OBS)
    th[L.OFF_W2:L.OFF_B2] = u(2 * HID * HID, HID); th[L.OFF_B2:L.OFF_WO] = u(FEAT, HID)
    wo = u(NOUT * HID, HID).view(NOUT, HID); bo = u(NOUT, HID)
    dead = [r for r in range(N_LOGITS + 1, NOUT) if not (state_dependent_log_std and r in LS_ROWS)]
    wo[dead] = 0; bo[dead] = 0
    th[L.OFF_WO:L.OFF_BO] = wo.reshape(-1); th[L.OFF_BO:L.OFF_LS] = bo
    th[L.OFF_LS:] = -0.5
    h1, h2 = (int(hidden[0]), int(hidden[1])) if hasattr(hidden, "__len__") else (int(hidden), int(hidden))
    if not (1 <= h1 <= HID and 1 <= h2 <= HID):
        raise ValueError(f"hidden widths must be in 1 .. {HID} (the kernels are compiled for {HID}-wide layers; got {hidden})")
    if (h1, h2) != (HID, HID):
        # nn.Linear's bound follows the LIVE fan-in: W2 reads h1 units, Wo reads h2 (the full-width draw above is rescaled, then the dead units are cut)
        w1 = th[L.OFF_W1:L.OFF_B1].view(2, HID, L.OBS); b1 = th[L.OFF_B1:L.OFF_W2].view(2, HID)
        w2 = th[L.OFF_W2:L.OFF_B2].view(2, HID, HID); b2 = th[L.OFF_B2:L.OFF_WO].view(2, HID)
        wo_v = th[L.OFF_WO:L.OFF_BO].view(NOUT, HID); bo_v = th[L.OFF_BO:L.OFF_LS]
        w2 *= math.sqrt(HID / h1); b2 *= math.sqrt(HID / h1); wo_v *= math.sqrt(HID / h2); bo_v *= math.sqrt(HID / h2)
        w1[:, h1:] = 0; b1[:, h1:] = 0
        w2[:, h2:, :] = 0; w2[:, :, h1:] = 0; b2[:, h2:] = 0
        wo_v[:, h2:] = 0
    return th


def hidden_widths(theta):
    """(h1, h2): the live units of the two hidden layers (the same in the policy and the value half) - a unit is dead when its bias and every weight into it are zero"""
    L = layout_of_params(theta.numel())
    th = theta.detach().float().cpu()
    w1, b1 = th[L.OFF_W1:L.OFF_B1].view(2, HID, L.OBS), th[L.OFF_B1:L.OFF_W2].view(2, HID)
    w2, b2 = th[L.OFF_W2:L.OFF_B2].view(2, HID, HID), th[L.OFF_B2:L.OFF_WO].view(2, HID)
    live1 = ((w1 != 0).any(-1) | (b1 != 0)).any(0)
    live2 = ((w2 != 0).any(-1) | (b2 != 0)).any(0)
    last = lambda m: int(m.nonzero().max()) + 1 if bool(m.any()) else 0       # noqa: E731
    return last(live1), last(live2)


def has_log_std_head(theta):
    """does this parameter vector carry a state-dependent log-std head (non-zero rows 25, 26 of the output layer)?"""
    L = layout_of_params(theta.numel())
    wo = theta[L.OFF_WO:L.OFF_BO].view(NOUT, HID)
    return bool((wo[list(LS_ROWS)] != 0).any() or (theta[L.OFF_BO:L.OFF_LS][list(LS_ROWS)] != 0).any())


def theta_from_actor_critic(model):
    """ppo.ActorCritic (block matrices with masks) -> the fused layout."""
    H = model.hidden
    L = layout(model.l1.weight.shape[1] // 42)
    assert H == HID and model.l1.weight.shape[1] == L.OBS
    th = torch.zeros(L.PARAMS, dtype=torch.float32)
    with torch.no_grad():
        th[L.OFF_W1:L.OFF_B1] = model.l1.weight.detach().float().cpu().reshape(-1)
        th[L.OFF_B1:L.OFF_W2] = model.l1.bias.detach().float().cpu()
        w2 = model.l2.weight.detach().float().cpu()
        th[L.OFF_W2:L.OFF_B2] = torch.stack([w2[:H, :H], w2[H:, H:]]).reshape(-1)
        th[L.OFF_B2:L.OFF_WO] = model.l2.bias.detach().float().cpu()
        wo = model.out.weight.detach().float().cpu()
        blk = torch.zeros(NOUT, H)
        blk[:N_LOGITS] = wo[:N_LOGITS, :H]; blk[N_LOGITS] = wo[N_LOGITS, H:]
        bo = model.out.bias.detach().float().cpu().clone()
        if getattr(model, "state_dependent_log_std", False):
            blk[list(LS_ROWS)] = wo[list(LS_ROWS), :H]; bo[N_LOGITS + 3:] = 0
        else:
            bo[N_LOGITS + 1:] = 0
        th[L.OFF_WO:L.OFF_BO] = blk.reshape(-1)
        th[L.OFF_BO:L.OFF_LS] = bo
        th[L.OFF_LS:] = model.log_std.detach().float().cpu()
    return th


def actor_critic_from_theta(theta, dtype=torch.float32):
    from .ppo import ActorCritic
    th = theta.detach().float().cpu()
    L = layout_of_params(th.numel())
    sd = has_log_std_head(th)
    m = ActorCritic(L.OBS, state_dependent_log_std=sd).to(dtype)
    H = HID
    with torch.no_grad():
        m.l1.weight.copy_(th[L.OFF_W1:L.OFF_B1].view(FEAT, L.OBS)); m.l1.bias.copy_(th[L.OFF_B1:L.OFF_W2])
        w2 = th[L.OFF_W2:L.OFF_B2].view(2, H, H)
        m.l2.weight.zero_(); m.l2.weight[:H, :H] = w2[0]; m.l2.weight[H:, H:] = w2[1]; m.l2.bias.copy_(th[L.OFF_B2:L.OFF_WO])
        wo = th[L.OFF_WO:L.OFF_BO].view(NOUT, H)
        m.out.weight.zero_(); m.out.weight[:N_LOGITS, :H] = wo[:N_LOGITS]; m.out.weight[N_LOGITS, H:] = wo[N_LOGITS]
        if sd:
            m.out.weight[list(LS_ROWS), :H] = wo[list(LS_ROWS)]
        m.out.bias.copy_(th[L.OFF_BO:L.OFF_LS]); m.log_std.copy_(th[L.OFF_LS:])
    return m


def _r(t):
    """round to bfloat16 and back (what an MFMA operand sees)"""
    return t.to(torch.bfloat16).to(t.dtype)


def reference_outputs(theta, x, emulate_bf16=True, dtype=torch.float64, keep=False):
    """The network in plain PyTorch on the CPU: out [n, 32].  emulate_bf16: operands (inputs, weights, activations between layers) rounded
    to bfloat16 as the kernels do, products and sums in `dtype`.  keep: also return (xb, h1, h2) as the kernels store them."""
    th = theta.detach().cpu().to(dtype)
    x = x.detach().cpu().to(dtype)
    L = layout_of_params(th.numel())
    rd = _r if emulate_bf16 else (lambda t: t)
    W1, b1 = rd(th[L.OFF_W1:L.OFF_B1].view(FEAT, L.OBS)), th[L.OFF_B1:L.OFF_W2]
    W2, b2 = rd(th[L.OFF_W2:L.OFF_B2].view(2, HID, HID)), th[L.OFF_B2:L.OFF_WO]
    Wo, bo = rd(th[L.OFF_WO:L.OFF_BO].view(NOUT, HID)), th[L.OFF_BO:L.OFF_LS]
    xb = rd(x)
    h1 = rd(torch.tanh(xb @ W1.t() + b1))
    h2 = torch.cat([rd(torch.tanh(h1[:, :HID] @ W2[0].t() + b2[:HID])), rd(torch.tanh(h1[:, HID:] @ W2[1].t() + b2[HID:]))], dim=1)
    out = torch.zeros(x.shape[0], NOUT, dtype=dtype)
    out[:, POLICY_ROWS] = h2[:, :HID] @ Wo[POLICY_ROWS].t() + bo[POLICY_ROWS]          # (rows 25, 26: the log-std head's offsets; zero rows without the head)
    out[:, N_LOGITS] = h2[:, HID:] @ Wo[N_LOGITS] + bo[N_LOGITS]
    return (out, xb, h1, h2) if keep else out


def reference_gradients(theta, xb, h1, h2, d_out, dtype=torch.float64):
    """The back-propagation the kernels perform, in plain PyTorch: gradient of theta (dense vector, log_std entries zero) for given
    d_out [n, 32], with the kernels' roundings (d_out, dz2, dz1 rounded to bfloat16 where they become operands)."""
    th = theta.detach().cpu().to(dtype)
    L = layout_of_params(th.numel())
    OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_WO, OFF_BO, OFF_LS, PARAMS = L.OFF_W1, L.OFF_B1, L.OFF_W2, L.OFF_B2, L.OFF_WO, L.OFF_BO, L.OFF_LS, L.PARAMS      # (this depth's, not the module's)
    W2 = _r(th[OFF_W2:OFF_B2].view(2, HID, HID)); Wo = _r(th[OFF_WO:OFF_BO].view(NOUT, HID))
    d_out = d_out.detach().cpu().to(dtype)
    dob = _r(d_out)
    dh2 = torch.cat([dob[:, POLICY_ROWS] @ Wo[POLICY_ROWS], dob[:, N_LOGITS:N_LOGITS + 1] @ Wo[N_LOGITS:N_LOGITS + 1]], dim=1)
    dz2 = _r(dh2 * (1 - h2 * h2))
    dh1 = torch.cat([dz2[:, :HID] @ W2[0], dz2[:, HID:] @ W2[1]], dim=1)
    dz1 = _r(dh1 * (1 - h1 * h1))
    g = torch.zeros(PARAMS, dtype=dtype)
    g[OFF_W1:OFF_B1] = (dz1.t() @ xb).reshape(-1); g[OFF_B1:OFF_W2] = dz1.sum(0)
    g[OFF_W2:OFF_B2] = torch.stack([dz2[:, :HID].t() @ h1[:, :HID], dz2[:, HID:].t() @ h1[:, HID:]]).reshape(-1); g[OFF_B2:OFF_WO] = dz2.sum(0)
    gwo = torch.zeros(NOUT, HID, dtype=dtype)
    gwo[POLICY_ROWS] = dob[:, POLICY_ROWS].t() @ h2[:, :HID]; gwo[N_LOGITS] = dob[:, N_LOGITS] @ h2[:, HID:]
    g[OFF_WO:OFF_BO] = gwo.reshape(-1)
    gbo = torch.zeros(NOUT, dtype=dtype); gbo[:N_LOGITS + 3] = d_out[:, :N_LOGITS + 3].sum(0)
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
    """theta (f32 master copy), Adam state and the bf16 operand blob on one HIP device.  storage = (theta row, wb row): views into a PolicyBank's
    banks instead of tensors of its own (the league's kernels address a net as a row of the banks)."""

    def __init__(self, device, theta=None, seed=0, storage=None, n_hist=None, state_dependent_log_std=None, hidden=(HID, HID)):
        """n_hist: the history depth of the observations (default: the depth `theta` was laid out for, else the reference's 4).
        state_dependent_log_std: RLlib's default head for Box actions - the policy network emits two log-std offsets per row (output rows 25, 26) on top of the free
        log_std vector, and the update trains them (FusedUpdate reads this attribute); default: what `theta` carries (has_log_std_head), False for a fresh network.
        hidden: the widths of a FRESH network's two hidden layers, <= 256 each (init_theta: the reference's `fcnet_hiddens`)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FusedPolicy needs a HIP device; the PyTorch statement of the network is ppo.ActorCritic")
        if theta is None:
            g = torch.Generator().manual_seed(int(seed))
            theta = init_theta(42 * int(n_hist or 4), generator=g, state_dependent_log_std=bool(state_dependent_log_std), hidden=hidden)
        self.L = L = layout_of_params(theta.numel())
        self.state_dependent_log_std = has_log_std_head(theta) if state_dependent_log_std is None else bool(state_dependent_log_std)
        if n_hist is not None and int(n_hist) != L.hist:
            raise ValueError(f"theta is laid out for n_hist = {L.hist}, not {n_hist}")
        if storage is None:
            self.theta = theta.detach().float().to(self.device).contiguous()
            self.wb = torch.zeros(L.WB_ELEMS, dtype=torch.bfloat16, device=self.device)
        else:
            self.theta, self.wb = storage
            assert self.theta.shape == (L.PARAMS,) and self.theta.is_contiguous() and self.wb.shape == (L.WB_ELEMS,) and self.wb.is_contiguous()
            self.theta.copy_(theta.detach().float())
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_step = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.pack()

    @classmethod
    def from_actor_critic(cls, model, device):
        return cls(device, theta=theta_from_actor_critic(model))

    def to_actor_critic(self, dtype=torch.float32):
        return actor_critic_from_theta(self.theta, dtype)

    @property
    def log_std(self):
        return self.theta[self.L.OFF_LS:]

    def pack(self):
        _check(self.L.fn("cda_mlp_pack")(self.theta.data_ptr(), self.wb.data_ptr(), _stream(self.device)), "cda_mlp_pack")

    def forward(self, obs, first_row=0, n_rows=None, out=None):
        """network outputs f32 [rows, 32] (columns 0..23 policy, 24 value) for rows [first_row, first_row + n_rows) of obs f32[*, 168]"""
        obs = obs.contiguous()
        assert obs.dtype == torch.float32 and obs.shape[-1] == self.L.OBS and obs.device == self.device
        n_rows = obs.shape[0] - first_row if n_rows is None else n_rows
        if out is None:
            out = torch.zeros((obs.shape[0], NOUT), dtype=torch.float32, device=self.device)
        _check(self.L.fn("cda_mlp_forward")(self.wb.data_ptr(), self.theta.data_ptr(), obs.data_ptr(), int(first_row), int(n_rows), out.data_ptr(),
                                            _stream(self.device)), "cda_mlp_forward")
        return out

    def policy_step(self, obs, num_agents, seed, counter, draw, first_market=0, n_markets=None, outs=None):
        """one launch: forward + sampling for the markets [first_market, first_market + n_markets).  Returns the dict of output tensors
        ([N, A] each; a_cont [N, A, 2]; value [N])."""
        obs = obs.contiguous()
        assert obs.dtype == torch.float32 and obs.shape[-1] == self.L.OBS
        N = obs.shape[0]
        n_markets = N - first_market if n_markets is None else n_markets
        A, dev = int(num_agents), self.device
        if outs is None:
            e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)         # noqa: E731
            outs = {"category": e((N, A), torch.int32), "size_mean": e((N, A), torch.float32), "size_sigma": e((N, A), torch.float32),
                    "price": e((N, A), torch.int32), "price_offset": e((N, A), torch.int32), "a_cont": e((N, A, 2), torch.float32),
                    "logp": e((N, A), torch.float32), "value": e((N,), torch.float32)}
        _check(self.L.fn("cda_mlp_policy_step")(self.wb.data_ptr(), self.theta.data_ptr(), obs.data_ptr(), int(first_market), int(n_markets), A,
                                          int(seed) & (2 ** 64 - 1), counter.data_ptr(), int(draw),
                                          *[outs[k].data_ptr() for k in ("category", "size_mean", "size_sigma", "price", "price_offset", "a_cont", "logp", "value")],
                                          _stream(dev)), "cda_mlp_policy_step")
        return outs


LEAGUE_RANDOM, LEAGUE_MAX_NETS = -1, 16


class PolicyBank:
    """The league's networks as rows of two banks (include/cda_mlp.h `cda_league`): theta f32 [n_max, PARAMS], wb bf16 [n_max, WB_ELEMS]; rows
    0 .. n_trainable - 1 are the trainable policies (FusedPolicy objects whose parameters ARE those rows: an optimiser step is seen by the next
    rollout with no copy), the rows behind them frozen snapshots (champions: league_based_self_play_callback.py:938-1170).  slot_net i32 [N, A] names
    the row that plays each (market, slot), LEAGUE_RANDOM = the uniform random module."""

    def __init__(self, device, n_markets, num_agents, n_trainable, max_frozen=8, seed=0, random_seed=0, n_hist=4, state_dependent_log_std=False, hidden=(HID, HID)):
        from ._lib import League
        self.device = torch.device(device)
        self.L = L = layout(n_hist)
        PARAMS, WB_ELEMS = L.PARAMS, L.WB_ELEMS
        self.n_trainable, self.max_frozen, self.n_frozen = int(n_trainable), int(max_frozen), 0
        if not 1 <= self.n_trainable <= num_agents or self.n_trainable + self.max_frozen > LEAGUE_MAX_NETS:
            raise ValueError(f"need 1 <= n_trainable <= num_agents and n_trainable + max_frozen <= {LEAGUE_MAX_NETS}")
        n_max = self.n_trainable + self.max_frozen
        self.theta = torch.zeros((n_max, PARAMS), dtype=torch.float32, device=self.device)
        self.wb = torch.zeros((n_max, WB_ELEMS), dtype=torch.bfloat16, device=self.device)
        self.policies = [FusedPolicy(self.device, seed=seed + 7919 * p, storage=(self.theta[p], self.wb[p]), n_hist=L.hist, state_dependent_log_std=state_dependent_log_std, hidden=hidden)
                         for p in range(self.n_trainable)]
        self.slot_net = torch.full((int(n_markets), int(num_agents)), LEAGUE_RANDOM, dtype=torch.int32, device=self.device)
        self.slot_net[:, :self.n_trainable] = torch.arange(self.n_trainable, dtype=torch.int32, device=self.device)
        self.random_seed = int(random_seed) & (2 ** 64 - 1)
        self._struct = League()
        self._refresh()

    def _refresh(self):
        s = self._struct
        s.wb_bank, s.theta_bank, s.slot_net = self.wb.data_ptr(), self.theta.data_ptr(), self.slot_net.data_ptr()
        # n_nets = the bank's CAPACITY, not the rows in use: a frozen row nobody plays costs a workgroup that reads its tile's slot_net and leaves, and the
        # launch grid (baked into a captured rollout graph) never changes when a champion joins - no re-capture (1.3 ms per promotion at 2048 x 8)
        s.n_nets, s.n_trainable, s.random_seed = self.n_trainable + self.max_frozen, self.n_trainable, self.random_seed

    @property
    def n_nets(self):
        return self.n_trainable + self.n_frozen

    def struct(self):
        return self._struct

    def snapshot(self, source, frozen_slot=None):
        """freeze a copy of bank row `source` (a trainable net) as frozen net `frozen_slot` (default: the next free one); returns its bank row.
        Device-side copies on the current stream: no host sync, and a captured rollout graph stays valid (it addresses the banks by row; see _refresh)."""
        if frozen_slot is None:
            if self.n_frozen >= self.max_frozen:
                raise ValueError("the bank is full: overwrite a slot (frozen_slot=...)")
            frozen_slot = self.n_frozen
            self.n_frozen += 1
        row = self.n_trainable + int(frozen_slot)
        self.theta[row].copy_(self.theta[int(source)])
        self.wb[row].copy_(self.wb[int(source)])
        self._refresh()
        return row

    def set_slots(self, slot_net):
        """slot_net: [N, A] integers (bank rows, LEAGUE_RANDOM for the random module); copied into the resident tensor (same address: graphs stay valid)"""
        t = torch.as_tensor(slot_net).to(dtype=torch.int32)
        if t.shape != self.slot_net.shape:
            raise ValueError(f"need shape {tuple(self.slot_net.shape)}")
        if int(t.max()) >= self.n_nets or int(t.min()) < LEAGUE_RANDOM:
            raise ValueError("slot_net names a bank row that does not exist")
        self.slot_net.copy_(t.to(self.device), non_blocking=True)


class RolloutChains:
    """Whole rollouts of a CDAVecEnv (auto_reset on) under a FusedPolicy as G independent chains: chain g = the markets of group g, on
    its own stream: for every step {policy forward + sampling -> env step -> auto reset}, then the bootstrap value of the last
    observation.  Weights are frozen during a rollout and markets never interact, so no chain ever waits for another; the caller's
    stream forks into the chains before the first step and joins them after the last.  Buffers are [T(+1), N, ...] and every step's
    kernels read / write their own slot directly - nothing is copied between steps.  Each chain's launch sequence is captured into a HIP
    graph once and replayed (use_graphs), or enqueued by one native call per chain (cda_mlp_rollout_chain).

    policy: a FusedPolicy (one shared policy plays every slot) or a PolicyBank (league self-play: per-slot modules, value / dist per trainable net).
    with_dist: also record the rollout policy's distribution per market-step (the update's KL term).
    capture_ends: episode-end capture (include/cda.h cda_step_range_capture): the last observation of every episode that ends inside the rollout is kept
        (fin_obs / fin_index) - gae() then bootstraps time-limit truncations with V(that observation), as RLlib does.
    info_markets = S > 0: the LAST S markets are a chain of their own that also writes the info tensors of every step into [T, S, ...] buffers
        (`info`: what BatchedEpisodeRecorder.record_rollout reads; the reference records one episode in N, train/episode_record.py:197)."""

    def __init__(self, env, policy, horizon, groups=4, seed=0, use_graphs=True, with_dist=False, capture_ends=False, info_markets=0):
        from ._lib import RolloutBufs
        from . import _capi as K
        self.env, self.policy, self.T = env, policy, int(horizon)
        self.bank = policy if isinstance(policy, PolicyBank) else None
        N, A, dev, T = env.n_markets, env.num_agents, env.device, int(horizon)
        self.L = L = policy.L
        if env.obs_dim != L.OBS:
            raise ValueError(f"the policy is laid out for {L.OBS}-float observations (n_hist = {L.hist}), the env emits {env.obs_dim}")
        OBS = L.OBS
        if not bool(env.config.get("auto_reset", False)):
            raise ValueError("RolloutChains needs an auto_reset env (episode ends are handled on the device)")
        self.N, self.A, self.device = N, A, dev
        kn = self.bank.n_trainable if self.bank else 1
        self.n_value_nets = kn
        e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)             # noqa: E731
        vshape = (kn, T + 1, N) if self.bank else (T + 1, N)
        self.buf = {"obs": e((T + 1, N, OBS), torch.float32), "category": e((T, N, A), torch.int32), "size_mean": e((T, N, A), torch.float32),
                    "size_sigma": e((T, N, A), torch.float32), "price": e((T, N, A), torch.int32), "price_offset": e((T, N, A), torch.int32),
                    "a_cont": e((T, N, A, 2), torch.float32), "logp": e((T, N, A), torch.float32), "value": e(vshape, torch.float32),
                    "reward": e((T, N, A), torch.float64), "terminated": e((T, N), torch.uint8), "truncated": e((T, N), torch.uint8),
                    "record": e((T, N, A, 8), torch.float32)}          # include/cda_mlp.h CDA_REC_*: what the update's loss reads, one line per row
        ptrs = {k: v.data_ptr() for k, v in self.buf.items()}
        if with_dist:
            self.buf["dist"] = e((kn, T, N, DIST_LD) if self.bank else (T, N, DIST_LD), torch.float32)
            self.log_std_old = e((kn, 2), torch.float32)               # the rollout policy's log_std (the update moves theta's)
            ptrs["dist"] = self.buf["dist"].data_ptr()
        self.with_dist = bool(with_dist)
        self.capture_ends = bool(capture_ends)
        if self.capture_ends:
            cap = N * (T // max(1, int(env.max_step)) + 1)
            self.fin_cap = cap
            self.buf["fin_index"], self.buf["fin_obs"], self.buf["fin_count"] = torch.full((T, N), -1, dtype=torch.int32, device=dev), e((cap, OBS), torch.float32), e((1,), torch.int32)
            self.fin_value = e((kn, cap), torch.float32)
            ptrs.update(fin_index=self.buf["fin_index"].data_ptr(), fin_obs=self.buf["fin_obs"].data_ptr(), fin_count=self.buf["fin_count"].data_ptr(), fin_cap=cap)
        self.adv_stats = torch.zeros((kn, 2), dtype=torch.float64, device=dev)
        self.seed = int(seed) & (2 ** 64 - 1)
        # the chains: G groups over the info-less markets (+ one chain of the sampled markets with info tensors)
        S = max(0, min(int(info_markets), N - 1))
        self.info_markets, n_plain = S, N - S
        G = max(1, min(int(groups), n_plain))
        if S and G >= 4:
            G = 3                                                  # (the GPU runs four hardware queues by default: the sampled chain takes the fourth, streams.py)
        self.ranges = []
        for g in range(G):
            first, cnt = C.c_int32(), C.c_int32()
            _lib().cda_group_range(n_plain, G, g, C.byref(first), C.byref(cnt))
            self.ranges.append((first.value, cnt.value))
        self._cbufs = [RolloutBufs(**ptrs) for _ in range(G)]
        self.info = None
        if S:
            self.ranges.append((n_plain, S))
            self.info, self._info_steps = {}, (K.InfoPtrs * T)()
            for name, ct, per_agent, dims in K.INFO_FIELDS:
                shape = ((S, A) if per_agent else (S,)) + tuple(dims)
                shape = shape + (16,) if ct is K.Dec else shape
                dt = torch.uint8 if ct is K.Dec else {C.c_int32: torch.int32, C.c_double: torch.float64, C.c_uint8: torch.uint8}[ct]
                t = e((T,) + shape, dt)
                self.info[name] = t
                per_market = t[0, 0].numel() * t.element_size()                  # the kernel indexes by GLOBAL market: shift the base so that market n_plain lands on row 0
                for step in range(T):
                    setattr(self._info_steps[step], name, t[step].data_ptr() - n_plain * per_market)
            cb = RolloutBufs(**ptrs)
            cb.info_steps = C.cast(self._info_steps, C.c_void_p)
            self._cbufs.append(cb)
        G = len(self.ranges)
        # the rollout counter (fresh draws per rollout): ONE PER CHAIN, all equal, each bumped by its own chain as the chain's last launch (cda_rollout_bufs.counter_bump) -
        # a shared counter had to be incremented by a launch AHEAD of the fork: 20-30 us on the critical path of every rollout (tools/policy_leg_probe.py)
        self._counters = torch.ones(G, dtype=torch.int64, device=dev)
        for g in range(G):
            self._cbufs[g].counter_bump = self._counters[g:].data_ptr()
        if G > 1:
            from .streams import concurrent_streams
            self.streams = list(concurrent_streams(dev, G))
            self._capture_stream = None
        else:
            # one chain: it runs on the CALLER's stream, whatever that is at run() - no fork, no join, one graph launch per rollout; the capture needs a
            # stream of its own (nothing is captured on the default stream)
            self.streams = [None]
            self._capture_stream = torch.cuda.Stream(dev) if use_graphs else None
        self._fork = torch.cuda.Event()
        self._joins = [torch.cuda.Event() for _ in range(G)]
        self.graphs = None
        self._have_obs, self._env_epoch = False, 0
        self.use_graphs = bool(use_graphs)
        self.join_mode = "events"

    @property
    def counter(self):
        """the rollout counter the LAST run() sampled with (i64 [1], a fresh device tensor): what cda_mlp_policy_step needs to reproduce its draws"""
        return self._counters[:1] - 1

    def _enqueue(self, g, copy_first_obs):
        first, cnt = self.ranges[g]
        st = torch.cuda.current_stream(self.device).cuda_stream
        ctr = self._counters[g:].data_ptr()
        with torch.cuda.device(self.device):
            if self.bank is not None:
                _check(self.L.fn("cda_mlp_league_rollout_chain")(self.env._h, C.byref(self.bank.struct()), first, cnt, self.T, self.seed, ctr,
                                                                 C.byref(self._cbufs[g]), int(copy_first_obs), st), "cda_mlp_league_rollout_chain")
            else:
                _check(self.L.fn("cda_mlp_rollout_chain")(self.env._h, self.policy.wb.data_ptr(), self.policy.theta.data_ptr(), first, cnt, self.T, self.seed,
                                                          ctr, C.byref(self._cbufs[g]), int(copy_first_obs), st), "cda_mlp_rollout_chain")

    def _capture(self):
        graphs = []
        for g, s in enumerate(self.streams):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s if s is not None else self._capture_stream):
                self._enqueue(g, True)
            graphs.append(gr)
        self.graphs = graphs

    def run(self):
        """one rollout of `horizon` steps; returns the buffer dict (views stay valid; the next run() overwrites them)"""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if not self._have_obs or self._env_epoch != getattr(self.env, "host_epoch", 0):
            # the very first rollout - and the first one after the markets were reset / stepped from the HOST (env.host_epoch) - starts from the env's own
            # observation tensor; every other rollout continues from its predecessor's last observation (the chains do not write env.obs)
            self.env.join()
            self.buf["obs"][self.T].copy_(self.env.obs)
            self._have_obs, self._env_epoch = True, getattr(self.env, "host_epoch", 0)
        if self.capture_ends:
            self.buf["fin_count"].zero_(); self.buf["fin_index"].fill_(-1)
        if self.with_dist:                                        # the log_std the rollout samples with (a row per trainable net)
            src = self.bank.theta[:self.bank.n_trainable, self.L.OFF_LS:] if self.bank else self.policy.theta[self.L.OFF_LS:].view(1, 2)
            self.log_std_old.copy_(src)
        if self.use_graphs and self.graphs is None and len(self.streams) >= 1:
            torch.cuda.synchronize(dev)
            try:
                self._capture()
            except Exception as ex:  # noqa: BLE001 - the native loop is always available
                import warnings
                warnings.warn(f"HIP graph capture of the rollout chains failed ({ex}); using direct launches")
                self.graphs, self.use_graphs = None, False
        if len(self.streams) == 1 and self.streams[0] is None:    # one chain: on the caller's stream, in order with everything around it
            if self.graphs is not None:
                self.graphs[0].replay()
            else:
                self._enqueue(0, True)
            return self.buf
        # the fork: the chains start behind everything the caller's stream holds - nothing to wait for when that stream is idle (every earlier launch of the
        # caller has then completed; the host-side query costs a microsecond, an event edge per chain ~10)
        fork = not cur.query()
        if fork:
            self._fork.record(cur)
        for g, s in enumerate(self.streams):
            if fork and s.cuda_stream != cur.cuda_stream:
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
        """advantages and returns of the last run() straight into the sample records (one launch; ppo.gae's recursion; with capture_ends a time-limit
        truncation bootstraps with the value of the episode's captured last observation - one more value launch over the captured list).  Returns
        (records [T * N, A, 8], the sums the update normalises the advantages with, their count); league: the sums are [n_trainable, 2] and the count is
        per trainable net (T * N: one slot each)."""
        L, st = _lib(), _stream(self.device)
        k = self.bank.n_trainable if self.bank else 0
        fin_index = fin_value = None
        if self.capture_ends:
            wb, th = (self.bank.wb, self.bank.theta) if self.bank else (self.policy.wb, self.policy.theta)
            # (bounded by the device-side count: an iteration in which no episode ended costs an empty launch, not a forward pass over the list's capacity)
            _check(self.L.fn("cda_mlp_values_counted")(wb.data_ptr(), th.data_ptr(), max(k, 1), self.buf["fin_obs"].data_ptr(), self.fin_cap, self.buf["fin_count"].data_ptr(),
                                                       self.fin_value.data_ptr(), self.fin_cap, st), "cda_mlp_values_counted")
            fin_index, fin_value = self.buf["fin_index"].data_ptr(), self.fin_value.data_ptr()
        _check(L.cda_gae_records_bootstrap(self.buf["reward"].data_ptr(), self.buf["value"].data_ptr(), self.buf["terminated"].data_ptr(), self.buf["truncated"].data_ptr(),
                                           self.T, self.N, self.A, k, float(reward_scale), float(gamma), float(lam), fin_index, fin_value, self.fin_cap if self.capture_ends else 0,
                                           self.buf["record"].data_ptr(), self.adv_stats.data_ptr(), st), "cda_gae_records_bootstrap")
        if self.bank:
            return self.buf["record"].view(self.T * self.N, self.A, 8), self.adv_stats, self.T * self.N
        return self.buf["record"].view(self.T * self.N, self.A, 8), self.adv_stats[0], self.T * self.N * self.A


    def check_capture_overflow(self):
        """after a rollout (synchronising read): did more episodes end than the capture list holds?  The consumer then bootstrapped the overflowed truncations with 0;
        warn - silently it would bias the value targets.  Returns the number of ends that found no slot."""
        if not self.capture_ends:
            return 0
        lost = int(self.buf["fin_count"].item()) - self.fin_cap
        if lost > 0:
            import warnings
            warnings.warn(f"episode-end capture overflowed: {lost} of {lost + self.fin_cap} episode ends of this rollout found no slot (their time-limit bootstrap fell back to 0)")
        return max(0, lost)


class EpisodeReturns:
    """Returns of COMPLETED episodes, accumulated on the device from rollout to rollout (cda_episode_returns): what a learning curve is drawn from when the
    horizon is shorter than an episode - the mean reward of a rollout slice depends on which part of the episodes the slice happens to cover."""

    def __init__(self, n_markets, num_agents, device, per_slot=False):
        self.N, self.A = int(n_markets), int(num_agents)
        self.running = torch.zeros((self.N, self.A), dtype=torch.float64, device=device)
        self.acc = torch.zeros((2, self.A), dtype=torch.float64, device=device)          # [0] sum of completed episodes' returns per slot, [1] their number
        self.per_slot = torch.zeros((self.N, self.A, 2), dtype=torch.float64, device=device) if per_slot else None   # (sum, number) per (market, slot), this rollout
        self.device = torch.device(device)

    def update(self, buf, horizon):
        self.acc.zero_()
        _check(_lib().cda_episode_returns(buf["reward"].data_ptr(), buf["terminated"].data_ptr(), buf["truncated"].data_ptr(), int(horizon), self.N, self.A,
                                          self.running.data_ptr(), self.acc[0].data_ptr(), self.acc[1].data_ptr(),
                                          self.per_slot.data_ptr() if self.per_slot is not None else None, _stream(self.device)), "cda_episode_returns")
        return self.acc                                                                   # (device tensor: read it with the iteration's other statistics)


class _Fns:
    """attribute access -> the entry point compiled for a layout's history depth (`_Fns(L).cda_mlp_wgrad` = the library's cda_mlp_wgrad[_h<H>])"""

    def __init__(self, L):
        self._L = L

    def __getattr__(self, name):
        return self._L.fn(name)


class FusedUpdate:
    """The PPO update on the kernels of include/cda_mlp.h: per epoch a keyed permutation of the R unique observation rows, per minibatch
    {gather + forward + loss + back-propagation (one launch), weight gradients, reduce, clip + Adam}: four launches, no autograd, no GEMM
    library.  fused=False: the same step as separate kernels (a gather / convert pass per epoch, then forward, loss, backward, ...)."""

    def __init__(self, policy, n_rows, rows_mb, num_agents, chunks=None, sub_batches=1, fused=True, allreduce=None, world=1):
        """fused (default; used when run() is given sample records): gather, forward, loss and back-propagation of a minibatch are ONE launch
        (cda_mlp_forward_backward) - a minibatch step is {that, weight gradients, reduce, Adam}; otherwise the separate kernels run (prep_rows per
        epoch, forward, loss, backward).
        sub_batches > 1 (separate kernels only): a minibatch step runs {forward, loss, backward, weight gradients} once per sub-batch of rows_mb / sub_batches rows and
        reduces all their partial sums in one optimiser step - the same step, with activations of a sub-batch small enough to stay in the
        256-MB Infinity Cache between the kernel that writes them and the ones that read them.
        allreduce (fused path; a data-parallel learner, one process per GPU each with its own shard of markets): callable(tensor) summing it in place over
        the `world` ranks - called on the gradient (0.9 MB) between the reduce and the optimiser launches of every minibatch step, the only collective of
        the loop; every rank's loss is normalised with the GLOBAL minibatch (rows x world) so the sum is the global gradient."""
        self.p, self.R, self.rows_mb, self.A = policy, int(n_rows), int(rows_mb), int(num_agents)
        self.sub = max(1, int(sub_batches))
        self.fused = bool(fused) and self.sub == 1
        self.allreduce, self.world = allreduce, max(1, int(world))
        if self.R % 32 or self.rows_mb % 32 or self.rows_mb > self.R:
            raise ValueError("rows and minibatch rows must be multiples of 32")
        dev = policy.device
        self.L = Lo = policy.L
        KX, XT, SLAB, PARAMS = Lo.KX, Lo.XT, Lo.SLAB, Lo.PARAMS          # (this policy's history depth)
        self.tile_rows = int(Lo.fn("cda_mlp_tile_rows")())
        self.n_tiles = (self.rows_mb + self.tile_rows - 1) // self.tile_rows
        pad = self.n_tiles * self.tile_rows                            # the kernels write whole workgroup tiles
        pad = ((pad + 63) // 64) * 64                                  # (the fused kernel's are 64 rows whatever CDA_MLP_MT says)
        jobs = int(Lo.fn("cda_mlp_wgrad_jobs")())
        self.chunks = int(chunks) if chunks else max(1, min(255 // jobs, self.rows_mb // 512))      # n_hist 4: 5 jobs x 51 chunks = 255 workgroups: one wave of the 256 CUs
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)                      # noqa: E731
        self.x_rm, self.x_pk = e(self.R * KX, bf), e(self.R * 32 * XT, bf)         # (the separate kernels' per-epoch images of the shuffled rows)
        self.h1p, self.h2p, self.dz1p, self.dz2p = (e(pad * FEAT, bf) for _ in range(4))
        self.doutp = e(pad * NOUT, bf)
        self.out, self.d_out = e(pad * NOUT, f32).view(-1, NOUT), e(pad * NOUT, f32).view(-1, NOUT)
        self.slab, self.bias_slab = e(self.sub * self.chunks * SLAB, f32), e((max(self.n_tiles, pad // 64) + self.sub) * BSLAB, f32)
        # the gradient and, right behind it, the step's loss statistics (out6: CDA_LOSS_OUT_WORDS): ONE buffer, so that a data-parallel learner's all-reduce sums both - every
        # rank's out6 holds its share of the GLOBAL means (its sums over loss_samples = the global minibatch), their sum is the global mean: pg / value loss, entropy and
        # the KL that ppo.adapt_kl_coef steers kl_coef with are then the same numbers on every rank (round-5 ADVICE: per-rank partials made the ranks' objectives diverge)
        self._grad_stats = e(PARAMS + 8, f32)
        self.grad, self.out6 = self._grad_stats[:PARAMS], self._grad_stats[PARAMS:]
        self.norm2 = e(512, torch.float64)                                  # (norm2[2] = the squared gradient norm of the last step)
        self.sums5 = e(64 * 8, torch.float64)                               # CDA_MLP_LOSS_SLOTS x 8: the loss sums (slot 0, words 0..4 for the separate loss kernels)
        self.perm = torch.zeros(self.R, dtype=torch.int64, device=dev)
        pad64 = ((self.rows_mb + 63) // 64) * 64
        self.x_pk_mb = e(pad64 * 32 * XT, bf) if self.fused else None         # the fused kernel's packed image of the minibatch's observations
        self.shuffle_seed, self._epochs_done = 0x5DEECE66D, 0
        self._extra = None
        self.set_extra()                                                    # (a policy with the state-dependent log-std head trains it from the first step)

    def set_extra(self, rec_stride=0, kl_coef=0.0, vf_clip=0.0, dist_old=None, log_std_old=None, sd_log_std=None):
        """what cda_ppo_extra carries (fused path): the record stride of a league update, RLlib's KL penalty and value-error clamp, and whether the state-dependent
        log-std head trains (default: the policy's own `state_dependent_log_std`).  dist_old: rows of DIST_LD floats (RolloutChains.buf["dist"]); log_std_old is not
        read any more (the rows carry the log-stds they were sampled with)."""
        from ._lib import PpoExtra
        sd = bool(getattr(self.p, "state_dependent_log_std", False)) if sd_log_std is None else bool(sd_log_std)
        if sd and not self.fused:
            raise ValueError("the state-dependent log-std head trains on the fused update kernel only (FusedUpdate(fused=True))")
        if not (rec_stride or kl_coef or vf_clip or sd):
            self._extra, self._kl = None, 0.0
            return
        x = PpoExtra()
        x.rec_stride, x.kl_coef, x.vf_clip, x.sd_log_std = int(rec_stride), float(kl_coef), float(vf_clip), int(sd)
        x.dist_old = dist_old.data_ptr() if (dist_old is not None and kl_coef) else None
        x.log_std_old = log_std_old.data_ptr() if (log_std_old is not None and kl_coef) else None
        self._extra, self._kl, self._extra_keep = x, float(kl_coef), (dist_old, log_std_old)

    def minibatch_step(self, s, rows, acts, logp_old, adv, ret, clip, vf_coef, ent_coef, lr, betas, eps, max_norm, apply=True, records=None, obs_rows=None,
                       debug_outputs=False):
        """rows [s, s + rows) of the prepared (shuffled) observations: one optimiser step.  obs_rows (with records, fused=True): the unshuffled f32
        observation rows - the fused kernel gathers rows perm[s .. s + rows) itself, no prepared images needed.  records = (rec f32 [R, A, 8] or its
        address, adv sums f64[2] or None, their count): the loss reads sample records (RolloutChains.gae) instead of the seven per-sample arrays."""
        p, dev = self.p, self.p.device
        L = _Fns(self.L)
        st = _stream(dev)
        ptr = lambda x: x if isinstance(x, int) else x.data_ptr()           # noqa: E731
        if obs_rows is not None and records is not None and self.fused:
            rec, stats, count = records
            chunks, tiles = max(1, min(self.chunks, rows // 32)), (rows + 63) // 64
            kl = getattr(self, "_kl", 0.0)
            _check(L.cda_mlp_forward_backward(p.wb.data_ptr(), p.theta.data_ptr(), obs_rows.data_ptr(), self.perm.data_ptr() + s * 8, rows, rows * self.world, ptr(rec),
                                              stats.data_ptr() if stats is not None else None, int(count), self.A, float(clip), float(vf_coef), float(ent_coef),
                                              C.byref(self._extra) if self._extra is not None else None,
                                              self.x_pk_mb.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(),
                                              self.bias_slab.data_ptr(), self.sums5.data_ptr(), self.out6.data_ptr(), 0 if apply else 1, 0 if apply else 1,
                                              self.out.data_ptr() if debug_outputs else None, self.d_out.data_ptr() if debug_outputs else None, st), "cda_mlp_forward_backward")
            _check(L.cda_mlp_wgrad(self.x_pk_mb.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(), rows, chunks,
                                   self.slab.data_ptr(), st), "cda_mlp_wgrad")
            if apply and self.allreduce is None:
                _check(L.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), self.slab.data_ptr(), chunks,
                                      self.bias_slab.data_ptr(), tiles, self.sums5.data_ptr(), rows * self.A, float(vf_coef), float(ent_coef), kl, self.out6.data_ptr(),
                                      float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm),
                                      self.grad.data_ptr(), self.norm2.data_ptr(), st), "cda_mlp_adam")
            elif apply:
                # data parallel: this rank's share of the gradient (its loss normalised by the global minibatch), summed over the ranks, then the same step everywhere
                _check(L.cda_mlp_reduce(self.slab.data_ptr(), chunks, self.bias_slab.data_ptr(), tiles, self.sums5.data_ptr(), rows * self.A * self.world, float(vf_coef), float(ent_coef), kl,
                                        self.out6.data_ptr(), p.adam_step.data_ptr(), self.grad.data_ptr(), self.norm2.data_ptr(), st), "cda_mlp_reduce")
                self.allreduce(self._grad_stats)                                  # gradient | loss statistics: one collective
                _check(L.cda_mlp_apply(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), self.grad.data_ptr(), 1,
                                       float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm), self.norm2.data_ptr(), st), "cda_mlp_apply")
            return chunks, tiles
        sub = self.sub if (rows % (32 * self.sub) == 0 and rows // self.sub >= 32) else 1
        rs = rows // sub                                           # rows per sub-batch
        tiles_sub = (rs + self.tile_rows - 1) // self.tile_rows
        chunks = max(1, min(self.chunks, rs // 32))
        for k in range(sub):
            o = s + k * rs
            x_rm = self.x_rm.data_ptr() + o * self.L.KX * 2
            x_pk = self.x_pk.data_ptr() + o * 32 * self.L.XT * 2
            _check(L.cda_mlp_forward_train(p.wb.data_ptr(), p.theta.data_ptr(), x_rm, rs, self.h1p.data_ptr(), self.h2p.data_ptr(), self.out.data_ptr(), st), "cda_mlp_forward_train")
            last = k == sub - 1
            if records is not None:
                rec, stats, count = records
                _check(L.cda_ppo_loss_records(self.out.data_ptr(), p.theta.data_ptr() + self.L.OFF_LS * 4, ptr(rec), stats.data_ptr() if stats is not None else None, int(count),
                                              self.perm.data_ptr() + o * 8, rs, self.A, NOUT, float(clip), float(vf_coef), float(ent_coef), self.d_out.data_ptr(),
                                              self.sums5.data_ptr(), self.out6.data_ptr(), rows, 0 if (apply or k) else 1, 0 if (apply or not last) else 1, st),
                       "cda_ppo_loss_records")
            else:
              _check(L.cda_ppo_loss32(self.out.data_ptr(), p.theta.data_ptr() + self.L.OFF_LS * 4, acts[0].data_ptr(), acts[1].data_ptr(), acts[2].data_ptr(), acts[3].data_ptr(),
                                    logp_old.data_ptr(), adv.data_ptr(), ret.data_ptr(), self.perm.data_ptr() + o * 8, rs, self.A, NOUT,
                                    float(clip), float(vf_coef), float(ent_coef), self.d_out.data_ptr(), self.sums5.data_ptr(), self.out6.data_ptr(), rows,
                                    0 if (apply or k) else 1, 0 if (apply or not last) else 1, st),
                     "cda_ppo_loss32")          # (apply: the sums are finished and cleared by cda_mlp_adam; they accumulate over the sub-batches)
            _check(L.cda_mlp_backward(p.wb.data_ptr(), self.d_out.data_ptr(), self.h1p.data_ptr(), self.h2p.data_ptr(), rs, self.dz1p.data_ptr(), self.dz2p.data_ptr(),
                                      self.doutp.data_ptr(), self.bias_slab.data_ptr() + k * tiles_sub * BSLAB * 4, st), "cda_mlp_backward")
            _check(L.cda_mlp_wgrad(x_pk, self.h1p.data_ptr(), self.h2p.data_ptr(), self.dz1p.data_ptr(), self.dz2p.data_ptr(), self.doutp.data_ptr(), rs, chunks,
                                   self.slab.data_ptr() + k * chunks * self.L.SLAB * 4, st), "cda_mlp_wgrad")
        chunks, tiles = chunks * sub, tiles_sub * sub
        if apply:
            _check(L.cda_mlp_adam(p.theta.data_ptr(), p.adam_m.data_ptr(), p.adam_v.data_ptr(), p.adam_step.data_ptr(), p.wb.data_ptr(), self.slab.data_ptr(), chunks,
                                  self.bias_slab.data_ptr(), tiles, self.sums5.data_ptr(), rows * self.A, float(vf_coef), float(ent_coef), 0.0, self.out6.data_ptr(),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), float(max_norm),
                                  self.grad.data_ptr(), self.norm2.data_ptr(), st), "cda_mlp_adam")
        return chunks, tiles

    def run(self, obs_rows, acts=None, logp_old=None, adv=None, ret=None, epochs=4, clip=0.2, vf_coef=0.5, ent_coef=0.01, lr=5e-5, betas=(0.9, 0.999), eps=1e-8,
            max_norm=0.5, perms=None, records=None):
        """obs_rows f32 [R, 168] (one row per market-step); acts = (category i32, price i32, price_offset i32, a_cont f32[.., 2]) and logp_old / adv
        / ret f32, R * A entries each, sample r * A + a belonging to row r.  adv is expected normalised.  perms: optional i64 [epochs, R]
        (tests); default a keyed permutation per epoch.  records = (rec, adv sums or None, count) replaces acts / logp_old / adv / ret (see
        minibatch_step); with the sums given the advantages are normalised inside the loss.  max_norm = math.inf: no gradient clipping."""
        L, dev = _Fns(self.L), self.p.device
        assert obs_rows.shape == (self.R, self.L.OBS) and obs_rows.dtype == torch.float32 and obs_rows.is_contiguous()
        if records is not None:
            assert isinstance(records[0], int) or (records[0].dtype == torch.float32 and records[0].is_contiguous() and records[0].numel() >= self.R * self.A * 8)
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
        return {"pg_loss": self.out6[0], "v_loss": self.out6[1], "entropy": self.out6[2], "kl": self.out6[6]}
