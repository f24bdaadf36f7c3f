"""CDAVecEnv - N independent continuous-double-auction markets stepped in lockstep on one MI355X.

Host-side mirror of the reference env surface (continuousDoubleAuction_env.py:21-309) for a batch:
`reset()` / `step()` over torch tensors that stay resident in HBM.  All compute happens in the HIP
kernels behind the C-ABI (include/cda.h); PyTorch is only used for device memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi as K
from ._lib import CDAError, check, lib

_TORCH_OF = {C.c_int32: torch.int32, C.c_double: torch.float64, C.c_uint8: torch.uint8}

DEC_DTYPE = K.DEC_DTYPE

ACTION_KEYS = ("category", "size_mean", "size_sigma", "price", "price_offset")


class CDAVecEnv:
    """Batched env.  Tensors: actions [N,A]; obs f32[N, n_hist*42]; reward f64[N,A];
    terminated/truncated bool[N] (the reference's "__all__" flags); info = dict of SoA tensors."""

    def __init__(self, config=None, n_markets=1, device="cuda:0", with_info=True, out_buffers=1, groups=1, handback=False, group_streams=None):
        self.cfg_struct, self.config = K.make_config(config)
        self.n_markets = int(n_markets)
        self.host_epoch = 0                 # bumped by every host-side call that changes the markets (reset / step / run_random / place_order / set_state): consumers that keep
                                            # their own copy of the observations across calls (mlp.RolloutChains) compare it to know when theirs is stale
        self.num_agents = self.cfg_struct.num_agents
        self.n_hist = self.cfg_struct.n_hist
        self.obs_dim = self.n_hist * K.SNAPSHOT_DIM
        self.max_step = self.cfg_struct.max_step
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise CDAError("CDAVecEnv needs a HIP device (torch device 'cuda:<i>'); there is no CPU fallback")
        if not torch.cuda.is_available():
            raise CDAError("no GPU visible to PyTorch-ROCm; the HIP path cannot run and there is no CPU fallback")
        self.device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.agents = [f"agent_{i}" for i in range(self.num_agents)]
        h = C.c_void_p()
        check(lib().cda_create(C.byref(self.cfg_struct), self.n_markets, self.device_index, C.byref(h)), "cda_create")
        self._h = h
        self.book_capacity = int(lib().cda_book_capacity(h))     # LDS book tile: 256 or 512 resting orders per market (config['book_capacity'])
        self.book_spill = int(lib().cda_book_spill(h))           # HBM spill ring behind it: orders per side (config['book_spill']; 0 = none)
        self.book_spill_wanted = int(lib().cda_book_spill_wanted(h))   # ... and what an unbounded-inside-an-episode book needs
        if self.book_spill < self.book_spill_wanted:
            import warnings
            warnings.warn(f"the HBM spill ring was cut to {self.book_spill} orders per side ({self.book_spill_wanted} needed for a book that can never overflow inside an "
                          f"episode of {self.max_step} steps): device memory; a rest beyond tile + ring is dropped and flagged (CDA_FLAG_BOOK_OVERFLOW)", RuntimeWarning, stacklevel=2)
        N, A, dev = self.n_markets, self.num_agents, self.device
        # The per-step outputs of one launch live in ONE contiguous slab (obs | reward | terminated | truncated),
        # so a multi-GPU caller hands them to its peers with a single collective and no packing pass
        # (parallel.slab_layout describes the byte offsets).  out_buffers > 1 rotates the slab every step:
        # step t+1 writes a different slab while step t's is still being read by an all-gather in flight.
        from .parallel import slab_layout, slab_views
        self.slab_layout = slab_layout(N, self.obs_dim, A)
        self.with_info = bool(with_info)
        # ONE device allocation holds the output slab(s) and, behind them, every info tensor (16-byte aligned each):
        # a host-side consumer (the dict facades) moves a whole step's outputs with a single D2H copy of `packed`.
        B, sb = max(1, int(out_buffers)), self.slab_layout["bytes"]
        self.info_layout, off = {}, B * sb
        if self.with_info:
            for name, ct, per_agent, dims in K.INFO_FIELDS:
                shape = ((N, A) if per_agent else (N,)) + tuple(dims)
                shape = shape + (16,) if ct is K.Dec else shape
                dt = torch.uint8 if ct is K.Dec else _TORCH_OF[ct]
                nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
                self.info_layout[name] = (off, dt, shape, nbytes)
                off += (nbytes + 15) // 16 * 16
        self._all = torch.zeros(off, dtype=torch.uint8, device=dev)
        self._slabs = self._all[:B * sb].view(B, sb)
        self._views = [slab_views(self._slabs[b], self.slab_layout) for b in range(B)]
        self._out_ptrs = [tuple(t.data_ptr() for t in v) for v in self._views]
        self._step_call = lib().cda_step
        self._cur = 0
        self._bind_outputs()
        self.info = {}
        self._info_ptrs = K.InfoPtrs()
        for name, (o, dt, shape, nbytes) in self.info_layout.items():
            t = self._all[o:o + nbytes].view(dt).view(shape)
            self.info[name] = t
            setattr(self._info_ptrs, name, t.data_ptr())
        self._info_ref = C.byref(self._info_ptrs) if self.with_info else None
        # handback: every step / reset also writes ONE compact record per market of what is new (newest frame | reward | flags; see
        # cda_set_handback in include/cda.h) - what a multi-GPU caller all-gathers instead of the whole observation (parallel.py)
        self.handback = None
        if handback:
            self.handback_stride = int(lib().cda_handback_stride(A))
            self.handback = torch.zeros((N, self.handback_stride), dtype=torch.uint8, device=dev)
            check(lib().cda_set_handback(h, self.handback.data_ptr()), "cda_set_handback")
        # groups > 1: step() launches the batch as `groups` contiguous market groups, each an independent chain of
        # launches on its own stream (cda_step_groups).  Markets never interact, so nothing is lost - and a group's
        # slowest market no longer stalls the markets of the other groups.  See group_streams / join().
        self.groups = int(groups)
        if not 1 <= self.groups <= min(K.MAX_GROUPS, self.n_markets):
            raise ValueError(f"groups must be in 1..{min(K.MAX_GROUPS, self.n_markets)}")
        self.group_ranges = []
        for g in range(self.groups):
            first, cnt = C.c_int32(), C.c_int32()
            lib().cda_group_range(self.n_markets, self.groups, g, C.byref(first), C.byref(cnt))
            self.group_ranges.append((first.value, cnt.value))
        self.group_streams = []
        if self.groups > 1:
            from .streams import concurrent_streams
            if group_streams is not None:                   # the caller's own streams (e.g. a second groups > 1 env of the process)
                if len(group_streams) != self.groups:
                    raise ValueError(f"need {self.groups} group streams")
                self.group_streams = list(group_streams)
            else:
                self.group_streams = list(concurrent_streams(dev, self.groups))     # streams on DISTINCT hardware queues (measured once per process)
            self._stream_arr = (C.c_void_p * self.groups)(*[s.cuda_stream for s in self.group_streams])
            self._groups_call = lib().cda_step_groups
            self._fork_ev = torch.cuda.Event()
            self._join_ev = [torch.cuda.Event() for _ in range(self.groups)]
            self._need_fork = True

    @property
    def packed(self):
        """uint8 view of everything a step writes: [output slab(s) | info tensors] (`slab_layout`, `info_layout`)."""
        return self._all

    def unpack_host(self, host):
        """numpy views (no copies) into a HOST copy of `packed` (a uint8 numpy array or CPU tensor): the current
        slab's (obs, reward, terminated, truncated) and the info dict."""
        h = host.numpy() if isinstance(host, torch.Tensor) else host
        lay, sb = self.slab_layout, self.slab_layout["bytes"]
        base = self._cur * sb
        n, od, a = lay["n"], lay["obs_dim"], lay["num_agents"]
        obs = h[base + lay["obs"]: base + lay["obs"] + n * od * 4].view(np.float32).reshape(n, od)
        rew = h[base + lay["reward"]: base + lay["reward"] + n * a * 8].view(np.float64).reshape(n, a)
        term = h[base + lay["terminated"]: base + lay["terminated"] + n]
        trunc = h[base + lay["truncated"]: base + lay["truncated"] + n]
        info = {}
        for name, (o, dt, shape, nbytes) in self.info_layout.items():
            npdt = {torch.uint8: np.uint8, torch.int32: np.int32, torch.float64: np.float64}[dt]
            info[name] = h[o:o + nbytes].view(npdt).reshape(shape)
        return obs, rew, term, trunc, info

    # ------------------------------------------------------------------ host-resident step I/O (the dict facades)
    def bind_host_io(self):
        """The dict facades' step I/O without a copy in either direction: ONE pinned host block with the layout of `packed` (slab 0 | info tensors) that the step
        kernel WRITES over the fabric (its outputs are write-only, nontemporal stores), and action arrays the kernel READS from pinned host memory (pinned host
        memory is device-addressable at its own address on ROCm).  A one-market step then costs one launch and one stream synchronisation: no H2D / D2H copy call,
        no second trip through the caching allocator.  Returns the block as a uint8 numpy array (valid after `sync_host_io()`); groups == 1 only."""
        if self.groups != 1:
            raise CDAError("host-resident step I/O is for single-launch envs (groups == 1)")
        if getattr(self, "_hio", None) is None:
            host = torch.zeros(self._all.numel(), dtype=torch.uint8).pin_memory()
            base = host.data_ptr()
            lay = self.slab_layout
            outs = (base + lay["obs"], base + lay["reward"], base + lay["terminated"], base + lay["truncated"])
            ptrs = K.InfoPtrs()
            for name, (o, _dt, _shape, _nbytes) in self.info_layout.items():
                setattr(ptrs, name, base + o)
            self._hio = (host, outs, ptrs, C.byref(ptrs) if self.with_info else None)
        return self._hio[0].numpy()

    def step_host_io(self, action_ptrs):
        """cda_step with the six action arrays at `action_ptrs` (category, size_mean, size_sigma, price, price_offset, present: pinned host or device addresses) and the
        outputs into the block of bind_host_io(), on the caller's current stream.  Nothing is valid on the host before sync_host_io()."""
        self.host_epoch += 1
        _host, outs, _ptrs, info_ref = self._hio
        stream = torch.cuda.current_stream(self.device)
        if torch.cuda.current_device() == self.device_index:
            rc = self._step_call(self._h, *action_ptrs, *outs, info_ref, stream.cuda_stream)
        else:
            with torch.cuda.device(self.device):
                rc = self._step_call(self._h, *action_ptrs, *outs, info_ref, stream.cuda_stream)
        if rc != 0:
            check(rc, "cda_step")
        self._hio_stream = stream

    def reset_host_io(self, seed=None, mask=None):
        """reset() whose observation lands in the block of bind_host_io() (slab 0's obs) instead of the device tensor."""
        self.reset(seed=seed, mask=mask, _obs_ptr=self._hio[1][0])
        self._hio_stream = torch.cuda.current_stream(self.device)

    def sync_host_io(self):
        self._hio_stream.synchronize()

    def _bind_outputs(self):
        self.out_slab = self._slabs[self._cur]
        self.obs, self.reward, self._term, self._trunc = self._views[self._cur]

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            lib().cda_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ reset / step
    def reset(self, seed=None, mask=None, _obs_ptr=None):
        """reset(seed=None): every selected market keeps its RNG stream (a never-seeded market is seeded
        with its index).  seed=int s: market i is seeded SeedSequence(s + i).  seed=array/tensor of N
        unsigned 64-bit seeds: per market.  mask: bool/uint8 [N] selecting the markets to reset.
        (_obs_ptr: where the first observations go instead of `self.obs` - reset_host_io.)"""
        seeds_t = None
        if seed is not None:
            if isinstance(seed, (int, np.integer)):
                arr = (np.uint64(int(seed)) + np.arange(self.n_markets, dtype=np.uint64)).astype(np.uint64)
            elif isinstance(seed, torch.Tensor):
                arr = seed.detach().cpu().numpy().astype(np.int64).view(np.uint64)
            else:
                arr = np.ascontiguousarray(seed, dtype=np.uint64)
            if arr.shape != (self.n_markets,):
                raise ValueError(f"need {self.n_markets} seeds, got shape {arr.shape}")
            seeds_t = torch.from_numpy(arr.view(np.int64).copy()).to(self.device)
        mask_t = None
        if mask is not None:
            mask_t = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            if mask_t.shape != (self.n_markets,):
                raise ValueError("mask must have shape [n_markets]")
        self.host_epoch += 1
        self.join()                         # (groups > 1: the reset is issued on the caller's stream, after every group's last step)
        with torch.cuda.device(self.device):
            check(lib().cda_reset(self._h, seeds_t.data_ptr() if seeds_t is not None else None,
                                  mask_t.data_ptr() if mask_t is not None else None,
                                  self.obs.data_ptr() if _obs_ptr is None else _obs_ptr, self._stream()), "cda_reset")
        self._keep = (seeds_t, mask_t)
        if self.groups > 1:
            self._need_fork = True          # the group streams must see this reset (issued on the caller's stream)
        return self.obs

    # ------------------------------------------------------------------ market groups (groups > 1)
    def fork(self):
        """Order every group stream after the work enqueued so far on the caller's current stream (the reset, the
        actions a policy just produced).  step() does it by itself after reset() and join()."""
        self._fork_ev.record(torch.cuda.current_stream(self.device))
        for s in self.group_streams:
            s.wait_event(self._fork_ev)
        self._need_fork = False

    def sync(self):
        """Host-side barrier: wait until the device has finished everything enqueued on ANY stream (the group streams included).
        Afterwards no stream has pending work, so the next step() needs no fork and a consumer needs no join - unlike join(),
        this puts no dependency edge into any stream."""
        torch.cuda.synchronize(self.device)
        if self.groups > 1:
            self._need_fork = False

    def join(self):
        """Order the caller's current stream after every group's last step, so that the outputs can be consumed there;
        the next step() forks again.  A caller that pipelines per group works inside `torch.cuda.stream(group_streams[g])`
        on the rows `group_ranges[g]` instead and never joins."""
        if self.groups > 1:
            cur = torch.cuda.current_stream(self.device)
            for ev, s in zip(self._join_ev, self.group_streams):
                ev.record(s)
                cur.wait_event(ev)
            self._need_fork = True

    def _prep(self, x, dtype):
        if (isinstance(x, torch.Tensor) and x.dtype == dtype and x.device == self.device and x.is_contiguous()
                and x.numel() == self.n_markets * self.num_agents):
            return x                        # the hot loop's case: already a resident tensor of the ABI's layout
        t = torch.as_tensor(x)
        if t.device != self.device or t.dtype != dtype:
            t = t.to(device=self.device, dtype=dtype)
        t = t.reshape(self.n_markets, self.num_agents)
        return t.contiguous()

    def step(self, category, size_mean=None, size_sigma=None, price=None, price_offset=None, present=None, pipelined=False):
        """One env step for all markets.  `category` may also be a dict holding the five action tensors.  `present` u8[N,A]
        (optional): 0 = the agent is not in this step's action dict; non-zero values also carry the dict's iteration order
        (agents are processed by ascending value, ties by agent index - see include/cda.h).

        groups > 1: pipelined=True sends the launches to the group streams with no edge to the caller's stream: the caller orders things itself (fork() /
        join() / sync(), or per-group work on group_streams[g]) - what bench.py's free-running loop and the rollout chains do; a group's slowest market
        then overlaps the other groups' next steps.  The default (pipelined=False) is ordered AFTER everything enqueued on the caller's current stream (the
        tensors a policy just wrote) and the caller's stream AFTER the step, so it composes like any other stream-ordered op - since round 6 as ONE launch of
        the whole batch on the caller's stream: a step that is joined at once has no tail to overlap, and the fork + join event edges of G chains cost more
        than G concurrent launches gain (bench.py value_ordered_per_step: 202 M with the edges, the one launch's rate without; results are identical)."""
        if isinstance(category, dict):
            d = category
            present = d.get("present", present)
            category, size_mean, size_sigma, price, price_offset = (d[k] for k in ACTION_KEYS)
        self.host_epoch += 1
        cat = self._prep(category, torch.int32)
        sm = self._prep(size_mean, torch.float32)
        ss = self._prep(size_sigma, torch.float32)
        pr = self._prep(price, torch.int32)
        po = self._prep(price_offset, torch.int32)
        ps = None if present is None else self._prep(present, torch.uint8)
        if len(self._views) > 1:
            self._cur = (self._cur + 1) % len(self._views)
            self._bind_outputs()
        if self.groups > 1 and not pipelined and not self._need_fork:
            self.join()                     # the group streams hold steps the caller's stream has not seen: order it after them (and the next pipelined step forks)
        if self.groups > 1 and pipelined:
            if self._need_fork:
                self.fork()
            fn = self._groups_call
            call = (self._h, self.groups, cat.data_ptr(), sm.data_ptr(), ss.data_ptr(), pr.data_ptr(), po.data_ptr(),
                    ps.data_ptr() if ps is not None else None, *self._out_ptrs[self._cur], self._info_ref, self._stream_arr)
        else:
            fn = self._step_call
            call = (self._h, cat.data_ptr(), sm.data_ptr(), ss.data_ptr(), pr.data_ptr(), po.data_ptr(),
                    ps.data_ptr() if ps is not None else None, *self._out_ptrs[self._cur],
                    self._info_ref, torch.cuda.current_stream(self.device).cuda_stream)
        if torch.cuda.current_device() == self.device_index:
            rc = fn(*call)
        else:                               # the library selects its own device; keep the caller's current one intact
            with torch.cuda.device(self.device):
                rc = fn(*call)
        if rc != 0:
            check(rc, "cda_step")
        # keep the inputs of the last TWO steps alive: a pipelined caller may have step t still reading them on a group stream
        # when step t+1 is enqueued (the caching allocator would otherwise hand the memory out again on the caller's stream)
        self._keep_prev = getattr(self, "_keep", None)
        self._keep = (cat, sm, ss, pr, po, ps)
        return self.obs, self.reward, self._term.view(torch.bool), self._trunc.view(torch.bool), self.info

    def run_random(self, n_steps, action_seed=0, market_index_base=0):
        """The reference's `CDA_rand.run_random` for every market, in ONE kernel launch: uniform random agents (the
        counter-based sampler of include/cda_random_agents.h) play up to `n_steps` steps, each market stopping at its own
        episode end.  Returns (obs f32[N,obs_dim], episode_return f64[N,A], terminated, truncated, steps_taken i32[N]);
        bit-identical to `n_steps` calls of step() on `random_actions(t)`."""
        self.host_epoch += 1
        N, A, dev = self.n_markets, self.num_agents, self.device
        self.join()
        if not hasattr(self, "_rr_steps"):
            self._rr_return = torch.zeros((N, A), dtype=torch.float64, device=dev)
            self._rr_steps = torch.zeros(N, dtype=torch.int32, device=dev)
        with torch.cuda.device(self.device):
            check(lib().cda_run_random(self._h, int(n_steps), int(action_seed) & (2 ** 64 - 1), int(market_index_base) & (2 ** 64 - 1),
                                       self.obs.data_ptr(), self._rr_return.data_ptr(), self._term.data_ptr(), self._trunc.data_ptr(),
                                       self._rr_steps.data_ptr(), self._stream()), "cda_run_random")
        return self.obs, self._rr_return, self._term.view(torch.bool), self._trunc.view(torch.bool), self._rr_steps

    def random_actions_device(self, step0, n_steps, action_seed=0, market_index_base=0):
        """The same stream generated on the device: five [n_steps, N, A] tensors for steps step0 .. step0 + n_steps - 1."""
        N, A, dev = self.n_markets, self.num_agents, self.device
        cat, price, off = (torch.empty((n_steps, N, A), dtype=torch.int32, device=dev) for _ in range(3))
        mean, sigma = (torch.empty((n_steps, N, A), dtype=torch.float32, device=dev) for _ in range(2))
        with torch.cuda.device(self.device):
            check(lib().cda_random_actions(int(action_seed) & (2 ** 64 - 1), int(market_index_base) & (2 ** 64 - 1), int(step0), int(n_steps), N, A,
                                           cat.data_ptr(), mean.data_ptr(), sigma.data_ptr(), price.data_ptr(), off.data_ptr(), self._stream()),
                  "cda_random_actions")
        return cat, mean, sigma, price, off

    def random_actions(self, step, action_seed=0, market_index_base=0):
        """The five [N,A] action arrays (numpy, host) the random agents of `run_random` play at step `step`."""
        N, A = self.n_markets, self.num_agents
        cat, price, off = (np.zeros((N, A), np.int32) for _ in range(3))
        mean, sigma = (np.zeros((N, A), np.float32) for _ in range(2))
        check(lib().cda_random_actions_host(int(action_seed) & (2 ** 64 - 1), int(market_index_base) & (2 ** 64 - 1), int(step), N, A,
                                            cat.ctypes.data, mean.ctypes.data, sigma.ctypes.data, price.ctypes.data, off.ctypes.data),
              "cda_random_actions_host")
        return cat, mean, sigma, price, off

    # ------------------------------------------------------------------ diagnostics
    def place_order(self, market, trader, type_, side, size, price=1):
        self.host_epoch += 1
        check(lib().cda_place_order(self._h, market, trader, type_, side, size, price), "cda_place_order")

    def mark_to_mkt(self, market=0):
        check(lib().cda_mark_to_mkt(self._h, market), "cda_mark_to_mkt")

    def get_state(self, market=0):
        s = K.MarketState()
        torch.cuda.synchronize(self.device)
        check(lib().cda_get_state(self._h, market, C.byref(s)), "cda_get_state")
        return s

    def get_book(self, market=0, side=None):
        """One market's book, whole, in queue order (best price first, FIFO inside a level): int32 [n, 5] rows of
        (price, qty, owner, order_id, timestamp); side 0 bids, 1 asks, None = (bids, asks).  Unlike get_state() this is
        not limited to the first BOOK_CAP_MAX orders of a side."""
        if side is None:
            return self.get_book(market, 0), self.get_book(market, 1)
        torch.cuda.synchronize(self.device)
        n = C.c_int32()
        check(lib().cda_get_book(self._h, market, side, None, 0, C.byref(n)), "cda_get_book")
        buf = (K.Order * max(n.value, 1))()
        check(lib().cda_get_book(self._h, market, side, C.cast(buf, C.c_void_p), n.value, C.byref(n)), "cda_get_book")
        return np.ctypeslib.as_array(buf).view(np.int32).reshape(-1, 5)[: n.value].copy()

    def set_state(self, market, state):
        self.host_epoch += 1
        torch.cuda.synchronize(self.device)
        check(lib().cda_set_state(self._h, market, C.byref(state)), "cda_set_state")

    def raw_snapshot(self):
        self.join()
        raw = torch.zeros((self.n_markets, K.RAW_DIM), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().cda_get_raw_snapshot(self._h, raw.data_ptr(), self._stream()), "cda_get_raw_snapshot")
        return raw

    def flags(self):
        self.join()
        f = torch.zeros(self.n_markets, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().cda_last_flags(self._h, f.data_ptr(), self._stream()), "cda_last_flags")
        return f

    def book_peak(self):
        """Census: the most resting orders (both sides together) each market has held since its last reset, int32[N]."""
        self.join()
        p = torch.zeros(self.n_markets, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().cda_book_peak(self._h, p.data_ptr(), self._stream()), "cda_book_peak")
        return p

    def check_invariants(self):
        """Structural invariants of every market on the device (include/cda.h CDA_INV_*): int32[N] of violation bits, 0 = sides
        sorted by price, book uncrossed, quantities positive, cash_on_hold == value of own resting orders, positions
        net to zero."""
        self.join()
        v = torch.zeros(self.n_markets, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().cda_check_invariants(self._h, v.data_ptr(), self._stream()), "cda_check_invariants")
        return v

    def nav_conservation(self, tolerance=1e-6):
        """The reference's end-of-episode invariant for every market, computed on the device in the ledger's own decimal
        arithmetic: (float(|sum of NAV - A * init_cash|) f64[N], violated bool[N])."""
        self.join()
        err = torch.zeros(self.n_markets, dtype=torch.float64, device=self.device)
        bad = torch.zeros(self.n_markets, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().cda_nav_conservation(self._h, float(tolerance), err.data_ptr(), bad.data_ptr(), self._stream()), "cda_nav_conservation")
        return err, bad.view(torch.bool)

    # ------------------------------------------------------------------ every episode checked and summarised on the device
    def enable_episode_metrics(self, on=True, nav_tolerance=1e-6):
        """From now on every step tallies what the reference's callback tallies per episode, and every episode END is checked (exact sum of NAV against
        num_agents x init_cash, `nav_tolerance` = the callback's) and credited on the device, in the cold paths that handle it - the in-kernel auto reset
        included (include/cda.h cda_episode_metrics_enable; train/callbk/league_based_self_play_callback.py:541-755)."""
        check(lib().cda_episode_metrics_enable(self._h, 1 if on else 0, float(nav_tolerance)), "cda_episode_metrics_enable")
        self.episode_metrics_on = bool(on)

    def collect_episode_metrics(self, module_of=None, n_modules=1, clear=True, out=None):
        """The episodes that ended since the last collection, reduced on the device (two launches, a fixed order): (f64 [n_modules, EM_AGENT_FIELDS] per-module
        sums over the (episode, agent) pairs module m played - module_of i32 [N, A], None = one module -, f64 [EM_ENV_FIELDS] per-env sums); device tensors,
        columns = _capi.EM_*.  episode_metrics.summarise() turns them into the callback's metrics."""
        self.join()
        if out is None:                                     # (both are written whole by the second launch)
            out = (torch.empty((int(n_modules), K.EM_AGENT_FIELDS), dtype=torch.float64, device=self.device), torch.empty(K.EM_ENV_FIELDS, dtype=torch.float64, device=self.device))
        if module_of is not None:
            assert module_of.dtype == torch.int32 and module_of.is_contiguous() and module_of.numel() == self.n_markets * self.num_agents and module_of.device == self.device
        with torch.cuda.device(self.device):
            check(lib().cda_episode_metrics_collect(self._h, module_of.data_ptr() if module_of is not None else None, int(n_modules), out[0].data_ptr(), out[1].data_ptr(),
                                                    1 if clear else 0, self._stream()), "cda_episode_metrics_collect")
        return out

    def state_bytes_per_market(self):
        return int(lib().cda_state_bytes_per_market(self._h))

    def nav_decimals(self):
        """info['nav'] as exact decimal.Decimal objects, [N][A] nested lists (host copy)."""
        raw = self.info["nav"].cpu().numpy().view(DEC_DTYPE).reshape(self.n_markets, self.num_agents)
        return [[K.dec_to_decimal(raw[i, a]) for a in range(self.num_agents)] for i in range(self.n_markets)]


def selftest_dec(op, a, b=None, device=0):
    """Device self-test of the ledger arithmetic: numpy DEC_DTYPE arrays in/out."""
    a = np.ascontiguousarray(a)
    out = np.zeros(len(a), DEC_DTYPE)
    bp = None if b is None else np.ascontiguousarray(b).ctypes.data
    check(lib().cda_selftest_dec(device, op, len(a), a.ctypes.data, bp, out.ctypes.data), "cda_selftest_dec")
    return out


def selftest_rng(seed, lo, hi, n_steps, n_normals, perm_n, device=0):
    first = np.zeros(2, np.int32)
    normals = np.zeros((n_steps, n_normals), np.float64)
    perms = np.zeros((n_steps, max(perm_n, 1)), np.int32)
    fs = np.zeros(8, np.uint64)
    perm_arg = perms[:, :perm_n].copy() if perm_n else perms
    check(lib().cda_selftest_rng(device, seed, lo, hi, n_steps, n_normals, perm_n, first.ctypes.data,
                                 normals.ctypes.data, perm_arg.ctypes.data, fs.ctypes.data), "cda_selftest_rng")
    return int(first[0]), normals, perm_arg[:, :perm_n], fs


def selftest_libm(op, x, device=0):
    """The restated libm function (op 0: log1p, 1: exp) on the DEVICE (device >= 0) or by the same source compiled for the
    host (device=None; needs no GPU): float64 numpy in / out."""
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x)
    if device is None:
        check(lib().cda_selftest_libm_host(op, len(x), x.ctypes.data, y.ctypes.data), "cda_selftest_libm_host")
    else:
        check(lib().cda_selftest_libm(device, op, len(x), x.ctypes.data, y.ctypes.data), "cda_selftest_libm")
    return y
