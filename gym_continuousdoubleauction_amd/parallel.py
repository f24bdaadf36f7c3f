"""Multi-GPU sharding of the market batch: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Markets are fully independent (each owns its book, accounts and RNG - SURVEY §8e), so the simulation
needs NO collective: rank r steps the contiguous block [r*N/G, (r+1)*N/G).  Seeds derive from the GLOBAL
market index, so results do not depend on the GPU count.  The only exchange is the hand-back of the
per-market outputs to a central learner: ONE all-gather per step of the env's output slab
(obs f32[n,obs_dim] | reward f64[n,A] | terminated u8[n] | truncated u8[n], the bytes `k_step` itself
wrote - no packing pass).  On the fully connected xGMI node each rank pushes its shard directly to its 7
peers, so the gather is per-link bound (2.9 MB per rank at 4096 x 4: ~50 us at ~55 GB/s per link); it
is issued asynchronously (`gather_async`) and overlaps the NEXT step's kernel, whose outputs go to the
other slab of a double-buffered env.  `gather()` is the simple synchronous packed variant for any env.
"""
import torch


def shard_range(rank, world, n_total):
    """Contiguous block partition of the market axis: (first, count) of `rank`.  Every rank holds n_total // world
    markets; a remainder goes to the LAST rank (the slab all-gather pads every shard to the largest one)."""
    if not 0 <= rank < world or n_total < world:
        raise ValueError(f"need 0 <= rank < world <= n_markets_total, got rank={rank} world={world} n_markets_total={n_total}")
    per = n_total // world
    return rank * per, per + (n_total - per * world if rank == world - 1 else 0)


def shard_pad(world, n_total):
    """Markets of the largest shard (what every rank's slab is laid out for)."""
    return n_total // world + n_total % world


def global_seeds(seed_base, first, count):
    """Seed of global market i = seed_base + i (uint64 bit pattern carried in an int64 tensor)."""
    return (torch.arange(first, first + count, dtype=torch.int64) + int(seed_base))


def slab_layout(n, obs_dim, num_agents):
    """Byte offsets of one rank's per-step outputs inside its output slab (16-byte padded)."""
    o_obs = 0
    o_rew = o_obs + n * obs_dim * 4
    o_rew = (o_rew + 7) // 8 * 8
    o_term = o_rew + n * num_agents * 8
    o_trunc = o_term + n
    total = (o_trunc + n + 15) // 16 * 16
    return {"n": n, "obs_dim": obs_dim, "num_agents": num_agents,
            "obs": o_obs, "reward": o_rew, "terminated": o_term, "truncated": o_trunc, "bytes": total}


def slab_views(slab, lay):
    """Typed views (no copies) into uint8 slab(s) of shape [..., bytes]:
    obs f32[..., n, obs_dim], reward f64[..., n, A], terminated u8[..., n], truncated u8[..., n]."""
    n, od, a = lay["n"], lay["obs_dim"], lay["num_agents"]
    lead = slab.shape[:-1]
    obs = slab[..., lay["obs"]:lay["obs"] + n * od * 4].view(torch.float32).view(*lead, n, od)
    rew = slab[..., lay["reward"]:lay["reward"] + n * a * 8].view(torch.float64).view(*lead, n, a)
    term = slab[..., lay["terminated"]:lay["terminated"] + n]
    trunc = slab[..., lay["truncated"]:lay["truncated"] + n]
    return obs, rew, term, trunc


class GatherHandle:
    """An all-gather in flight.  wait() orders the caller's CURRENT stream after it (no host sync on a
    GPU) and returns per-rank views [world, n_local, ...] into the gathered buffer: obs, reward,
    terminated (bool), truncated (bool).  `.reshape(world * n_local, ...)` flattens them (one copy)."""

    def __init__(self, work, gathered, lay):
        self.work, self.gathered, self.lay = work, gathered, lay

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        obs, rew, term, trunc = slab_views(self.gathered, self.lay)
        return obs, rew, term != 0, trunc != 0


def pack_outputs(obs, reward, terminated, truncated, out=None):
    """[n, obs_dim] f32, [n, A] f64, [n] bool, [n] bool -> [n, obs_dim + 2A + 2] f32 (bit-preserving)."""
    n, od = obs.shape
    a2 = reward.shape[1] * 2
    if out is None:
        out = torch.empty((n, od + a2 + 2), dtype=torch.float32, device=obs.device)
    out[:, :od].copy_(obs)
    out[:, od:od + a2].copy_(reward.contiguous().view(torch.float32))
    out[:, od + a2].copy_(terminated.to(torch.float32))
    out[:, od + a2 + 1].copy_(truncated.to(torch.float32))
    return out


def unpack_outputs(packed, obs_dim, num_agents):
    od, a2 = obs_dim, 2 * num_agents
    obs = packed[:, :od]
    reward = packed[:, od:od + a2].contiguous().view(torch.float64)
    terminated = packed[:, od + a2] != 0
    truncated = packed[:, od + a2 + 1] != 0
    return obs, reward, terminated, truncated


class ShardedVecEnv:
    """This rank's shard of a global batch of `n_markets_total` markets.

    env_factory(config, n_local, device) builds the local stepper (default: the HIP CDAVecEnv on this
    rank's GPU; the CPU tests inject a stand-in with the same interface)."""

    def __init__(self, config, n_markets_total, device=None, env_factory=None, dist=None):
        import torch.distributed as tdist
        self.dist = dist or tdist
        self.rank = self.dist.get_rank() if self.dist.is_initialized() else 0
        self.world = self.dist.get_world_size() if self.dist.is_initialized() else 1
        self.n_total = int(n_markets_total)
        self.first, self.n_local = shard_range(self.rank, self.world, self.n_total)
        if env_factory is None:
            from .vec_env import CDAVecEnv
            env_factory = lambda cfg, n, dev: CDAVecEnv(cfg, n_markets=n, device=dev, with_info=False, out_buffers=2)   # noqa: E731
        self.env = env_factory(config, self.n_local, device)
        self.obs_dim = self.env.obs_dim
        self.num_agents = self.env.num_agents
        self._packed = None
        self._padded = None
        self._gathered = None
        self.layout = slab_layout(self.n_local, self.obs_dim, self.num_agents)
        self._gbufs, self._gnext = [None, None], 0

    def reset(self, seed_base=0):
        seeds = global_seeds(seed_base, self.first, self.n_local)
        return self.env.reset(seed=seeds)

    def step(self, category, size_mean, size_sigma, price, price_offset, present=None):
        """Actions for THIS rank's markets ([n_local, A]); returns the local outputs."""
        return self.env.step(category, size_mean, size_sigma, price, price_offset, present)

    def gather(self, obs, reward, terminated, truncated):
        """All-gather the per-market outputs of every rank -> global (obs, reward, terminated, truncated)."""
        self._packed = pack_outputs(obs, reward, terminated, truncated, self._packed)
        if self.world == 1:
            return unpack_outputs(self._packed, self.obs_dim, self.num_agents)
        n_pad, cols = shard_pad(self.world, self.n_total), self._packed.shape[1]
        send = self._packed
        if n_pad != self.n_local:           # uneven shards (the last rank holds the remainder): every contribution is padded to the largest
            if self._padded is None:
                self._padded = torch.zeros((n_pad, cols), dtype=torch.float32, device=send.device)
            self._padded[:self.n_local].copy_(send)
            send = self._padded
        if self._gathered is None:
            self._gathered = torch.empty((self.world * n_pad, cols), dtype=torch.float32, device=send.device)
        self.dist.all_gather_into_tensor(self._gathered, send)
        g = self._gathered
        if self.n_total % self.world:
            g = torch.cat([g[r * n_pad: r * n_pad + shard_range(r, self.world, self.n_total)[1]] for r in range(self.world)], dim=0)
        return unpack_outputs(g, self.obs_dim, self.num_agents)

    def gather_async(self, outputs=None):
        """Start the all-gather of this rank's output slab (the env's current one, i.e. the outputs of the
        step just enqueued; or a slab built from `outputs` for an env without one) and return a GatherHandle.  Two gathered buffers rotate, so at most two
        handles may be outstanding; with a double-buffered env the caller's loop is
            step(t); h = gather_async(); prev.wait(); prev = h
        which lets gather(t) run under the kernel of step t+1."""
        if self.n_total % self.world:
            raise ValueError("the slab all-gather needs equal shards (n_markets_total divisible by the world size); use gather()")
        slab = getattr(self.env, "out_slab", None) if outputs is None else None
        if slab is None:                    # an env without slab-backed outputs: build the slab (copies)
            if outputs is None:
                raise ValueError("this env has no output slab: pass outputs=(obs, reward, terminated, truncated)")
            slab = torch.zeros(self.layout["bytes"], dtype=torch.uint8, device=outputs[0].device)
            for dst, src in zip(slab_views(slab, self.layout), outputs):
                dst.copy_(src.to(dst.dtype))
        b = self._gnext
        self._gnext ^= 1
        if self._gbufs[b] is None:
            self._gbufs[b] = torch.zeros((self.world, self.layout["bytes"]), dtype=torch.uint8, device=slab.device)
        g = self._gbufs[b]
        if self.world == 1:
            g[0].copy_(slab)
            return GatherHandle(None, g, self.layout)
        work = self.dist.all_gather_into_tensor(g.view(-1), slab, async_op=True)
        return GatherHandle(work, g, self.layout)

    def close(self):
        self.env.close()
