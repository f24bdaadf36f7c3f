"""Multi-GPU sharding of the market batch: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Markets are fully independent (each owns its book, accounts and RNG - SURVEY §8e), so the simulation
needs NO collective: rank r steps the contiguous block [r*N/G, (r+1)*N/G).  Seeds derive from the GLOBAL
market index, so results do not depend on the GPU count.  The only exchange is the hand-back of the
per-market outputs to a central learner: one all-gather per step of a packed [n_local, obs_dim + 2A + 2]
float32 buffer (obs | reward f64 as 2 x f32 | terminated | truncated).  On the fully connected xGMI node
each rank pushes its shard directly to its 7 peers, so the gather is per-link bound.
"""
import torch


def shard_range(rank, world, n_total):
    """Contiguous block partition of the market axis."""
    if n_total % world != 0:
        raise ValueError(f"n_markets_total={n_total} must be divisible by world size {world}")
    per = n_total // world
    return rank * per, per


def global_seeds(seed_base, first, count):
    """Seed of global market i = seed_base + i (uint64 bit pattern carried in an int64 tensor)."""
    return (torch.arange(first, first + count, dtype=torch.int64) + int(seed_base))


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
            env_factory = lambda cfg, n, dev: CDAVecEnv(cfg, n_markets=n, device=dev, with_info=False)   # noqa: E731
        self.env = env_factory(config, self.n_local, device)
        self.obs_dim = self.env.obs_dim
        self.num_agents = self.env.num_agents
        self._packed = None
        self._gathered = None

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
        if self._gathered is None:
            self._gathered = torch.empty((self.n_total, self._packed.shape[1]), dtype=torch.float32, device=self._packed.device)
        self.dist.all_gather_into_tensor(self._gathered, self._packed)
        return unpack_outputs(self._gathered, self.obs_dim, self.num_agents)

    def close(self):
        self.env.close()
