"""League slot mapping on the batched env: which policy module plays which agent slot of
which market for the coming episode.

Reference rule (train/callbk/league_based_self_play_callback.py:1286-1344, `get_mapping_fn`):
  * slots below `num_trainable` always play their own trainable module `policy_<slot>`;
  * every other slot draws from the pool `available_modules[num_trainable:]` with weight `champion_weight`
    for champion snapshots (`champion_*`), `original_opponent_weight` for the fixed opponents (`policy_*`),
    1.0 otherwise, normalised in float64;
  * the draw is `np.random.RandomState(seed).choice(pool, p=probs)` with
    `seed = (crc32(str(episode_id)) + slot) mod 2^32`, so it is a pure function of (episode id, slot).

`RandomState.choice(p=...)` consumes ONE `random_sample()` (a 53-bit double from two MT19937 words) and returns
`searchsorted(cumsum(p) / cumsum(p)[-1], u, side="right")`.  For a batch of N markets that is N x (A - k)
independent generators, so the first double of each is computed for all seeds at once (`mt19937_first_double`:
the 624-word `init_genrand` recurrence vectorised over the seed axis, then the tempering of words 0 and 1).
Module ids and prefixes follow config/tunable_constants.json -> module_id_prefixes of the reference.
"""
import zlib

import numpy as np

POLICY_PREFIX = "policy_"
CHAMPION_PREFIX = "champion_"


def policy_id(i):
    return f"{POLICY_PREFIX}{i}"


def mt19937_first_double(seeds):
    """First `random_sample()` of `np.random.RandomState(seed)` for every 32-bit seed in `seeds`."""
    s = np.asarray(seeds, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    n = s.shape[0]
    mt = np.empty((624, n), dtype=np.uint64)
    mt[0] = s
    m32 = np.uint64(0xFFFFFFFF)
    for i in range(1, 624):                                   # init_genrand (Matsumoto & Nishimura, mt19937ar.c)
        prev = mt[i - 1]
        mt[i] = (np.uint64(1812433253) * (prev ^ (prev >> np.uint64(30))) + np.uint64(i)) & m32

    def word(k):                                              # k-th output: twist of slot k, then tempering
        y = (mt[k] & np.uint64(0x80000000)) | (mt[k + 1] & np.uint64(0x7FFFFFFF))
        v = mt[k + 397] ^ (y >> np.uint64(1)) ^ np.where((y & np.uint64(1)) != 0, np.uint64(0x9908B0DF), np.uint64(0))
        v ^= v >> np.uint64(11)
        v ^= (v << np.uint64(7)) & np.uint64(0x9D2C5680)
        v ^= (v << np.uint64(15)) & np.uint64(0xEFC60000)
        v ^= v >> np.uint64(18)
        return v & m32

    a, b = word(0) >> np.uint64(5), word(1) >> np.uint64(6)
    return (a.astype(np.float64) * 67108864.0 + b.astype(np.float64)) / 9007199254740992.0


class LeagueSlotMapper:
    """Batched counterpart of the reference's agent-to-module mapping function."""

    def __init__(self, num_agents, num_trainable, num_fixed_opponents=None, original_opponent_weight=1.0, champion_weight=1.0):
        self.num_agents = int(num_agents)
        self.num_trainable = int(num_trainable)
        if not 0 <= self.num_trainable <= self.num_agents:
            raise ValueError("num_trainable must lie in [0, num_agents]")
        if num_fixed_opponents is None:
            num_fixed_opponents = self.num_agents - self.num_trainable
        # available_modules: trainables first, then the fixed opponents, then champions in the order they were added
        self.available_modules = [policy_id(i) for i in range(self.num_trainable + int(num_fixed_opponents))]
        self.original_opponent_weight = float(original_opponent_weight)
        self.champion_weight = float(champion_weight)
        self.champion_id_counter = 0

    def add_champion(self, module_id=None):
        """Register a champion snapshot in the matchmaking pool; ids are minted `champion_<n>` and never reused."""
        if module_id is None:
            self.champion_id_counter += 1
            module_id = f"{CHAMPION_PREFIX}{self.champion_id_counter}"
        if module_id in self.available_modules:
            raise ValueError(f"{module_id} is already in the pool")
        self.available_modules.append(module_id)
        return module_id

    def remove(self, module_id):
        self.available_modules.remove(module_id)

    def pool(self):
        return self.available_modules[self.num_trainable:]

    def pool_probabilities(self):
        w = np.array([self.champion_weight if c.startswith(CHAMPION_PREFIX) else
                      (self.original_opponent_weight if c.startswith(POLICY_PREFIX) else 1.0) for c in self.pool()], dtype=np.float64)
        return w / w.sum()

    def assign(self, episode_ids):
        """episode_ids: N episode identifiers (anything `str()` renders the way the caller's episodes do).
        Returns int64 [N, A]: index into `available_modules` of the module that plays slot a of market i."""
        n, A, k = len(episode_ids), self.num_agents, self.num_trainable
        out = np.empty((n, A), dtype=np.int64)
        out[:, :k] = np.arange(k)
        if A == k:
            return out
        cand = self.pool()
        if not cand:                                          # the reference's fallback for an empty pool
            out[:, k:] = np.arange(k, A)
            return out
        p = self.pool_probabilities()
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        base = np.array([zlib.crc32(str(e).encode("utf-8")) for e in episode_ids], dtype=np.uint64)
        slots = np.arange(k, A, dtype=np.uint64)
        seeds = ((base[:, None] + slots[None, :]) % np.uint64(2 ** 32)).reshape(-1)
        u = mt19937_first_double(seeds)
        out[:, k:] = k + np.searchsorted(cdf, u, side="right").reshape(n, A - k)
        return out

    def episode_crcs(self, episode_ids):
        """zlib.crc32(str(id)) per episode: the host half of the seed rule (the generator half runs on the device, assign_device)"""
        return np.array([zlib.crc32(str(e).encode("utf-8")) for e in episode_ids], dtype=np.uint32)

    def assign_device(self, bank, episode_ids=None, crcs=None, net_of=None, slot_pool=None):
        """assign() on the device (include/cda_mlp.h cda_league_assign), straight into `bank.slot_net` (mlp.PolicyBank): no host sync, one launch.
        net_of: {module id: bank row} for the pool's network modules (champions); modules not named there play the uniform random law
        (the reference's fixed opponents are RandomRLModules, train/model/model_handler.py:38-53).  slot_pool (optional i32 [N, A] device tensor)
        receives the draw itself: available_modules[num_trainable + slot_pool] is the module's id (-1: the slot's own trainable policy)."""
        import torch
        from ._lib import check, lib
        from .mlp import LEAGUE_RANDOM
        if crcs is None:
            crcs = self.episode_crcs(episode_ids)
        N, A = bank.slot_net.shape
        if len(crcs) != N or A != self.num_agents:
            raise ValueError("one episode per market of the bank")
        dev = bank.device
        cand = self.pool()
        if not cand:                                          # the reference's fallback for an empty pool: slot a plays policy_a (a random module here)
            bank.slot_net[:, self.num_trainable:] = LEAGUE_RANDOM
            if slot_pool is not None:
                slot_pool.fill_(-1)
            return bank.slot_net
        cdf = np.cumsum(self.pool_probabilities())
        cdf /= cdf[-1]
        nets = np.array([(net_of or {}).get(c, LEAGUE_RANDOM) for c in cand], dtype=np.int32)
        keep = (torch.from_numpy(np.asarray(crcs, dtype=np.uint32).view(np.int32).copy()).to(dev, non_blocking=True), torch.from_numpy(cdf).to(dev, non_blocking=True),
                torch.from_numpy(nets).to(dev, non_blocking=True))
        with torch.cuda.device(dev):
            check(lib().cda_league_assign(keep[0].data_ptr(), N, A, self.num_trainable, keep[1].data_ptr(), keep[2].data_ptr(), len(cand), bank.slot_net.data_ptr(),
                                          slot_pool.data_ptr() if slot_pool is not None else None, torch.cuda.current_stream(dev).cuda_stream), "cda_league_assign")
        self._keep = keep                                      # (the launch reads them asynchronously)
        return bank.slot_net

    def names(self, assignment):
        mods = np.array(self.available_modules, dtype=object)
        return mods[assignment]

    def group_by_module(self, assignment):
        """{module id: (market index array, slot index array)} - one batched policy forward per module."""
        groups = {}
        for idx in np.unique(assignment):
            mk, sl = np.nonzero(assignment == idx)
            groups[self.available_modules[int(idx)]] = (mk, sl)
        return groups


class RandomModule:
    """The fixed opponent of the reference (train/model/model_handler.py:38-53 RandomRLModule): every action
    component uniform over its space, drawn on the device."""

    def __init__(self, device, seed=0):
        import torch
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(int(seed))
        self.device = device

    def __call__(self, obs):
        import torch
        m, g, d = obs.shape[0], self.gen, self.device
        return (torch.randint(0, 9, (m,), generator=g, device=d, dtype=torch.int32),
                torch.rand((m,), generator=g, device=d) * 2.0 - 1.0, torch.rand((m,), generator=g, device=d),
                torch.randint(0, 10, (m,), generator=g, device=d, dtype=torch.int32),
                torch.randint(0, 3, (m,), generator=g, device=d, dtype=torch.int32))


def league_actions(mapper, assignment, modules, obs):
    """One batched forward per module over the (market, slot) pairs it plays.

    modules: {module id: callable(obs [M, obs_dim]) -> (category i32[M], size_mean f32[M], size_sigma f32[M],
    price i32[M], price_offset i32[M])}; obs: [N, obs_dim] (all slots of a market see the same vector).
    Returns the env's five [N, A] action tensors."""
    import torch
    n, A = assignment.shape
    dev = obs.device
    out = (torch.zeros((n, A), dtype=torch.int32, device=dev), torch.zeros((n, A), dtype=torch.float32, device=dev),
           torch.zeros((n, A), dtype=torch.float32, device=dev), torch.zeros((n, A), dtype=torch.int32, device=dev),
           torch.zeros((n, A), dtype=torch.int32, device=dev))
    for name, (mk, sl) in mapper.group_by_module(assignment).items():
        if name not in modules:
            raise KeyError(f"no module registered for '{name}'")
        mk_t, sl_t = torch.as_tensor(mk, device=dev), torch.as_tensor(sl, device=dev)
        acts = modules[name](obs.index_select(0, mk_t))
        for dst, src in zip(out, acts):
            dst[mk_t, sl_t] = src.to(dst.dtype)
    return out
