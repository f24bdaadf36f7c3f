"""Test helper: the reference callback's per-episode tallies and end-of-episode figures (train/callbk/league_based_self_play_callback.py:541-755), computed on the
host from the CPU oracle's per-step info - what the device-side accumulators of include/cda.h cda_episode_metrics_* must equal."""
from decimal import Decimal

import numpy as np

from gym_continuousdoubleauction_amd import _capi as K


class OracleEpisodeMetrics:
    """feed(info, reward, terminated, truncated) after every oracle step (BEFORE the masked reset of the ended markets); table(module_of, n_modules)
    gives the (agent table, env row) cda_episode_metrics_collect returns for the episodes that ended so far."""

    def __init__(self, n_markets, num_agents, init_cash, nav_tolerance=1e-6):
        self.N, self.A, self.init_cash, self.tol = int(n_markets), int(num_agents), int(init_cash), Decimal(str(nav_tolerance))
        N, A = self.N, self.A
        self.term_sum, self.term_sq = np.zeros((N, A, 5)), np.zeros((N, A, 5))
        self.ret = np.zeros((N, A))
        self.cnt = np.zeros((N, A, 5), np.int64)            # passes, rejections, placed, trades, passive
        self.steps = np.zeros(N, np.int64)
        self.F = np.zeros((N, A, K.EM_AGENT_FIELDS))
        self.M = np.zeros((N, K.EM_ENV_FIELDS))
        self.violating = []                                 # (market, error) of the episodes that broke conservation

    def feed(self, info, reward, terminated, truncated, done_mask_of=None):
        """done_mask_of(market) -> the env's done_set as a bit mask (the oracle's get_state), for the episodes that end with this step"""
        rt = info["reward_terms"]
        self.term_sum += rt                                 # per (market, agent): the same additions in the same (step) order as the device's
        self.term_sq += rt * rt
        self.ret += reward
        self.cnt[..., 0] += info["is_pass_action"].astype(np.int64)
        self.cnt[..., 1] += info["num_rejected_step"]
        self.cnt[..., 2] += info["order_step_placed"]
        self.cnt[..., 3] += info["num_trades_step"]
        self.cnt[..., 4] += info["num_passive_fills_step"]
        self.steps += 1
        ended = np.nonzero(np.asarray(terminated, bool) | np.asarray(truncated, bool))[0]
        for i in ended:
            self._end(int(i), info, bool(terminated[i]), None if done_mask_of is None else int(done_mask_of(int(i))))
        return ended

    def _end(self, i, info, terminated, done_mask):
        A, F, M = self.A, self.F, self.M
        navs = [K.dec_to_decimal(info["nav"][i, a]) for a in range(A)]
        total = Decimal(0)
        for n in navs:
            total += n                                      # prec 28, the callback's own arithmetic (:679-704)
        err = abs(total - Decimal(self.init_cash) * A)
        viol = err > self.tol
        if viol:
            self.violating.append((i, err))
        best = -1.0
        for a in range(A):
            f = F[i, a]
            first = f[K.EM_EPISODES] == 0
            nav = float(navs[a])
            f[K.EM_EPISODES] += 1
            f[K.EM_AGENT_STEPS] += self.steps[i]
            f[K.EM_PASSES:K.EM_PASSES + 5] += self.cnt[i, a]
            f[K.EM_TERM_SUM:K.EM_TERM_SUM + 5] += self.term_sum[i, a]
            f[K.EM_TERM_SQ:K.EM_TERM_SQ + 5] += self.term_sq[i, a]
            f[K.EM_RETURN_SUM] += self.ret[i, a]
            f[K.EM_RETURN_SQ] += self.ret[i, a] * self.ret[i, a]
            f[K.EM_NAV_SUM] += nav
            f[K.EM_NAV_MIN] = nav if first else min(f[K.EM_NAV_MIN], nav)
            f[K.EM_NAV_MAX] = nav if first else max(f[K.EM_NAV_MAX], nav)
            f[K.EM_DRAWDOWN_SUM] += info["drawdown"][i, a]
            f[K.EM_ABS_POSITION_SUM] += abs(float(info["net_position"][i, a]))
            f[K.EM_NUM_TRADES_SUM] += info["num_trades"][i, a]
            trades, passive = int(self.cnt[i, a, 3]), int(self.cnt[i, a, 4])
            if trades >= 5:
                r = passive / trades
                f[K.EM_MAKER_RATIO_SUM] += r; f[K.EM_MAKER_RATIO_N] += 1; f[K.EM_MAKER_RATIO_MAX] = max(f[K.EM_MAKER_RATIO_MAX], r)
                best = max(best, r)
            f[K.EM_BANKRUPT] += (1.0 if navs[a] <= 0 else 0.0) if done_mask is None else float((done_mask >> a) & 1)      # done_set is sticky (done_helper.py:3-18)
        M[i, K.EM_ENV_EPISODES] += 1
        M[i, K.EM_ENV_NAV_VIOLATIONS] += 1.0 if viol else 0.0
        M[i, K.EM_ENV_NAV_ERROR_SUM] += float(err)
        M[i, K.EM_ENV_NAV_ERROR_MAX] = max(M[i, K.EM_ENV_NAV_ERROR_MAX], float(err))
        if best >= 0:
            M[i, K.EM_ENV_MAKER_MAX_SUM] += best; M[i, K.EM_ENV_MAKER_MAX_N] += 1
        M[i, K.EM_ENV_STEPS] += self.steps[i]
        M[i, K.EM_ENV_TERMINATED] += 1.0 if terminated else 0.0
        self.term_sum[i] = 0; self.term_sq[i] = 0; self.ret[i] = 0; self.cnt[i] = 0; self.steps[i] = 0

    def discard(self, mask):
        """a reset of markets whose episode is not over: their running tallies are dropped"""
        m = np.asarray(mask, bool)
        self.term_sum[m] = 0; self.term_sq[m] = 0; self.ret[m] = 0; self.cnt[m] = 0; self.steps[m] = 0

    def table(self, module_of=None, n_modules=1, clear=True):
        N, A = self.N, self.A
        mod = np.zeros((N, A), np.int64) if module_of is None else np.asarray(module_of).reshape(N, A)
        T = np.zeros((n_modules, K.EM_AGENT_FIELDS))
        for m in range(n_modules):
            rows = self.F[(mod == m) & (self.F[..., K.EM_EPISODES] > 0)]
            if len(rows):
                T[m] = rows.sum(0)
                T[m, K.EM_NAV_MIN], T[m, K.EM_NAV_MAX], T[m, K.EM_MAKER_RATIO_MAX] = rows[:, K.EM_NAV_MIN].min(), rows[:, K.EM_NAV_MAX].max(), rows[:, K.EM_MAKER_RATIO_MAX].max()
        E = self.M.sum(0)
        E[K.EM_ENV_NAV_ERROR_MAX] = self.M[:, K.EM_ENV_NAV_ERROR_MAX].max()
        if clear:
            self.F[:] = 0; self.M[:] = 0
        return T, E


EXACT_AGENT = [K.EM_EPISODES, K.EM_AGENT_STEPS, K.EM_PASSES, K.EM_REJECTIONS, K.EM_PLACED, K.EM_TRADES, K.EM_PASSIVE, K.EM_NAV_MIN, K.EM_NAV_MAX, K.EM_NUM_TRADES_SUM,
               K.EM_ABS_POSITION_SUM, K.EM_MAKER_RATIO_N, K.EM_MAKER_RATIO_MAX, K.EM_BANKRUPT]
EXACT_ENV = [K.EM_ENV_EPISODES, K.EM_ENV_NAV_VIOLATIONS, K.EM_ENV_NAV_ERROR_MAX, K.EM_ENV_MAKER_MAX_N, K.EM_ENV_STEPS, K.EM_ENV_TERMINATED]


def assert_tables_equal(dev_agent, dev_env, ref_agent, ref_env, what=""):
    """integer / decimal-derived fields bit for bit; the f64 sums (added in another order across markets) within 1e-12 relative"""
    da, de = np.asarray(dev_agent), np.asarray(dev_env)
    assert da.shape == ref_agent.shape, (da.shape, ref_agent.shape)
    for f in EXACT_AGENT:
        assert np.array_equal(da[:, f], ref_agent[:, f]), (what, "agent field", f, da[:, f], ref_agent[:, f])
    for f in EXACT_ENV:
        assert de[f] == ref_env[f], (what, "env field", f, de[f], ref_env[f])
    scale = np.maximum(np.abs(ref_agent), 1e-300)
    assert (np.abs(da - ref_agent) <= 1e-12 * scale + 1e-9 * (np.abs(ref_agent) < 1e-6)).all(), (what, np.abs(da - ref_agent).max(0))
    assert np.allclose(de, ref_env, rtol=1e-12, atol=0), (what, de, ref_env)
