"""What the reference's callback reports around an episode, from the device-side accumulators (include/cda.h cda_episode_metrics_*).

Reference: train/callbk/league_based_self_play_callback.py - `_log_activity` :295-342 (pass / rejection fractions), `_log_maker_ratio` :344-372,
`_log_reward_terms` :374-416 (mean and variance share of the five reward terms), `_log_episode_account` :418-470 (NAV mean / min / max, drawdown, inventory,
trades at the episode's last step), `on_episode_end` :627-755 (sum of NAV == num_agents x init_cash, `nav_conservation_violations`), and the driver's half of
the check, train/train.py:1109-1164 (`_check_nav_conservation`: a violation stops a `strict_nav_check` run with NavConservationError).

The reference logs one value per EPISODE and lets RLlib's MetricsLogger average a window of ten; a batch of N markets ends N episodes at once, so the figures
here are the means over the episodes that ended since the last collection (sums of the per-episode numerators and denominators where the reference's own
statistic is a ratio of sums, e.g. the fractions of one episode)."""
import numpy as np

from . import _capi as K

REWARD_TERMS = ("nav", "order", "trade", "drawdown", "passive")      # reward_helper.py:75-81, the order of info["reward_terms"]


class NavConservationError(AssertionError):
    """A run stopped because the NAV conservation invariant broke (train/train.py:1100-1107)."""


def _row(r, terms=REWARD_TERMS):
    n = float(r[K.EM_EPISODES])
    if n <= 0:
        return None
    steps = float(r[K.EM_AGENT_STEPS])
    out = {"agent_episodes": n, "agent_steps": steps}
    if steps > 0:
        out["pass_action_fraction"] = float(r[K.EM_PASSES]) / steps
        out["order_rejection_fraction"] = float(r[K.EM_REJECTIONS]) / steps
        out["orders_placed_fraction"] = float(r[K.EM_PLACED]) / steps
        var = {}
        for j, t in enumerate(terms):
            mean = float(r[K.EM_TERM_SUM + j]) / steps
            var[t] = max(0.0, float(r[K.EM_TERM_SQ + j]) / steps - mean * mean)          # E[x^2] - E[x]^2, clamped (:399-404)
            out[f"reward_term_mean_{t}"] = mean
        total = sum(var.values())
        if total > 0.0:
            for t in terms:
                out[f"reward_term_var_share_{t}"] = var[t] / total
    trades = float(r[K.EM_TRADES])
    out["trades"], out["passive_fills"] = trades, float(r[K.EM_PASSIVE])
    mean_ret = float(r[K.EM_RETURN_SUM]) / n
    out["episode_return_mean"] = mean_ret
    out["episode_return_std"] = max(0.0, float(r[K.EM_RETURN_SQ]) / n - mean_ret * mean_ret) ** 0.5
    out["episode_nav_mean"], out["episode_nav_min"], out["episode_nav_max"] = float(r[K.EM_NAV_SUM]) / n, float(r[K.EM_NAV_MIN]), float(r[K.EM_NAV_MAX])
    out["mean_agent_drawdown"] = float(r[K.EM_DRAWDOWN_SUM]) / n
    out["mean_abs_net_position"] = float(r[K.EM_ABS_POSITION_SUM]) / n
    out["mean_num_trades"] = float(r[K.EM_NUM_TRADES_SUM]) / n
    if r[K.EM_MAKER_RATIO_N] > 0:
        out["maker_fill_ratio_mean"] = float(r[K.EM_MAKER_RATIO_SUM]) / float(r[K.EM_MAKER_RATIO_N])
        out["maker_fill_ratio_best"] = float(r[K.EM_MAKER_RATIO_MAX])
    out["bankrupt_fraction"] = float(r[K.EM_BANKRUPT]) / n
    return out


def summarise(agent_table, env_row, module_names=None):
    """agent_table f64 [n_modules, EM_AGENT_FIELDS], env_row f64 [EM_ENV_FIELDS] (device or host) -> {"episodes", "nav_conservation_violations",
    "nav_conservation_error", ..., "all": {...the callback's per-episode metrics over every agent...}, "modules": {name: {...}}}"""
    t = agent_table.detach().cpu().numpy() if hasattr(agent_table, "detach") else np.asarray(agent_table)
    e = env_row.detach().cpu().numpy() if hasattr(env_row, "detach") else np.asarray(env_row)
    out = {"episodes": float(e[K.EM_ENV_EPISODES]), "nav_conservation_violations": float(e[K.EM_ENV_NAV_VIOLATIONS]),
           "nav_conservation_error": float(e[K.EM_ENV_NAV_ERROR_MAX]), "nav_conservation_error_sum": float(e[K.EM_ENV_NAV_ERROR_SUM])}
    if e[K.EM_ENV_EPISODES] > 0:
        out["episode_len_mean"] = float(e[K.EM_ENV_STEPS]) / float(e[K.EM_ENV_EPISODES])
        out["terminated_fraction"] = float(e[K.EM_ENV_TERMINATED]) / float(e[K.EM_ENV_EPISODES])
    if e[K.EM_ENV_MAKER_MAX_N] > 0:
        out["maker_fill_ratio_max"] = float(e[K.EM_ENV_MAKER_MAX_SUM]) / float(e[K.EM_ENV_MAKER_MAX_N])      # mean over the episodes of their most maker-like agent (:344-372)
    tot = t.sum(axis=0)                                                   # every column is a sum, except:
    played = t[:, K.EM_EPISODES] > 0
    if played.any():
        tot[K.EM_NAV_MIN], tot[K.EM_NAV_MAX] = t[played, K.EM_NAV_MIN].min(), t[played, K.EM_NAV_MAX].max()
        tot[K.EM_MAKER_RATIO_MAX] = t[played, K.EM_MAKER_RATIO_MAX].max()
    allr = _row(tot)
    if allr is not None:
        out["all"] = allr
    names = list(module_names) if module_names is not None else [f"module_{i}" for i in range(t.shape[0])]
    mods = {}
    for i in range(min(t.shape[0], len(names))):
        r = _row(t[i])
        if r is not None:
            mods[names[i]] = r
    out["modules"] = mods
    return out


def check_nav_conservation(iteration, summary, strict=True, log=None):
    """train.py:1125-1164 `_check_nav_conservation`: a violated episode stops a strict run, wherever (and whenever inside the rollout) it ended."""
    v = summary.get("nav_conservation_violations", 0.0)
    if v <= 0:
        return
    msg = (f"NAV conservation was violated in {v:.0f} episode(s) during iteration {iteration} (largest |sum(NAV) - A * init_cash| = "
           f"{summary.get('nav_conservation_error', float('nan')):g}). The ledger has created or destroyed cash, so every reward computed from NAV after this point is "
           f"meaningless. The markets concerned carry CDA_FLAG_NAV_CONSERVATION (CDAVecEnv.flags()).")
    if not strict:
        if log is not None:
            log(msg + " Continuing: strict_nav_check is off.")
        return
    raise NavConservationError(msg)
