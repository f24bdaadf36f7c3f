"""Loads the HIP shared library (libcda_hip.so, built in-tree by __graft_entry__.build()) and binds
the C-ABI of include/cda.h with ctypes.  There is no CPU fallback: a missing library or a missing
GPU raises."""
import ctypes as C
import os

from . import _capi as K

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDA_HIP_LIB lets the tuning scripts load an experimental build of the SAME HIP library (never a CPU path)
LIB_PATH = os.environ.get("CDA_HIP_LIB") or os.path.join(_HERE, "libcda_hip.so")

_lib = None

# every symbol include/cda.h declares
SYMBOLS = [
    "cda_default_config", "cda_create", "cda_destroy", "cda_reset", "cda_step", "cda_place_order",
    "cda_mark_to_mkt", "cda_get_state", "cda_set_state", "cda_get_raw_snapshot", "cda_last_flags",
    "cda_selftest_dec", "cda_selftest_rng", "cda_strerror", "cda_num_markets", "cda_obs_dim",
    "cda_state_bytes_per_market", "cda_run_random", "cda_random_actions_host", "cda_nav_conservation",
    "cda_step_range", "cda_reset_range", "cda_step_groups", "cda_group_range", "cda_random_actions", "cda_book_peak", "cda_check_invariants", "cda_selftest_libm", "cda_selftest_libm_host", "cda_book_capacity",
    "cda_get_book", "cda_book_spill", "cda_book_spill_wanted", "cda_num_agents", "cda_handback_stride", "cda_set_handback", "cda_set_handback_geometry", "cda_handback_unpack", "cda_ppo_loss", "cda_policy_sample", "cda_gae", "cda_store_slots", "cda_step_groups_handback", "cda_handback_groups",
    "cda_step_range_capture", "cda_policy_step_supported", "cda_policy_step_advised", "cda_policy_step_range",
    "cda_episode_metrics_enable", "cda_episode_metrics_collect",
]


# every symbol include/cda_mlp.h declares
MLP_SYMBOLS = [
    "cda_mlp_tile_rows", "cda_mlp_permutation", "cda_mlp_pack", "cda_mlp_policy_step", "cda_mlp_forward", "cda_mlp_prep_rows", "cda_mlp_forward_train", "cda_mlp_backward",
    "cda_mlp_wgrad", "cda_mlp_adam", "cda_ppo_loss32", "cda_gae_records", "cda_ppo_loss_records", "cda_mlp_forward_backward", "cda_mlp_rollout_chain", "cda_mlp_selftest_mfma",
    "cda_mlp_reduce", "cda_mlp_apply", "cda_gae_records_bootstrap", "cda_mlp_values", "cda_mlp_values_counted", "cda_episode_returns", "cda_mlp_league_step", "cda_mlp_league_rollout_chain",
    "cda_gae_records_league", "cda_league_assign", "cda_mlp_wgrad_jobs",
]
# the same entry points compiled for other history depths carry the suffix _h<H> (include/cda_mlp.h CDA_MLP_HIST_VARIANTS, csrc/cda_mlp_variant.h)
MLP_HIST_VARIANTS = (1, 2, 3, 6, 7, 8)


class RolloutBufs(C.Structure):
    """cda_rollout_bufs (include/cda_mlp.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("obs", "category", "size_mean", "size_sigma", "price", "price_offset", "a_cont", "logp", "value", "reward",
                                          "terminated", "truncated", "record", "dist", "info_steps", "fin_index", "fin_obs", "fin_count")] + [("fin_cap", C.c_int32), ("counter_bump", C.c_void_p)]


class League(C.Structure):
    """cda_league (include/cda_mlp.h)"""
    _fields_ = [("wb_bank", C.c_void_p), ("theta_bank", C.c_void_p), ("slot_net", C.c_void_p), ("n_nets", C.c_int32), ("n_trainable", C.c_int32), ("random_seed", C.c_uint64)]


class PpoExtra(C.Structure):
    """cda_ppo_extra (include/cda_mlp.h)"""
    _fields_ = [("rec_stride", C.c_int32), ("kl_coef", C.c_float), ("vf_clip", C.c_float), ("dist_old", C.c_void_p), ("log_std_old", C.c_void_p), ("sd_log_std", C.c_int32)]


class CDAError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CDAError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    # PyTorch-ROCm ships its own HIP runtime.  It must be in the process BEFORE this library resolves libamdhip64: loaded the
    # other way round, the system runtime this library would pull in and torch's copy both initialise, and hipGetDeviceCount
    # then reports no device (seen as "cda_create failed (-2)" when build() ran before smoke() in one interpreter).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.cda_default_config.argtypes = [C.POINTER(K.Config)]
    L.cda_create.argtypes = [C.POINTER(K.Config), i32, i32, C.POINTER(vp)]
    L.cda_destroy.argtypes = [vp]
    L.cda_reset.argtypes = [vp, vp, vp, vp, vp]
    L.cda_step.argtypes = [vp] + [vp] * 6 + [vp] * 4 + [C.POINTER(K.InfoPtrs), vp]
    L.cda_step_range.argtypes = [vp, i32, i32] + [vp] * 6 + [vp] * 4 + [C.POINTER(K.InfoPtrs), vp]
    L.cda_reset_range.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.cda_step_groups.argtypes = [vp, i32] + [vp] * 6 + [vp] * 4 + [C.POINTER(K.InfoPtrs), C.POINTER(vp)]
    L.cda_group_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.cda_group_range.restype = None
    L.cda_random_actions.argtypes = [u64, u64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.cda_book_peak.argtypes = [vp, vp, vp]
    L.cda_check_invariants.argtypes = [vp, vp, vp]
    L.cda_selftest_libm.argtypes = [i32, i32, i32, vp, vp]
    L.cda_selftest_libm_host.argtypes = [i32, i32, vp, vp]
    L.cda_run_random.argtypes = [vp, i32, u64, u64, vp, vp, vp, vp, vp, vp]
    L.cda_random_actions_host.argtypes = [u64, u64, i32, i32, i32, vp, vp, vp, vp, vp]
    L.cda_nav_conservation.argtypes = [vp, C.c_double, vp, vp, vp]
    L.cda_place_order.argtypes = [vp] + [i32] * 6
    L.cda_mark_to_mkt.argtypes = [vp, i32]
    L.cda_get_state.argtypes = [vp, i32, C.POINTER(K.MarketState)]
    L.cda_set_state.argtypes = [vp, i32, C.POINTER(K.MarketState)]
    L.cda_get_raw_snapshot.argtypes = [vp, vp, vp]
    L.cda_last_flags.argtypes = [vp, vp, vp]
    L.cda_selftest_dec.argtypes = [i32, i32, i32, vp, vp, vp]
    L.cda_selftest_rng.argtypes = [i32, u64, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.cda_strerror.argtypes = [C.c_int]
    L.cda_strerror.restype = C.c_char_p
    L.cda_num_markets.argtypes = [vp]
    L.cda_num_agents.argtypes = [vp]
    L.cda_book_capacity.argtypes = [vp]
    L.cda_book_spill.argtypes = [vp]
    L.cda_book_spill_wanted.argtypes = [vp]
    L.cda_policy_sample.argtypes = [vp, i32, vp, vp, i64, i32, u64, vp] + [vp] * 10 + [vp]
    L.cda_gae.argtypes = [vp, vp, vp, vp, i32, i64, C.c_float, C.c_float, vp, vp, vp]
    L.cda_ppo_loss.argtypes = [vp] * 11 + [i64, i32, i32, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp, vp]
    L.cda_store_slots.argtypes = [i32, vp, vp, vp, vp, i32, vp]
    L.cda_step_groups_handback.argtypes = [vp, i32] + [vp] * 6 + [vp] * 4 + [C.POINTER(K.InfoPtrs), C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(vp)] + [vp] * 4
    L.cda_handback_groups.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), i32, C.POINTER(vp)] + [vp] * 4
    L.cda_handback_stride.argtypes = [i32]
    L.cda_set_handback.argtypes = [vp, vp]
    L.cda_handback_unpack.argtypes = [vp, i32, i32, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp]
    L.cda_set_handback_geometry.argtypes = [vp, i64, i64]
    L.cda_get_book.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.cda_obs_dim.argtypes = [vp]
    L.cda_state_bytes_per_market.argtypes = [vp]
    L.cda_state_bytes_per_market.restype = i64
    f32 = C.c_float
    L.cda_mlp_tile_rows.argtypes = []
    L.cda_mlp_tile_rows.restype = i32
    L.cda_mlp_wgrad_jobs.argtypes = []
    L.cda_mlp_wgrad_jobs.restype = i32
    L.cda_mlp_pack.argtypes = [vp, vp, vp]
    L.cda_mlp_permutation.argtypes = [u64, i64, vp, vp]
    L.cda_mlp_policy_step.argtypes = [vp, vp, vp, i32, i32, i32, u64, vp, i64] + [vp] * 8 + [vp]
    L.cda_mlp_forward.argtypes = [vp, vp, vp, i64, i64, vp, vp]
    L.cda_mlp_prep_rows.argtypes = [vp, vp, i64, vp, vp, vp]
    L.cda_mlp_forward_train.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.cda_mlp_backward.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.cda_mlp_wgrad.argtypes = [vp] * 6 + [i64, i32, vp, vp]
    L.cda_mlp_adam.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i64, f32, f32, f32, vp, f32, f32, f32, f32, f32, vp, vp, vp]
    L.cda_mlp_reduce.argtypes = [vp, i32, vp, i32, vp, i64, f32, f32, f32, vp, vp, vp, vp, vp]
    L.cda_mlp_apply.argtypes = [vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, vp, vp]
    L.cda_gae_records_bootstrap.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, f32, f32, f32, vp, vp, i64, vp, vp, vp]
    L.cda_gae_records_league.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, f32, f32, f32, vp, vp, vp]
    L.cda_mlp_values.argtypes = [vp, vp, i32, vp, i64, vp, i64, vp]
    L.cda_mlp_values_counted.argtypes = [vp, vp, i32, vp, i64, vp, vp, i64, vp]
    L.cda_episode_returns.argtypes = [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp]
    L.cda_mlp_league_step.argtypes = [C.POINTER(League), vp, i32, i32, i32, u64, vp, i64] + [vp] * 8 + [i64, vp, vp, i64, vp]
    L.cda_mlp_league_rollout_chain.argtypes = [vp, C.POINTER(League), i32, i32, i32, u64, vp, C.POINTER(RolloutBufs), i32, vp]
    L.cda_league_assign.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp]
    L.cda_step_range_capture.argtypes = [vp, i32, i32] + [vp] * 6 + [vp] * 4 + [C.POINTER(K.InfoPtrs), vp, i32, vp, vp, vp]
    L.cda_policy_step_supported.argtypes = [vp]
    L.cda_policy_step_advised.argtypes = [vp]
    L.cda_policy_step_range.argtypes = [vp, i32, i32, vp, vp, vp, u64, vp, i64] + [vp] * 5 + [vp] * 5 + [vp] * 4 + [vp, i32, vp, vp, vp]
    L.cda_episode_metrics_enable.argtypes = [vp, i32, C.c_double]
    L.cda_episode_metrics_collect.argtypes = [vp, vp, i32, vp, vp, i32, vp]
    L.cda_ppo_loss32.argtypes = [vp] * 10 + [i64, i32, i32, f32, f32, f32, vp, vp, vp, i64, i32, i32, vp]
    L.cda_gae_records.argtypes = [vp, vp, vp, vp, i32, i64, i32, f32, f32, f32, vp, vp, vp]
    L.cda_ppo_loss_records.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, f32, f32, f32, vp, vp, vp, i64, i32, i32, vp]
    L.cda_mlp_forward_backward.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, i64, i32, f32, f32, f32, C.POINTER(PpoExtra), vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.cda_mlp_rollout_chain.argtypes = [vp, vp, vp, i32, i32, i32, u64, vp, C.POINTER(RolloutBufs), i32, vp]
    L.cda_mlp_selftest_mfma.argtypes = [i32, vp, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("cda_strerror", "cda_group_range"):
            fn.restype = C.c_int
    for h in MLP_HIST_VARIANTS:                              # same signatures, other observation width
        for name in MLP_SYMBOLS:
            base, var = getattr(L, name), getattr(L, f"{name}_h{h}")
            var.argtypes, var.restype = base.argtypes, base.restype
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().cda_strerror(rc).decode()
        raise CDAError(f"{what} failed ({rc}): {msg}")
