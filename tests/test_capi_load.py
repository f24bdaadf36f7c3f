"""CPU: the C-ABI library loads and exports every symbol include/cda.h declares; the ctypes struct
mirror has the C layout; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cda.h")
MLP_HEADER = os.path.join(ROOT, "include", "cda_mlp.h")


def _declared_symbols(header=HEADER):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cda_[a-z_0-9]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def hip_lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_hip()
    from gym_continuousdoubleauction_amd import _lib
    return _lib.lib(), _lib


def test_every_declared_symbol_is_exported(hip_lib):
    L, _lib = hip_lib
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/cda.h but not exported by libcda_hip.so"
    assert sorted(_lib.SYMBOLS) == declared
    # ... and every symbol of the network's header (include/cda_mlp.h)
    declared_mlp = _declared_symbols(MLP_HEADER)
    assert len(declared_mlp) >= 10
    for name in declared_mlp:
        assert hasattr(L, name), f"{name} declared in include/cda_mlp.h but not exported by libcda_hip.so"
    assert sorted(_lib.MLP_SYMBOLS) == declared_mlp


def test_ctypes_layout_matches_c(tmp_path):
    from gym_continuousdoubleauction_amd import _capi as K
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(cda_config),sizeof(cda_dec),sizeof(cda_info_ptrs),sizeof(cda_order),sizeof(cda_account_state),'
                   'sizeof(cda_market_state),offsetof(cda_market_state,acc),offsetof(cda_market_state,hist));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(K.Config), C.sizeof(K.Dec), C.sizeof(K.InfoPtrs), C.sizeof(K.Order), C.sizeof(K.AccountState),
            C.sizeof(K.MarketState), K.MarketState.acc.offset, K.MarketState.hist.offset]
    assert got == want


def test_default_config_matches_reference_defaults(hip_lib):
    L, _ = hip_lib
    from gym_continuousdoubleauction_amd import _capi as K
    c = K.Config()
    assert L.cda_default_config(C.byref(c)) == 0
    d = K.ENV_DEFAULTS     # config/env_defaults.json:8-27 of the reference
    assert (c.num_agents, c.max_step, c.n_hist, c.tick_size, c.init_cash) == (d["num_of_agents"], d["max_step"], d["n_hist"], 1, d["init_cash"])
    assert (c.initial_price_min, c.initial_price_max, c.min_size, c.mkt_max_size, c.limit_size_multiple) == (10, 100, 1, 100, 10)
    assert (c.order_penalty, c.trade_penalty, c.drawdown_penalty, c.passive_bonus, c.loss_multiplier) == (0.1, 0.05, 0.2, 0.1, 1.5)
    assert L.cda_strerror(-2).decode().startswith("no HIP device")


def test_no_gpu_means_loud_failure_not_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L, _lib = hip_lib
    from gym_continuousdoubleauction_amd import _capi as K
    cfg, _ = K.make_config({"num_of_agents": 4})
    h = C.c_void_p()
    rc = L.cda_create(C.byref(cfg), 8, 0, C.byref(h))
    assert rc == K.ERR_NO_DEVICE and not h.value
    from gym_continuousdoubleauction_amd import CDAVecEnv, CDAEnv
    with pytest.raises(_lib.CDAError):
        CDAVecEnv({"num_of_agents": 4}, n_markets=8)
    with pytest.raises(_lib.CDAError):
        CDAEnv({"num_of_agents": 4})


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, load or mention it."""
    pkg = os.path.join(ROOT, "gym_continuousdoubleauction_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libcda_oracle" not in txt and "cda_oracle" not in txt, f


def test_product_library_has_no_measuring_probes():
    """The decimal micro-benchmark, the clock probe and the PMC calibration kernels live in tools/libcda_tools.so:
    the product library exports none of their entry points and holds none of their kernels."""
    import subprocess
    from gym_continuousdoubleauction_amd import _lib
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "cda_debug" not in syms, [ln for ln in syms.splitlines() if "cda_debug" in ln]
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"k_opbench", b"k_clock_probe", b"k_calib_read", b"k_calib_write"):
        assert name not in blob, name


def test_learner_side_entry_points_refuse_bad_arguments_before_touching_the_device(hip_lib):
    """cda_ppo_loss / cda_policy_sample / cda_gae / cda_store_slots validate their arguments first: NULL pointers, empty batches, more
    samples per row than agents exist, a packed stride that cannot hold logits | value - CDA_ERR_INVALID, no launch (no GPU needed)."""
    L, _ = hip_lib
    INVALID = -1
    one = C.c_void_p(16)                                          # a non-NULL pointer that is never dereferenced: validation fails first
    args_loss = lambda rows, per_row, stride, logits=one, value=one: L.cda_ppo_loss(   # noqa: E731
        logits, value, one, one, one, one, one, one, one, one, None, rows, per_row, stride, 0.2, 0.5, 0.01, one, one, one, one, None)
    assert args_loss(0, 1, 0) == INVALID and args_loss(8, 0, 0) == INVALID and args_loss(8, 17, 0) == INVALID
    assert args_loss(8, 4, 24) == INVALID and args_loss(8, 4, 30) == INVALID            # packed rows need more than 24 columns, a multiple of 4
    assert args_loss(8, 4, 0, value=None) == INVALID and args_loss(8, 4, 0, logits=None) == INVALID
    samp = lambda stride, value_out, rows=8, per_row=4: L.cda_policy_sample(one, stride, value_out, one, rows, per_row, 1, one, *([one] * 10), None)   # noqa: E731
    assert samp(20, None) == INVALID and samp(26, None) == INVALID and samp(24, one) == INVALID and samp(32, None, rows=0) == INVALID
    assert samp(32, None, per_row=17) == INVALID
    assert L.cda_gae(one, one, one, one, 0, 8, 0.99, 0.95, one, one, None) == INVALID and L.cda_gae(None, one, one, one, 4, 8, 0.99, 0.95, one, one, None) == INVALID
    src, dst, nb = (C.c_void_p * 1)(16), (C.c_void_p * 1)(16), (C.c_int64 * 1)(64)
    assert L.cda_store_slots(0, src, dst, nb, one, 0, None) == INVALID and L.cda_store_slots(13, src, dst, nb, one, 0, None) == INVALID
    assert L.cda_store_slots(1, src, dst, (C.c_int64 * 1)(0), one, 0, None) == INVALID and L.cda_store_slots(1, src, dst, nb, None, 0, None) == INVALID
    # the policy inside the step kernel (include/cda.h): no env / missing arrays are refused before anything is launched
    assert L.cda_policy_step_supported(None) == 0
    assert L.cda_policy_step_range(None, 0, 8, one, one, one, 1, one, 0, *([one] * 5), *([one] * 5), *([one] * 4), None, 0, None, None, None) == INVALID


def test_every_network_entry_point_exists_for_every_compiled_history_depth():
    """include/cda_mlp.h: the unsuffixed entry points are n_hist = 4; CDA_MLP_HIST_VARIANTS names the other depths the library holds them for, as <name>_h<H>.  The
    rename list (csrc/cda_mlp_variant.h) must cover every function the header declares, and the built library must export every one of them."""
    import ctypes as C
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "cda_mlp.h")).read()
    variants = [int(x) for x in re.search(r'#define CDA_MLP_HIST_VARIANTS "([0-9 ]+)"', hdr).group(1).split()]
    declared = set(re.findall(r"^(?:int|int32_t)\s+(cda_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 25 and variants == [1, 2, 3, 6, 7, 8]
    renames = set(re.findall(r"#define (cda_[a-z0-9_]+) CDA_MLP_SFX\(\1\)", open(os.path.join(root, "gym_continuousdoubleauction_amd", "csrc", "cda_mlp_variant.h")).read()))
    assert renames == declared, (sorted(declared - renames), sorted(renames - declared))
    from gym_continuousdoubleauction_amd import _lib
    assert set(_lib.MLP_SYMBOLS) == declared and tuple(variants) == tuple(_lib.MLP_HIST_VARIANTS)
    so = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        for h in variants:
            getattr(so, f"{name}_h{h}")
    import __graft_entry__ as G
    assert tuple(G.MLP_HIST_VARIANTS) == tuple(variants)


def test_network_entry_points_refuse_bad_arguments_before_touching_the_device(hip_lib):
    """include/cda_mlp.h: every entry point of the network / league / learner side, in every compiled history depth, validates first - all-NULL arguments and
    out-of-range sizes return CDA_ERR_INVALID with no launch (this test runs without a GPU), so a binding bug on the consumer's side is an error code, not a fault."""
    from gym_continuousdoubleauction_amd import _lib
    L, _ = hip_lib
    INVALID = -1

    def zero(t):
        if t is C.c_void_p or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return None
        return 0.0 if t in (C.c_float, C.c_double) else 0
    for name in _lib.MLP_SYMBOLS:
        for sfx in [""] + [f"_h{h}" for h in _lib.MLP_HIST_VARIANTS]:
            fn = getattr(L, name + sfx)
            if not fn.argtypes:
                assert fn() > 0                                    # (cda_mlp_tile_rows, cda_mlp_wgrad_jobs: constants of the build)
                continue
            assert fn(*[zero(t) for t in fn.argtypes]) == INVALID, name + sfx
    one = C.c_void_p(64)                                          # non-NULL, never dereferenced: the size checks fail first
    for sfx in ("", "_h8"):
        fb = getattr(L, "cda_mlp_forward_backward" + sfx)
        call_fb = lambda rows, agents, extra=None, adv=None, adv_n=0, finish=0, out6=one: fb(   # noqa: E731
            one, one, one, one, rows, 0, one, adv, adv_n, agents, 0.2, 0.5, 0.01, extra, one, one, one, one, one, one, one, one, out6, 1, finish, None, None, None)
        assert call_fb(0, 4) == INVALID and call_fb(48, 4) == INVALID and call_fb(64, 0) == INVALID and call_fb(64, 17) == INVALID
        assert call_fb(64, 4, adv=one, adv_n=1) == INVALID and call_fb(64, 4, finish=1, out6=None) == INVALID
        x = _lib.PpoExtra()
        x.rec_stride, x.kl_coef, x.vf_clip = 24, 0.0, 0.0          # a league record stride below agents * 8
        assert call_fb(64, 4, extra=C.pointer(x)) == INVALID
        x.rec_stride = 34                                          # not a multiple of 4 floats (the records are read as 16-byte pieces)
        assert call_fb(64, 4, extra=C.pointer(x)) == INVALID
        x.rec_stride, x.kl_coef = 0, 0.2                           # a KL penalty without the rollout's stored distributions
        assert call_fb(64, 4, extra=C.pointer(x)) == INVALID
        wg = getattr(L, "cda_mlp_wgrad" + sfx)
        assert wg(one, one, one, one, one, one, 64, 3, one, None) == INVALID and wg(one, one, one, one, one, one, 40, 1, one, None) == INVALID
        vals = getattr(L, "cda_mlp_values" + sfx)
        assert vals(one, one, 0, one, 64, one, 64, None) == INVALID and vals(one, one, 17, one, 64, one, 64, None) == INVALID and vals(one, one, 2, one, 0, one, 64, None) == INVALID
        assign = getattr(L, "cda_league_assign" + sfx)
        assert assign(one, 8, 4, 2, one, one, 0, one, None, None) == INVALID and assign(one, 0, 4, 2, one, one, 3, one, None, None) == INVALID
        assert assign(one, 8, 17, 2, one, one, 3, one, None, None) == INVALID and assign(one, 8, 4, 5, one, one, 3, one, None, None) == INVALID
        gl = getattr(L, "cda_gae_records_league" + sfx)
        assert gl(one, one, one, one, 8, 8, 4, 0, 1.0, 0.99, 0.95, one, one, None) == INVALID and gl(one, one, one, one, 8, 8, 4, 5, 1.0, 0.99, 0.95, one, one, None) == INVALID
        er = getattr(L, "cda_episode_returns" + sfx)
        assert er(one, one, one, 0, 8, 4, one, one, one, None, None) == INVALID and er(one, one, one, 8, 8, 17, one, one, one, None, None) == INVALID
        red = getattr(L, "cda_mlp_reduce" + sfx)
        assert red(one, 0, one, 1, None, 0, 0.5, 0.01, 0.0, None, one, one, one, None) == INVALID and red(one, 1, one, 1, one, 0, 0.5, 0.01, 0.0, None, one, one, one, None) == INVALID
