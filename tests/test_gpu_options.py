"""Round 6: what used to be environment variables of the library are options of skyjo_vec_set_option (include/skyjo_vec.h); the
shipped library reads no environment variable.  Results never depend on them."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu


def _get(e, opt):
    from skyjo_rl_amd import _lib
    v = C.c_int64()
    _lib.check(e._L.skyjo_vec_get_option(e._h, int(opt), C.byref(v)))
    return int(v.value)


def test_cycles_per_launch_and_workgroup_shape_do_not_change_results():
    import torch
    from skyjo_rl_amd import SkyjoNativeError, SkyjoVecEnv, _lib

    B, N = 4096, 3
    ref = None
    for cycles, s in ((16, 0), (1, 0), (3, 0), (16, 2), (16, 4)):
        e = SkyjoVecEnv(B, num_players=N, auto_reset=True)
        assert e.dealing_form() == "one kernel" and _get(e, _lib.OPT_CYCLE_S) == 1 and _get(e, _lib.OPT_MAX_CYCLES_PER_LAUNCH) == 16
        if s:
            e.set_option(_lib.OPT_CYCLE_S, s)
            assert _get(e, _lib.OPT_CYCLE_S) == s
        e.set_option(_lib.OPT_MAX_CYCLES_PER_LAUNCH, cycles)
        assert _get(e, _lib.OPT_MAX_CYCLES_PER_LAUNCH) == cycles
        e.set_deal_interval(32)
        e.seed(None, 5)
        K = 32 * 8
        rec, act = e.new_records(K), torch.empty((K, B), dtype=torch.int32, device="cuda")
        for _ in range(3):
            e.rollout(K, policy_seed=2, records=rec, actions=act)
        c = e.counters()
        out = (rec.clone(), act.clone(), {k: c[k] for k in ("steps", "episodes", "resets", "sum_len")})
        if ref is None:
            ref = out
            assert c["episodes"] > 0
        else:
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and out[2] == ref[2], (cycles, s)
        with pytest.raises(SkyjoNativeError):
            e.set_option(_lib.OPT_CYCLE_S, 2)  # after seeding
        with pytest.raises(SkyjoNativeError):
            e.set_option(_lib.OPT_MAX_CYCLES_PER_LAUNCH, 17)
        e.close()


def test_the_library_ignores_the_old_environment_variables(monkeypatch):
    from skyjo_rl_amd import SkyjoVecEnv, _lib

    for k, v in (("SKYJO_OVERLAP", "1"), ("SKYJO_MERGED", "0"), ("SKYJO_CYCLE_S", "3"), ("SKYJO_DEAL_INTERVAL", "7"), ("SKYJO_FUSED_SCAN", "0"),
                 ("SKYJO_PIPELINED", "0")):
        monkeypatch.setenv(k, v)
    if os.environ.get("SKYJO_LIB"):
        pytest.skip("a diagnostic build may be loaded")
    e = SkyjoVecEnv(4096, num_players=3)
    assert e.dealing_form() == "one kernel" and _get(e, _lib.OPT_CYCLE_S) == 1 and e.deal_interval() != 7
    assert _get(e, _lib.OPT_INLINE_WORK_LIST) == 0 and _get(e, _lib.OPT_UNPIPELINED) == 0
    e.close()
