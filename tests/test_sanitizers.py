"""Sanitizers where they can run (GPU AddressSanitizer is not available on this pool): the CPU oracle under
AddressSanitizer + UBSan (`make -C oracle check-asan-quick`; `check-asan` is the full three-minute form), and the HOST half of the
C ABI library - argument validation, struct handling, error strings - under UBSan (`python -m skyjo_rl_amd.build --ubsan-host`).
Both run in child processes: a sanitizer report aborts the child, which fails the test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_HOST_CALLS = r'''
import ctypes as C, sys
sys.path.insert(0, {root!r})
import numpy as np
from skyjo_rl_amd import _lib
L = _lib.load()
assert L.skyjo_vec_abi_version() == _lib.ABI_VERSION
h = C.c_void_p()
def cfg(**kw):
    d = dict(abi=_lib.ABI_VERSION, envs=64, players=3, ind=1, pen=2.0, mean=1.0, ref=0.001, ill=-1.0, dev=0, rng=0, auto=1, pad=0, gid=0)
    d.update(kw)
    return _lib.Config(d["abi"], d["envs"], d["players"], d["ind"], d["pen"], d["mean"], d["ref"], d["ill"], d["dev"], d["rng"], d["auto"], d["pad"], d["gid"])
# every refusal path of skyjo_vec_create that is decided on the host (no GPU here: the last one is the missing device itself)
for c in (cfg(abi=1), cfg(envs=0), cfg(envs=-5), cfg(players=0), cfg(players=13), cfg(rng=7), cfg()):
    rc = L.skyjo_vec_create(C.byref(c), C.byref(h))
    assert rc != 0 and not h.value and L.skyjo_vec_last_error()
assert L.skyjo_vec_create(None, C.byref(h)) != 0
# null handles / null arguments everywhere
for name, args in (("skyjo_vec_seed", (None, None, 0, None)), ("skyjo_vec_step", (None, None, None, None)), ("skyjo_vec_rollout", (None, 4, 0, None, None, None)),
                   ("skyjo_vec_observe", (None, None, None, None)), ("skyjo_vec_get_counters", (None, None, None)), ("skyjo_vec_get_state", (None, 0, None, None)),
                   ("skyjo_vec_set_option", (None, 1, 5)), ("skyjo_vec_check_error", (None, None)), ("skyjo_vec_snapshot_create", (None, None, None))):
    assert getattr(L, name)(*args) != 0, name
assert L.skyjo_vec_destroy(None) == 0 and L.skyjo_vec_mlp_destroy(None) == 0 and L.skyjo_vec_snapshot_destroy(None) == 0
# the scoring helpers validate their arguments before they look for a device
cards = np.zeros((2, 12), dtype=np.int8); won = np.array([5], dtype=np.int32); out = np.zeros(2)
assert L.skyjo_vec_evaluate_game(0, 1, 2, cards.ctypes.data, won.ctypes.data, 2.0, out.ctypes.data) != 0
assert L.skyjo_vec_evaluate_game(0, 1, 13, cards.ctypes.data, won.ctypes.data, 2.0, out.ctypes.data) != 0
assert L.skyjo_vec_calc_final_rewards(0, -1, 2, out.ctypes.data, won.ctypes.data, 1.0, 0.0, out.ctypes.data) != 0
# packing the weights of a policy net is host work: bad shapes are refused, a device is only needed at the very end
w = np.zeros(256 * 256, dtype=np.float32); m = C.c_void_p()
for od, out_dim, prec in ((0, 26, 0), (40, 26, 0), (31, 0, 0), (31, 64, 0), (31, 26, 9)):
    assert L.skyjo_vec_mlp_create(0, od, out_dim, prec, w.ctypes.data, w.ctypes.data, w.ctypes.data, w.ctypes.data, w.ctypes.data, w.ctypes.data, C.byref(m)) != 0
print("HOST-HALF-OK")
'''


def test_oracle_under_address_and_ub_sanitizer():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "check-asan-quick"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert " passed" in out.stdout and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_host_half_of_the_c_abi_under_ubsan():
    from skyjo_rl_amd import build

    rt = build.ubsan_runtime()
    if rt is None:
        import pytest
        pytest.skip("clang's shared UBSan runtime is not in this image")
    so = build.build_ubsan_host()
    # (the child sees no GPU wherever the suite runs - the last refusal it checks is the missing device itself - and sanitizers
    # run on the host half only: GPU sanitizers are not available on this pool)
    env = dict(os.environ, SKYJO_LIB=so, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", LD_PRELOAD=rt,
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", _HOST_CALLS.format(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "HOST-HALF-OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr
