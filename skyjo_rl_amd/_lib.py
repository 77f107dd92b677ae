"""ctypes loader for libskyjo_vec.so (the HIP / gfx950 engine).

The product has no CPU path: when the shared library is missing, cannot be loaded, or finds no
gfx950 device, every entry point raises.  Signatures mirror include/skyjo_vec.h.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKYJO_LIB") or os.path.join(_HERE, "libskyjo_vec.so")  # SKYJO_LIB: diagnostic builds

ABI_VERSION = 4
MAX_PLAYERS = 12
ST_OK, ST_ILLEGAL, ST_NOOP_DONE, ST_RESET, ST_ERROR = 0, 1, 2, 3, 4
RNG_MT19937, RNG_PHILOX = 0, 1
MLP_BF16, MLP_FP32 = 0, 1  # SKYJO_MLP_*: precision of a packed policy / value net
ACTION_SKIP = -1000  # SKYJO_ACTION_SKIP: leave this game as it is (skyjo_vec_step)
PROF_KERNELS = ("k_step", "k_scan", "k_deal", "k_publish", "k_mlp")


class SkyjoNativeError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("num_envs", C.c_int32), ("num_players", C.c_int32),
                ("observe_indirect", C.c_int32), ("score_penalty", C.c_double), ("mean_reward", C.c_double),
                ("reward_refunded", C.c_double), ("illegal_reward", C.c_double), ("device_id", C.c_int32),
                ("rng_mode", C.c_int32), ("auto_reset", C.c_int32), ("reserved", C.c_int32),
                ("game_id0", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_envs", "num_players", "obs_dim", "record_bytes", "mask_offset",
                                          "meta_offset", "state_bytes", "tile_games")]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("steps", "episodes", "illegal", "resets", "sum_len", "reshuffles",
                                           "iters", "waits")] + [
        ("sum_score", C.c_double * MAX_PLAYERS), ("sum_reward", C.c_double * MAX_PLAYERS),
        ("sum_reward_sq", C.c_double * MAX_PLAYERS), ("sum_refunded", C.c_double * MAX_PLAYERS)]


class GameState(C.Structure):
    _fields_ = [("players_cards", (C.c_int8 * 12) * MAX_PLAYERS), ("players_masked", (C.c_int8 * 12) * MAX_PLAYERS),
                ("drawpile", C.c_int8 * 150), ("discard_pile", C.c_int8 * 150), ("n_draw", C.c_int16),
                ("n_disc", C.c_int16), ("hand_card", C.c_int8), ("expected_player", C.c_uint8),
                ("expected_phase", C.c_uint8), ("is_terminated", C.c_uint8), ("done", C.c_uint8),
                ("status", C.c_uint8), ("episode_steps", C.c_uint16), ("episode", C.c_uint32),
                ("reshuffles", C.c_uint32), ("num_refunded", C.c_int32 * MAX_PLAYERS),
                ("num_placed", C.c_int32 * MAX_PLAYERS), ("final_score", C.c_double * MAX_PLAYERS),
                ("rewards", C.c_double * MAX_PLAYERS)]


# the same struct as a numpy record (aligned like the C compiler lays it out; tests/test_capi_symbols.py compares the sizes)
import numpy as _np  # noqa: E402

GAME_STATE_DTYPE = _np.dtype([
    ("players_cards", _np.int8, (MAX_PLAYERS, 12)), ("players_masked", _np.int8, (MAX_PLAYERS, 12)),
    ("drawpile", _np.int8, (150,)), ("discard_pile", _np.int8, (150,)), ("n_draw", _np.int16), ("n_disc", _np.int16),
    ("hand_card", _np.int8), ("expected_player", _np.uint8), ("expected_phase", _np.uint8), ("is_terminated", _np.uint8),
    ("done", _np.uint8), ("status", _np.uint8), ("episode_steps", _np.uint16), ("episode", _np.uint32),
    ("reshuffles", _np.uint32), ("num_refunded", _np.int32, (MAX_PLAYERS,)), ("num_placed", _np.int32, (MAX_PLAYERS,)),
    ("final_score", _np.float64, (MAX_PLAYERS,)), ("rewards", _np.float64, (MAX_PLAYERS,))], align=True)
assert GAME_STATE_DTYPE.itemsize == C.sizeof(GameState), (GAME_STATE_DTYPE.itemsize, C.sizeof(GameState))
for _f in ("n_draw", "hand_card", "episode_steps", "episode", "num_refunded", "final_score", "rewards"):
    assert GAME_STATE_DTYPE.fields[_f][1] == getattr(GameState, _f).offset, _f


class RolloutBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("records", "actions", "logp", "values", "final_rewards", "episode_end")]


# option ids / record layouts of include/skyjo_vec.h
OPT_RECORD_LAYOUT, REC_ROW_MAJOR, REC_TILE_PLANAR, REC_TILE_PLANAR_ALL = 6, 0, 1, 2
OPT_INLINE_WORK_LIST, OPT_UNPIPELINED, OPT_CYCLE_S, OPT_MAX_CYCLES_PER_LAUNCH = 7, 8, 9, 10

# name -> (restype, argtypes); this table is also what tests/test_capi_symbols.py checks against the header
VP, I32, I64, U64, U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
SIGNATURES = {
    "skyjo_vec_last_error": (C.c_char_p, []),
    "skyjo_vec_abi_version": (C.c_int, []),
    "skyjo_vec_create": (C.c_int, [C.POINTER(Config), C.POINTER(VP)]),
    "skyjo_vec_destroy": (C.c_int, [VP]),
    "skyjo_vec_get_info": (C.c_int, [VP, C.POINTER(Info)]),
    "skyjo_vec_seed": (C.c_int, [VP, VP, U64, VP]),
    "skyjo_vec_seed_one": (C.c_int, [VP, I32, U64, VP]),
    "skyjo_vec_reset": (C.c_int, [VP, VP, VP, VP]),
    "skyjo_vec_step": (C.c_int, [VP, VP, VP, VP]),
    "skyjo_vec_rollout": (C.c_int, [VP, I32, U64, VP, VP, VP]),
    "skyjo_vec_observe": (C.c_int, [VP, VP, VP, VP]),
    "skyjo_vec_unpack": (C.c_int, [VP, VP, I64, VP, VP, VP, VP, VP, VP, VP]),
    "skyjo_vec_unpack_tiles": (C.c_int, [VP, VP, I64, VP, VP, VP, VP, VP, VP, VP]),
    "skyjo_vec_sample_actions": (C.c_int, [VP, VP, VP, I64, U64, U64, I32, VP, VP, VP, VP]),
    "skyjo_vec_mlp_create": (C.c_int, [I32, I32, I32, I32, VP, VP, VP, VP, VP, VP, C.POINTER(VP)]),
    "skyjo_vec_mlp_destroy": (C.c_int, [VP]),
    "skyjo_vec_mlp_forward": (C.c_int, [VP, VP, I32, I64, VP, VP]),
    "skyjo_vec_mlp_act": (C.c_int, [VP, VP, VP, I64, U64, U64, I32, VP, VP, VP, VP]),
    "skyjo_vec_mlp_act_value": (C.c_int, [VP, VP, VP, VP, I64, U64, U64, I32, VP, VP, VP, VP, VP]),
    "skyjo_vec_episode_ends": (C.c_int, [VP, VP, VP, VP, VP]),
    "skyjo_vec_episode_ends_layout": (C.c_int, [VP, VP, I32, VP, VP, VP]),
    "skyjo_vec_sample_actions_layout": (C.c_int, [VP, VP, I32, VP, I64, U64, U64, I32, VP, VP, VP, VP]),
    "skyjo_vec_mlp_forward_layout": (C.c_int, [VP, VP, I32, I32, I64, VP, VP]),
    "skyjo_vec_mlp_act_value_layout": (C.c_int, [VP, VP, VP, VP, I32, I64, U64, U64, I32, VP, VP, VP, VP, VP]),
    "skyjo_vec_step_collect": (C.c_int, [VP, VP, VP, VP, VP, VP]),
    "skyjo_vec_model_rollout": (C.c_int, [VP, VP, VP, I32, U64, U64, I32, C.POINTER(RolloutBuffers), VP]),
    "skyjo_vec_rewards_ptr": (VP, [VP]),
    "skyjo_vec_scores_ptr": (VP, [VP]),
    "skyjo_vec_done_ptr": (VP, [VP]),
    "skyjo_vec_check_error": (C.c_int, [VP, VP]),
    "skyjo_vec_get_counters": (C.c_int, [VP, C.POINTER(Counters), VP]),
    "skyjo_vec_reset_counters": (C.c_int, [VP, VP]),
    "skyjo_vec_get_state": (C.c_int, [VP, I32, C.POINTER(GameState), VP]),
    "skyjo_vec_set_state": (C.c_int, [VP, I32, C.POINTER(GameState), VP]),
    "skyjo_vec_seed_raw": (C.c_int, [VP, I32, U32, VP]),
    "skyjo_vec_rng_set_state": (C.c_int, [VP, I32, VP, I32, VP]),
    "skyjo_vec_rng_get_state": (C.c_int, [VP, I32, VP, C.POINTER(I32), VP]),
    "skyjo_vec_profile": (C.c_int, [VP, C.c_int, C.POINTER(C.c_double), C.POINTER(I64)]),
    "skyjo_vec_snapshot_create": (C.c_int, [VP, C.POINTER(VP), VP]),
    "skyjo_vec_snapshot_restore": (C.c_int, [VP, VP, VP]),
    "skyjo_vec_snapshot_bytes": (C.c_int, [VP, C.POINTER(C.c_size_t)]),
    "skyjo_vec_snapshot_destroy": (C.c_int, [VP]),
    "skyjo_vec_debug_stamps": (C.c_int, [VP, VP]),
    "skyjo_vec_debug_trace": (C.c_int, [VP, VP]),
    "skyjo_vec_set_option": (C.c_int, [VP, C.c_int, I64]),
    "skyjo_vec_get_option": (C.c_int, [VP, C.c_int, C.POINTER(I64)]),
    "skyjo_vec_step_host": (C.c_int, [VP, VP, VP]),
    "skyjo_vec_observe_host": (C.c_int, [VP, VP, VP]),
    "skyjo_vec_reset_host": (C.c_int, [VP, VP, VP]),
    "skyjo_vec_get_rewards_host": (C.c_int, [VP, VP, VP, VP]),
    "skyjo_vec_evaluate_game": (C.c_int, [I32, I32, I32, VP, VP, C.c_double, VP]),
    "skyjo_vec_calc_final_rewards": (C.c_int, [I32, I32, I32, VP, VP, C.c_double, C.c_double, VP]),
    "skyjo_dev_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(VP)]),
    "skyjo_dev_free": (C.c_int, [VP]),
    "skyjo_dev_copy": (C.c_int, [VP, VP, C.c_size_t, C.c_int, VP]),
    "skyjo_dev_sync": (C.c_int, [VP]),
}

_lib = None


def load():
    """Load the HIP engine or raise SkyjoNativeError - never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SkyjoNativeError(
            f"{LIB_PATH} is missing: build it with `python -m skyjo_rl_amd.build` (hipcc, gfx950). "
            "skyjo_rl_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own the
    # GPU ("No HIP GPUs are available" in whichever initialises second), so when torch is installed its copy
    # is loaded first and libskyjo_vec.so binds to that one through the shared SONAME.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise SkyjoNativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.skyjo_vec_abi_version() != ABI_VERSION:
        raise SkyjoNativeError("libskyjo_vec.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().skyjo_vec_last_error()
        raise SkyjoNativeError(f"libskyjo_vec error {rc}: {msg.decode() if msg else ''}")
