"""ctypes binding of oracle/skyjo_oracle.c.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under skyjo_rl_amd/ does (the product fails loudly without its HIP library).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libskyjo_oracle.so")
# SKYJO_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (`make -C oracle asan`); the interpreter must then run with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) - `make -C oracle check-asan` does both
_SANITIZE = os.environ.get("SKYJO_ORACLE_SANITIZE") == "1"
if _SANITIZE:
    _SO = os.path.join(_HERE, "_build", "libskyjo_oracle_asan.so")

ST_OK, ST_ILLEGAL, ST_NOOP_DONE, ST_RESET = 0, 1, 2, 3
RNG_MT19937, RNG_PHILOX = 0, 1
ACTION_SKIP = -1000
MAXP = 12


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("skyjo_oracle.c", "skyjo_oracle.h")]
    if force or not os.path.exists(_SO) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        if all(os.path.exists(s) for s in src):
            subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _SANITIZE else []), stdout=subprocess.DEVNULL)
    return _SO


class _Rng(C.Structure):
    _fields_ = [("mode", C.c_int), ("mt", C.c_uint32 * 624), ("idx", C.c_int), ("px_key", C.c_uint64),
                ("px_ctr", C.c_uint32 * 4), ("px_buf", C.c_uint32 * 4), ("px_pos", C.c_int)]


class Game(C.Structure):
    _fields_ = [
        ("num_players", C.c_int), ("score_penalty", C.c_double), ("indirect", C.c_int),
        ("players_cards", (C.c_int8 * 12) * MAXP), ("players_masked", (C.c_int8 * 12) * MAXP),
        ("drawpile", C.c_int8 * 158), ("n_draw", C.c_int), ("discard_pile", C.c_int8 * 158), ("n_disc", C.c_int),
        ("hand_card", C.c_int), ("exp_player", C.c_int), ("exp_phase", C.c_int), ("is_terminated", C.c_int),
        ("num_refunded", C.c_int * MAXP), ("num_placed", C.c_int * MAXP), ("final_score", C.c_double * MAXP),
        ("episode", C.c_uint32), ("reshuffles", C.c_uint32), ("reshuffles_total", C.c_uint64), ("rng", _Rng)]


class Vec(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int), ("num_players", C.c_int), ("indirect", C.c_int), ("rng_mode", C.c_int),
        ("auto_reset", C.c_int), ("score_penalty", C.c_double), ("mean_reward", C.c_double),
        ("reward_refunded", C.c_double), ("illegal_reward", C.c_double), ("game_id0", C.c_uint64),
        ("games", C.POINTER(Game)), ("done", C.POINTER(C.c_uint8)), ("status", C.POINTER(C.c_uint8)),
        ("rewards", C.POINTER(C.c_double)), ("steps", C.c_uint64), ("episodes", C.c_uint64),
        ("illegal", C.c_uint64), ("resets", C.c_uint64), ("sum_len", C.c_uint64), ("iter", C.c_uint64),
        ("ep_len", C.POINTER(C.c_uint32)), ("acc_score", C.POINTER(C.c_double)), ("acc_refunded", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        assert L.sko_sizeof() == C.sizeof(Game), (L.sko_sizeof(), C.sizeof(Game))
        L.sko_sizeof.restype = C.c_size_t
        L.sko_rng_next.restype = C.c_uint32
        L.sko_rng_interval.restype = C.c_uint32
        L.sko_vec_create.restype = C.POINTER(Vec)
        L.sko_vec_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.c_int, C.c_uint64]
        L.sko_init.argtypes = [C.POINTER(Game), C.c_int, C.c_double, C.c_int, C.c_int]
        L.sko_set_seed.argtypes = [C.POINTER(Game), C.c_uint64]
        L.sko_vec_seed.argtypes = [C.POINTER(Vec), C.c_void_p, C.c_uint64]
        L.sko_vec_seed_one.argtypes = [C.POINTER(Vec), C.c_int, C.c_uint64]
        L.sko_vec_reset.argtypes = [C.POINTER(Vec), C.c_void_p]
        L.sko_vec_step.argtypes = [C.POINTER(Vec), C.c_void_p, C.c_int]
        L.sko_vec_observe.argtypes = [C.POINTER(Vec), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sko_vec_rollout.argtypes = [C.POINTER(Vec), C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.sko_vec_rollout_rec.argtypes = [C.POINTER(Vec), C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int]
        L.sko_policy_action.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.sko_final_rewards.argtypes = [C.POINTER(Game), C.c_double, C.c_double, C.c_void_p]
        L.sko_evaluate_game.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.sko_set_state.argtypes = [C.POINTER(Game), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int]
        L.sko_rng_seed_legacy.argtypes = [C.POINTER(_Rng), C.c_uint32]
        L.sko_rng_next.argtypes = [C.POINTER(_Rng)]
        L.sko_shuffle_i32.argtypes = [C.POINTER(_Rng), C.c_void_p, C.c_int]
        L.sko_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleGame:
    """One game with the call surface of rlskyjo's SkyjoGame (skyjo.py:52,84,148,308,500,503)."""

    def __init__(self, num_players=3, score_penalty=2.0, observe_other_player_indirect=False,
                 rng_mode=RNG_MT19937):
        self.L = lib()
        self.g = Game()
        self.L.sko_init(C.byref(self.g), num_players, float(score_penalty), int(observe_other_player_indirect),
                        rng_mode)
        self.num_players = num_players
        self.obs_dim = self.L.sko_obs_dim(C.byref(self.g))

    def set_seed(self, value):
        self.L.sko_set_seed(C.byref(self.g), int(value))

    def seed_legacy_raw(self, seed):
        self.L.sko_rng_seed_legacy(C.byref(self.g.rng), int(seed) & 0xFFFFFFFF)

    def reset(self):
        self.L.sko_reset(C.byref(self.g))

    def collect_observation(self, player):
        obs = np.zeros(self.obs_dim, dtype=np.int8)
        mask = np.zeros(26, dtype=np.int8)
        self.L.sko_observe(C.byref(self.g), int(player), _p(obs), _p(mask))
        return obs, mask

    def act(self, player, action):
        return self.L.sko_act(C.byref(self.g), int(player), int(action))

    @property
    def expected_action(self):
        return [self.g.exp_player, self.g.exp_phase]

    @property
    def is_terminated(self):
        return bool(self.g.is_terminated)

    def set_state(self, cards, masked, draw, disc, hand, player, phase):
        cards = np.ascontiguousarray(cards, dtype=np.int8)
        masked = np.ascontiguousarray(masked, dtype=np.int8)
        draw = np.ascontiguousarray(draw, dtype=np.int8)
        disc = np.ascontiguousarray(disc, dtype=np.int8)
        self.L.sko_set_state(C.byref(self.g), _p(cards), _p(masked), _p(draw), len(draw), _p(disc), len(disc),
                             int(hand), int(player), int(phase))

    def snapshot(self):
        g, N = self.g, self.num_players
        return dict(
            cards=np.array([list(g.players_cards[p]) for p in range(N)], dtype=np.int8),
            masked=np.array([list(g.players_masked[p]) for p in range(N)], dtype=np.int8),
            draw=np.array(list(g.drawpile[: g.n_draw]), dtype=np.int8), n_draw=g.n_draw,
            disc=np.array(list(g.discard_pile[: g.n_disc]), dtype=np.int8), n_disc=g.n_disc,
            hand=g.hand_card, player=g.exp_player, phase=g.exp_phase)

    def final_score(self):
        return np.array(list(self.g.final_score[: self.num_players]), dtype=np.float64)

    def metrics(self):
        N = self.num_players
        return (np.array(list(self.g.num_refunded[:N]), dtype=np.int32),
                np.array(list(self.g.num_placed[:N]), dtype=np.int32))

    def final_rewards(self, mean_reward, reward_refunded):
        out = np.zeros(self.num_players, dtype=np.float64)
        self.L.sko_final_rewards(C.byref(self.g), float(mean_reward), float(reward_refunded), _p(out))
        return out


class OracleVec:
    """Batched oracle with the same call surface as skyjo_rl_amd.vec_env.SkyjoVecEnv (host numpy)."""

    def __init__(self, num_envs, num_players=3, score_penalty=2.0, observe_other_player_indirect=True,
                 mean_reward=1.0, reward_refunded=0.001, rng_mode=RNG_MT19937, auto_reset=True, game_id0=0):
        self.L = lib()
        self.v = self.L.sko_vec_create(num_envs, num_players, float(score_penalty),
                                       int(observe_other_player_indirect), float(mean_reward),
                                       float(reward_refunded), rng_mode, int(auto_reset), int(game_id0))
        if not self.v:
            raise ValueError("bad oracle vec config")
        self.num_envs, self.num_players = num_envs, num_players
        self.obs_dim = 31 if observe_other_player_indirect else 19 + 12 * num_players

    def __del__(self):
        if getattr(self, "v", None):
            self.L.sko_vec_destroy(self.v)
            self.v = None

    def seed(self, seeds=None, base=0):
        s = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        self.L.sko_vec_seed(self.v, _p(s), int(base))

    def seed_one(self, game, value):
        self.L.sko_vec_seed_one(self.v, int(game), int(value))

    def set_state(self, game, cards, masked, draw, disc, hand=15, player=0, phase=0):
        """Fixture injection into one game of the batch (sko_set_state)."""
        N = self.num_players
        c = np.ascontiguousarray(np.asarray(cards, dtype=np.int8).reshape(N, 12))
        m = np.ascontiguousarray(np.asarray(masked, dtype=np.int8).reshape(N, 12))
        d = np.ascontiguousarray(draw, dtype=np.int8).ravel()
        s = np.ascontiguousarray(disc, dtype=np.int8).ravel()
        self.L.sko_set_state(C.byref(self.v.contents.games[game]), _p(c), _p(m), _p(d), len(d), _p(s), len(s), int(hand),
                             int(player), int(phase))
        self.v.contents.done[game] = 0

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.sko_vec_reset(self.v, _p(m))

    def step(self, actions, threads=1):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        self.L.sko_vec_step(self.v, _p(a), threads)

    def rollout(self, iters, policy_seed, threads=1, record_actions=False, record_obs=False):
        """`iters` lockstep iterations with the restated on-device policy.  record_obs: also what the engine's record of EVERY
        iteration holds - returns (actions, obs [iters, B, D], mask [iters, B, 26], meta [iters, B, 4] = agent, phase, done,
        status, episode_steps [iters, B])."""
        acts = np.zeros((iters, self.num_envs), dtype=np.int32) if (record_actions or record_obs) else None
        if not record_obs:
            self.L.sko_vec_rollout(self.v, iters, int(policy_seed), _p(acts), threads)
            return acts
        B = self.num_envs
        obs = np.zeros((iters, B, self.obs_dim), dtype=np.int8)
        mask = np.zeros((iters, B, 26), dtype=np.int8)
        meta = np.zeros((iters, B, 4), dtype=np.uint8)
        eplen = np.zeros((iters, B), dtype=np.uint16)
        self.L.sko_vec_rollout_rec(self.v, iters, int(policy_seed), _p(acts), _p(obs), _p(mask), _p(meta), _p(eplen), threads)
        return acts, obs, mask, meta, eplen

    def observe(self, players=None):
        B = self.num_envs
        obs = np.zeros((B, self.obs_dim), dtype=np.int8)
        mask = np.zeros((B, 26), dtype=np.int8)
        agent = np.zeros(B, dtype=np.uint8)
        phase = np.zeros(B, dtype=np.uint8)
        pl = None if players is None else np.ascontiguousarray(players, dtype=np.int32)
        self.L.sko_vec_observe(self.v, _p(pl), _p(obs), _p(mask), _p(agent), _p(phase))
        return obs, mask, agent, phase

    @property
    def dones(self):
        return np.ctypeslib.as_array(self.v.contents.done, shape=(self.num_envs,)).copy()

    @property
    def status(self):
        return np.ctypeslib.as_array(self.v.contents.status, shape=(self.num_envs,)).copy()

    @property
    def rewards(self):
        return np.ctypeslib.as_array(self.v.contents.rewards, shape=(self.num_envs, self.num_players)).copy()

    def game(self, i):
        return self.v.contents.games[i]

    def seat_sums(self):
        """Sums over every finished episode since creation: (final score per seat, num_refunded per seat)."""
        shp = (self.num_envs, self.num_players)
        return (np.ctypeslib.as_array(self.v.contents.acc_score, shape=shp).sum(axis=0),
                np.ctypeslib.as_array(self.v.contents.acc_refunded, shape=shp).sum(axis=0))

    def counters(self):
        c = self.v.contents
        return dict(steps=c.steps, episodes=c.episodes, illegal=c.illegal, resets=c.resets, sum_len=c.sum_len,
                    iter=c.iter)


def policy_action(policy_seed, game_id, it, mask):
    m = np.ascontiguousarray(mask, dtype=np.int8)
    return lib().sko_policy_action(int(policy_seed), int(game_id), int(it), _p(m))
