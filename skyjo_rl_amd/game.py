"""SkyjoGame - single-game view with the call surface of rlskyjo.game.skyjo.SkyjoGame.

Every method forwards to a batched engine (by default a 1-game ``SkyjoVecEnv`` on the GPU); the
view itself holds no game logic beyond mirroring the reference's argument checks, so loops written
for the reference core API (rlskyjo/game/sample_game.py:5-28) run unchanged:

    game = SkyjoGame(num_players=3)
    game.set_seed(42)
    while not game.is_terminated:
        pid, phase = game.expected_action
        obs, mask = game.collect_observation(pid)
        game.act(pid, policy_ra(obs, mask))

``SkyjoGame(engine=shared_engine, index=i)`` is a view of game ``i`` of a batched engine: its
``set_seed`` / ``reset`` / ``act`` touch that one game (``seed_one``, a one-game reset mask, and a
step in which every other game gets ``ACTION_SKIP``), so single-game code can be pointed at any
game of a running batch.

``SkyjoGame(..., global_rng=True)`` reproduces the reference's RNG contract for single-game use: the
reference deals and reshuffles from numpy's PROCESS-GLOBAL legacy stream (skyjo.py:49,81,94,101,135),
the very stream ``policy_ra(obs, mask)`` without a generator draws from
(random_admissible_policy.py:22-23), so ``np.random.seed(s)`` followed by ``sample_run`` /
``simple_episode`` is reproducible there.  In this mode the game owns no stream: ``np.random``'s state
is handed to the device before every call that can draw (a deal; a draw from an empty draw pile, which
reshuffles) and taken back afterwards, and nothing is dealt ahead (the engine runs with ``no_bank``) -
the loops of rlskyjo/game/sample_game.py:5-28 and rlskyjo/environment/vanilla_env_example.py:6-41
replay bit for bit (tests/golden/global_*.npz).  The default (``global_rng=False``) keeps one private
stream per game, which is what a vector of games needs.

CAVEAT - this is the reference's contract WITHOUT numba's JIT (``NUMBA_DISABLE_JIT=1``, the mode its seeded
test pins, tests/environment/test_skyjo_env_jit.py:1-2, and the mode the golden fixtures were recorded in).
With numba active - the reference's normal mode - ``np.random.*`` inside its ``@njit`` functions draws from
numba's OWN generator, not from numpy's global stream, and the reference's ``set_seed`` docstring says it does
not touch ``np.random.seed()`` of the caller.  ``global_rng=True`` here follows the no-JIT behaviour:
``set_seed(v)`` calls ``np.random.seed(v + 1)`` on the CALLER's global stream (re-seeding it, unlike the jitted
reference), and deals / reshuffles advance it.  Use the default private streams when that is not wanted.

An *engine* is any object with the host-style methods of ``SkyjoVecEnv`` (seed, seed_one,
reset_host, step_host, observe_host, rewards_host, get_state, set_state - and rng_set / rng_get /
no_bank for the global-RNG mode); tests inject an oracle-backed one to check this file without a GPU.
"""
import math
import os
import warnings

import numpy as np

from ._lib import ST_ILLEGAL

_PHASES = ("draw", "place")


class SkyjoGame(object):
    def __init__(self, num_players=3, score_penalty=2, observe_other_player_indirect=False, engine=None,
                 index=0, device=0, seed=None, global_rng=False):
        # rlskyjo/game/skyjo.py:24-26
        assert 0 < num_players <= 12, "Skyjo can be played from 1 up to 8 (recommended) / 12 (theoretical) players"
        self.num_players = num_players
        self.score_penalty = score_penalty
        # rlskyjo/game/skyjo.py:33-37
        self.fill_masked_unk_value = 15
        self.fill_masked_refunded_value = -14
        self.card_dtype = np.int8
        self._name_draw, self._name_place = _PHASES
        self.observe_other_player_indirect = observe_other_player_indirect
        # rlskyjo/game/skyjo.py:43-46
        self.obs_shape = (19 + 12,) if observe_other_player_indirect else (19 + num_players * 12,)
        self.action_mask_shape = (26,)
        if engine is None:
            from .vec_env import SkyjoVecEnv

            engine = SkyjoVecEnv(1, num_players=num_players, score_penalty=float(score_penalty),
                                 observe_other_player_indirect=observe_other_player_indirect, device=device,
                                 auto_reset=False, no_bank=bool(global_rng))
        self._engine, self._i = engine, index
        assert engine.num_players == num_players and tuple(engine.obs_shape) == self.obs_shape
        self._rec = None
        self._global_rng = bool(global_rng)
        if self._global_rng:
            assert getattr(engine, "no_bank", False), "global_rng needs an engine created with no_bank=True (nothing may be dealt ahead)"
            engine.seed(np.zeros(engine.num_envs, dtype=np.uint64))  # (makes the engine usable; this private deal is thrown away)
            self.rng = np.random.default_rng(seed)  # attribute parity (skyjo.py:86): unused, as in the reference
            if seed is not None:
                self.set_seed(seed)
            else:
                self.reset()  # skyjo.py:49: the constructor deals from the global stream where it stands
            return
        # the reference deals in __init__ from the unseeded global RNG (skyjo.py:49): deal from entropy here
        self.set_seed(int.from_bytes(os.urandom(4), "little") if seed is None else seed)

    # ---- engine plumbing (single-game engines only need index 0) ------------------------------
    def _take(self, obs):
        i = self._i
        self._rec = dict(observations=np.array(obs.observations[i]), action_mask=np.array(obs.action_mask[i]),
                         agent=int(obs.agent[i]), phase=int(obs.phase[i]), done=bool(obs.done[i]),
                         status=int(obs.status[i]))
        self._state = None

    def _take_row(self, row):
        """The same from one raw record (include/skyjo_vec.h has the layout): ``SkyjoVecEnv.step_one``'s fast path."""
        D = self.obs_shape[0]
        dp = (D + 3) & ~3
        i8 = row.view(np.int8)
        self._rec = dict(observations=i8[:D].copy(), action_mask=i8[dp:dp + 26].copy(), agent=int(row[dp + 26]),
                         phase=int(row[dp + 27]), done=bool(row[dp + 28]), status=int(row[dp + 29]))
        self._state = None

    def _state_now(self):
        if self._state is None:
            self._state = self._engine.get_state(self._i)
        return self._state

    def _only_me(self, value, fill, dtype):
        a = np.full(self._engine.num_envs, fill, dtype=dtype)
        a[self._i] = value
        return a

    def sync(self):
        """Re-read this game from the engine (after the engine was stepped or its state injected behind the view's back)."""
        self._take(self._engine.observe_host())

    # ---- the caller's stream (global_rng=True) ----------------------------------------------------
    def _stream_to_device(self):
        st = np.random.get_state(legacy=True)
        self._np_tail = (st[3], st[4])  # (has_gauss, cached_gaussian: not ours to touch)
        self._engine.rng_set(self._i, st[1], st[2])

    def _stream_from_device(self):
        key, pos = self._engine.rng_get(self._i)
        np.random.set_state(("MT19937", key, pos) + self._np_tail)

    # ---- reset utils (skyjo.py:52-94) -----------------------------------------------------------
    def reset(self):
        if self._global_rng:
            self._stream_to_device()
        self._take(self._engine.reset_host(self._only_me(1, 0, np.uint8)))
        if self._global_rng:
            self._stream_from_device()
        assert self.expected_action[1] == self._name_draw, "expect to draw after reset"

    def set_seed(self, value):
        """np.random.seed(value + 1) on this game's private legacy stream - on numpy's global one with global_rng=True,
        exactly the reference's call - then deal (skyjo.py:84-88)."""
        if self._global_rng:
            np.random.seed(value + 1)                 # skyjo.py:92-94
            self.rng = np.random.default_rng(value)   # skyjo.py:86
            self.reset()
            return
        self.rng = np.random.default_rng(value)  # kept for attribute parity; unused, as in the reference
        if self._engine.num_envs == 1:
            self._engine.seed(np.array([value], dtype=np.uint64))
        else:
            self._engine.seed_one(self._i, value)  # the other games of a shared engine keep their streams
        self._take(self._engine.observe_host())

    # ---- observation (skyjo.py:148-199) -----------------------------------------------------------
    def collect_observation(self, player_id):
        if self._rec is not None and player_id == self._rec["agent"]:
            return self._rec["observations"].copy(), self._rec["action_mask"].copy()
        o = self._engine.observe_host(self._only_me(player_id, 0, np.int32))
        return np.array(o.observations[self._i]), np.array(o.action_mask[self._i])

    # ---- actions (skyjo.py:308-335) -----------------------------------------------------------------
    def act(self, player_id, action_int):
        exp_player, exp_phase = self._rec["agent"], self._rec["phase"]
        assert exp_player == player_id, (
            f"ILLEGAL ACTION: expected {exp_player} but requested was {player_id}")
        assert 0 <= action_int <= 25, f"action int {action_int} not in range(0,26)"
        if self.is_terminated:
            warnings.warn("Attemp playing terminated game. game has been already terminated by pervios player.")
            return True
        if 24 <= action_int <= 25:
            assert exp_phase == 0, (
                "ILLEGAL ACTION. requested draw action"
                f" {self.render_action_explainer(action_int)}already have a hand card {self.hand_card} ")
        else:
            assert exp_phase == 1, (
                f"ILLEGAL ACTION. requested place action but not having a hand card {self.hand_card} ")
            # skyjo.py:399; a refunded slot (mask 0) is additionally rejected here, where the reference
            # silently corrupts its histogram
            assert self._rec["action_mask"][action_int] == 1, (
                f"illegal action {self.render_action_explainer(action_int)}."
                f"card is already revealed: {self.players_masked[player_id]}")
        # (the other games of a shared engine stay as they are: ACTION_SKIP)
        # the one step that draws random numbers: taking a card from an EMPTY draw pile reshuffles the discard pile first
        # (skyjo.py:361-365) - with the caller's stream when that is the contract
        lend = self._global_rng and action_int == 24 and self._state_now()["n_draw"] == 0 and self._state_now()["hand"] == 15
        if lend:
            self._stream_to_device()
        step_one = getattr(self._engine, "step_one", None)
        if step_one is not None:
            self._take_row(step_one(self._i, int(action_int)))
        else:
            self._take(self._engine.step_host(self._only_me(int(action_int), self._engine.ACTION_SKIP, np.int32)))
        if lend:
            self._stream_from_device()
        assert self._rec["status"] != ST_ILLEGAL
        return self._rec["done"]

    # ---- accessors (skyjo.py:500-504 and attributes) -----------------------------------------------
    @property
    def expected_action(self):
        return [self._rec["agent"], _PHASES[self._rec["phase"]]]

    def get_expected_action(self):
        return list(self.expected_action)

    @property
    def is_terminated(self):
        return bool(self._rec["done"])

    @property
    def hand_card(self):
        return self._state_now()["hand"]

    @property
    def players_cards(self):
        return self._state_now()["cards"]

    @property
    def players_masked(self):
        return self._state_now()["masked"]

    @property
    def drawpile(self):
        return list(self._state_now()["draw"])

    @property
    def discard_pile(self):
        return list(self._state_now()["disc"])

    @property
    def game_metrics(self):
        """``final_score`` is typed as the reference types it WITHOUT numba (``numba.config.DISABLE_JIT`` - the mode its seeded
        test pins and oracle/gen_golden.py records): np.float64 where a column counted, Python float where not.  With real
        numba ``_evaluate_game`` is jitted and returns plain Python floats throughout; only the repr in ``render_table()``'s
        "Results: {...}" line differs (tests/golden/render.npz holds the non-JIT text)."""
        s = self._state_now()
        final = False
        if s["is_terminated"]:
            # skyjo.py:477-498 starts from a list of Python floats and adds numpy integers to it: an entry becomes a
            # numpy float64 as soon as one column of that player counts, and stays a Python float (0.0) otherwise.
            # The difference shows in render_table()'s "Results: {...}" (repr of the values), so it is kept.
            counted = [any(len(set(row[3 * c:3 * c + 3])) > 1 for c in range(4)) for row in s["cards"].tolist()]
            final = [np.float64(x) if k else float(x) for x, k in zip(s["final_score"], counted)]
        return {"num_refunded": [int(x) for x in s["num_refunded"]], "num_placed": [int(x) for x in s["num_placed"]],
                "final_score": final}

    def get_game_metrics(self):
        return self.game_metrics

    @staticmethod
    def _evaluate_game(players_cards, player_won_id, score_penalty: float = 2.0, device=0):
        """``SkyjoGame._evaluate_game`` (skyjo.py:477-498) for caller-supplied hands - the reference's notebook calls it
        directly (notebooks/trainpettingzoo.ipynb:52745-52758).  ``players_cards`` int8 [num_players, 12]; returns the list of
        scores.  Computed by the engine's own scoring arithmetic on the device (``skyjo_vec_evaluate_game``): there is no host
        implementation behind this."""
        import ctypes as C

        from . import _lib

        cards = np.ascontiguousarray(players_cards, dtype=np.int8)
        assert cards.ndim == 2 and cards.shape[1] == 12, "players_cards must be [num_players, 12]"
        won = np.asarray([player_won_id], dtype=np.int32)
        out = np.zeros(cards.shape[0], dtype=np.float64)
        _lib.check(_lib.load().skyjo_vec_evaluate_game(int(device), 1, int(cards.shape[0]), cards.ctypes.data_as(C.c_void_p),
                                                   won.ctypes.data_as(C.c_void_p), float(score_penalty), out.ctypes.data_as(C.c_void_p)))
        return [float(x) for x in out]

    # ---- render utils: same text as skyjo.py:508-602 ------------------------------------------------
    def render_table(self):
        bar = "=" * 7
        out = f"{bar} render board: {'=' * 5} \n" + self._render_game_stats()
        show_hidden = False
        if self.is_terminated:
            results = dict(zip(range(self.num_players), self.game_metrics["final_score"]))
            out += f"{bar} GAME DONE {'=' * 8} \nResults: {results} \n"
            show_hidden = True
        return out + "".join(self.render_player(p, show_hidden) for p in range(self.num_players))

    def _render_game_stats(self):
        hand = self.hand_card if -2 <= self.hand_card <= 12 else "empty"
        pile = self.discard_pile
        top = pile[-1] if pile else "empty"
        who, what = self.expected_action
        return (f"{'=' * 7} stats {'=' * 12} \n"
                f"next turn: {what} by Player {who} \n"
                f"holding card player {who}: {hand} \n"
                f"discard pile top: {top} \n")

    def _render_player_cards(self, player_id, render_cards_open):
        cards, masked = self.players_cards[player_id], self.players_masked[player_id]
        cells = []
        for c, m in zip(cards, masked):
            if m == 0:
                cells.append("d")
            elif m == 2:
                cells.append(f"u{c}" if render_cards_open else "u")
            else:
                cells.append(str(c))
        grid = np.array(cells, dtype=np.str_).reshape(4, -1).T  # 3 rows x 4 columns
        return np.array2string(grid, separator="\t ", formatter={"str_kind": lambda x: str(x)})

    def render_player(self, player_id, render_cards_open=False):
        return f"{'=' * 7} Player {player_id} {'=' * 10} \n" + self._render_player_cards(player_id, render_cards_open) + "\n"

    @classmethod
    def render_action_explainer(cls, action_int):
        assert action_int in range(0, 26), "action not valid action int {action_int}"
        if action_int == 24:
            return "draw from drawpile"
        if action_int == 25:
            return "draw from discard pile"
        if action_int < 12:
            place_id, text = action_int, f"place card ({action_int}) - "
        else:
            place_id, text = action_int - 12, f"handcard discard & reveal card ({action_int}) - "
        # the reference reports row = place_id % 4 (skyjo.py:583-587); kept for string parity
        return text + f"col:{math.floor(place_id / 3)} row:{place_id % 4}"

    @classmethod
    def render_actions(cls):
        ids = np.arange(12).reshape(4, -1).T
        grid = np.array([[f"{a}/{a + 12}" for a in row] for row in ids], dtype=np.str_)
        text = np.array2string(grid, separator="\t ", formatter={"str_kind": lambda x: str(x)})
        return (f"action ids 0-25: \n(put handcard here / reveal this card) \n {text} \n"
                f"24: draw from drawpile \n 25: draw from discard pile")
