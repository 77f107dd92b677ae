#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (rlskyjo) in this container.

Test infrastructure, not product code.  Run from the repo root:

    python oracle/gen_golden.py

It imports the reference from /root/reference (read-only, never copied) through the
identity-JIT ``numba`` stand-in in oracle/_shims (== numba.config.DISABLE_JIT = True, the mode
pinned by the reference's seeded test, tests/environment/test_skyjo_env_jit.py:1-2) and records
inputs -> outputs of the hot path:

  rng_kat.npz      numpy legacy RandomState KATs used by rlskyjo/game/skyjo.py:81,94,101,135
  traj_*.npz       full seeded trajectories of SkyjoGame (skyjo.py:52-504), actions from
                   policy_ra (random_admissible_policy.py:6-28) and STORED, 3 episodes
                   back-to-back (the process-global RNG carries over)
  dense_*.npz      games from injected decks / stalling policies that force the rare branches
                   (column collapse, multi-collapse, empty discard, mid-game reshuffle)
  scenarios.npz    hand-built states: every parity trap of SURVEY.md section 8.1
  env_*.npz        SimpleSkyjoEnv (skyjo_env.py:29-334) driven through pettingzoo STAND-INS
                   ("wrapper semantics unpinned"): obs / rewards / dones per agent_iter turn,
                   including the TerminateIllegalWrapper flow
  render.npz       the text the reference's render utils print (skyjo.py:508-602) for fresh,
                   mid-game, collapsed-column, empty-discard and terminated states
  global_*.npz     the reference's OWN caller loops on the process-global numpy stream: np.random.seed(s),
                   then sample_run (rlskyjo/game/sample_game.py:5-28) / simple_episode
                   (rlskyjo/environment/vanilla_env_example.py:6-41) with policy_ra(obs, mask) and NO generator
                   (random_admissible_policy.py:22-23): deals, mid-game reshuffles and the policy's draws all
                   come out of the one stream, in the order the loops make them; the stream's state is
                   recorded at every episode end

  policy_stats.npz the reference's random admissible policy in numbers: for N in {2, 3, 4}, 6 000 games of SkyjoGame played
                   by policy_ra (random_admissible_policy.py:26-28: choice(arange(26), p=mask/sum(mask))) - per game the
                   episode length, per seat the final score and num_refunded; per place / draw turn the rank of the
                   chosen action among the legal ones, counted by the number of legal actions.  The on-device policy
                   (a different random stream by construction) is held to these distributions (tests/test_gpu_policy_stats.py)

The fixtures are data (inputs and expected outputs) - no reference source text is stored.
"""
import itertools
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "_shims"), "/root/reference"]

from rlskyjo.game.skyjo import SkyjoGame  # noqa: E402
from rlskyjo.models.random_admissible_policy import policy_ra  # noqa: E402
from rlskyjo.environment import skyjo_env  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

REWARD_CFGS = [(1.0, 0.001), (0.0, 0.0), (-1.0, 0.01), (1.0, 0.0)]  # (mean_reward, reward_refunded)


def calc_rewards(final_score, num_refunded, mean_reward, reward_refunded):
    """SimpleSkyjoEnv._calc_final_rewards (skyjo_env.py:293-312) called unbound."""
    ns = types.SimpleNamespace(mean_reward=mean_reward, reward_refunded=reward_refunded)
    return np.asarray(
        skyjo_env.SimpleSkyjoEnv._calc_final_rewards(ns, final_score=final_score, num_refunded=num_refunded),
        dtype=np.float64,
    )


def pad150(lst):
    a = np.full(150, 99, dtype=np.int8)
    a[: len(lst)] = np.asarray(lst, dtype=np.int8) if len(lst) else []
    return a


def force_turn(g, player, phase):
    """Put the reference's itertools.cycle at (player, phase) (skyjo.py:114-125)."""
    g.actions = itertools.cycle(
        ([p, ph] for p in range(g.num_players) for ph in [g._name_draw, g._name_place])
    )
    for _ in range(1 + 2 * player + (1 if phase == "place" else 0)):
        g._internal_next_action()


def snapshot(g):
    return dict(
        cards=np.array(g.players_cards, dtype=np.int8).copy(),
        masked=np.array(g.players_masked, dtype=np.int8).copy(),
        draw=pad150(list(g.drawpile)),
        n_draw=len(g.drawpile),
        disc=pad150(list(g.discard_pile)),
        n_disc=len(g.discard_pile),
        hand=int(g.hand_card),
        player=int(g.expected_action[0]),
        phase=0 if g.expected_action[1] == "draw" else 1,
    )


class Recorder:
    """Per-step record of one SkyjoGame driven through its public API."""

    def __init__(self, g):
        self.g = g
        self.steps = {k: [] for k in (
            "player", "phase", "action", "hand", "n_draw", "n_disc", "obs", "mask",
            "obs_other", "mask_other", "game_over")}
        self.ep_start = [0]
        self.deals = {k: [] for k in ("cards", "masked", "draw", "n_draw", "disc", "n_disc", "hand", "player", "phase")}
        self.ends = {k: [] for k in ("cards", "masked", "draw", "n_draw", "disc", "n_disc", "hand", "player", "phase")}
        self.final_score, self.num_refunded, self.num_placed = [], [], []
        self.rewards = []

    def begin_episode(self):
        for k, v in snapshot(self.g).items():
            self.deals[k].append(v)

    def step(self, choose):
        g = self.g
        pid, phase = g.expected_action
        obs, mask = g.collect_observation(pid)
        other = (pid + 1) % g.num_players
        obs_o, mask_o = g.collect_observation(other)
        a = int(choose(obs, mask))
        s = self.steps
        s["player"].append(pid)
        s["phase"].append(0 if phase == "draw" else 1)
        s["action"].append(a)
        s["hand"].append(int(g.hand_card))
        s["n_draw"].append(len(g.drawpile))
        s["n_disc"].append(len(g.discard_pile))
        s["obs"].append(np.asarray(obs, dtype=np.int8))
        s["mask"].append(np.asarray(mask, dtype=np.int8))
        s["obs_other"].append(np.asarray(obs_o, dtype=np.int8))
        s["mask_other"].append(np.asarray(mask_o, dtype=np.int8))
        over = bool(g.act(pid, a))
        s["game_over"].append(int(over))
        return over

    def end_episode(self):
        g = self.g
        self.ep_start.append(len(self.steps["action"]))
        for k, v in snapshot(g).items():
            self.ends[k].append(v)
        m = g.get_game_metrics()
        self.final_score.append(np.asarray(m["final_score"], dtype=np.float64))
        self.num_refunded.append(np.asarray(m["num_refunded"], dtype=np.int32))
        self.num_placed.append(np.asarray(m["num_placed"], dtype=np.int32))
        self.rewards.append(np.stack([
            calc_rewards(m["final_score"], m["num_refunded"], mr, rr) for mr, rr in REWARD_CFGS]))
        # trap 18: acting on a terminated game is a no-op returning True (skyjo.py:316-321)
        before = snapshot(g)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert g.act(g.expected_action[0], 24) is True
        after = snapshot(g)
        for k in before:
            assert np.array_equal(before[k], after[k])

    def arrays(self, **meta):
        out = dict(meta)
        out["ep_start"] = np.asarray(self.ep_start, dtype=np.int32)
        for k, v in self.steps.items():
            out[k] = np.asarray(v, dtype=np.int8 if k in ("obs", "mask", "obs_other", "mask_other") else np.int32)
        for k, v in self.deals.items():
            out["deal_" + k] = np.asarray(v)
        for k, v in self.ends.items():
            out["end_" + k] = np.asarray(v)
        out["final_score"] = np.asarray(self.final_score, dtype=np.float64)
        out["num_refunded"] = np.asarray(self.num_refunded, dtype=np.int32)
        out["num_placed"] = np.asarray(self.num_placed, dtype=np.int32)
        out["rewards"] = np.asarray(self.rewards, dtype=np.float64)  # [E, len(REWARD_CFGS), N]
        out["reward_cfgs"] = np.asarray(REWARD_CFGS, dtype=np.float64)
        return out


# --------------------------------------------------------------------------------------
def gen_rng_kat():
    """G6: the legacy stream the reference consumes."""
    seeds = [0, 1, 43, 123457, 2 ** 32 - 1]
    raw, shuf150, shuf114, perm12 = [], [], [], []
    for s in seeds:
        rs = np.random.RandomState(s)
        raw.append(rs._bit_generator.random_raw(700).astype(np.uint32))
        np.random.seed(s)
        a = np.arange(150)
        np.random.shuffle(a)
        shuf150.append(a.copy())
        b = np.arange(114)
        np.random.shuffle(b)
        shuf114.append(b.copy())
        perm12.append(np.stack([np.random.choice(12, 2, replace=False) for _ in range(3)]))
    np.savez_compressed(
        os.path.join(OUT, "rng_kat.npz"), seeds=np.asarray(seeds, dtype=np.uint64), raw=np.asarray(raw),
        shuf150=np.asarray(shuf150, dtype=np.int32), shuf114=np.asarray(shuf114, dtype=np.int32),
        perm12=np.asarray(perm12, dtype=np.int32))


def gen_traj(N, seed, indirect, penalty=2.0, episodes=3, tag=None):
    """G1+G2: set_seed(seed) deal (skyjo.py:84-88), then `episodes` games back to back."""
    g = SkyjoGame(N, penalty, indirect)
    g.set_seed(seed)
    rng = np.random.default_rng(seed)
    rec = Recorder(g)
    for e in range(episodes):
        if e > 0:
            g.reset()
        rec.begin_episode()
        while not g.is_terminated:
            rec.step(lambda o, m: policy_ra(o, m, rng=rng))
        rec.end_episode()
    name = tag or "traj_N%d_s%d_%s" % (N, seed, "ind" if indirect else "dir")
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        **rec.arrays(kind="traj", num_players=N, seed=seed, indirect=int(indirect), score_penalty=float(penalty)))
    return len(rec.steps["action"])


def inject_deck(g, deck, deck_rng):
    """Replace the dealt cards by `deck` (shuffled with a private Generator, not the global RNG)."""
    N = g.num_players
    deck = np.asarray(deck, dtype=np.int8).copy()
    deck_rng.shuffle(deck)
    g.players_cards = deck[: 12 * N].reshape(N, -1).copy()
    rest = [np.int8(x) for x in deck[12 * N:]]
    g.discard_pile = [rest.pop()]
    g.drawpile = rest
    g._reset_start_player()  # first argmax of revealed sums (skyjo.py:105-125)


def gen_dense(N, seed, indirect, deck_values, stall, episodes=2, penalty=2.0, draw_cut=None, name=None):
    """Games whose decks / policies force the rare branches.

    deck_values: few distinct values -> many column collapses (skyjo.py:431-469).
    stall: probability of preferring "replace an open card / draw" -> long games ->
           empty drawpile -> _reshuffle_discard_pile (skyjo.py:361-365).
    draw_cut: keep only this many cards in the drawpile (rest moved under the discard pile).
    The reshuffle consumes the process-global legacy RNG seeded by set_seed(seed).
    """
    g = SkyjoGame(N, penalty, indirect)
    g.set_seed(seed)
    prng = np.random.default_rng(seed + 7)
    deck_rng = np.random.default_rng(seed + 11)
    rec = Recorder(g)

    def choose(obs, mask):
        legal = np.flatnonzero(mask)
        if prng.random() < stall:
            if indirect:  # own cards follow the 19 global features (skyjo.py:180-188)
                pref = [a for a in legal if a == 24 or (a < 12 and obs[19 + a] != 15)]
            else:
                pid = g.expected_action[0]
                pref = [a for a in legal if a == 24 or (a < 12 and obs[19 + 12 * pid + a] != 15)]
            if pref:
                return int(prng.choice(pref))
        return int(prng.choice(legal))

    for e in range(episodes):
        if e > 0:
            g.reset()
        deck = np.resize(np.asarray(deck_values, dtype=np.int8), 150)
        inject_deck(g, deck, deck_rng)
        if draw_cut is not None and len(g.drawpile) > draw_cut:
            moved = g.drawpile[draw_cut:]
            g.drawpile = g.drawpile[:draw_cut]
            g.discard_pile = moved + g.discard_pile
        rec.begin_episode()
        guard = 0
        while not g.is_terminated and guard < 6000:
            rec.step(choose)
            guard += 1
        assert g.is_terminated
        rec.end_episode()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        **rec.arrays(kind="dense", num_players=N, seed=seed, indirect=int(indirect), score_penalty=float(penalty)))
    collapses = int(np.sum(rec.num_refunded))
    return len(rec.steps["action"]), collapses


def run_scenario(sc):
    """One injected state + action list -> per-step outputs (obs BEFORE the action, state AFTER)."""
    N = sc["N"]
    g = SkyjoGame(N, sc.get("penalty", 2.0), sc.get("indirect", True))
    g.players_cards = np.asarray(sc["cards"], dtype=np.int8).reshape(N, 12).copy()
    g.players_masked = np.asarray(sc["masked"], dtype=np.int8).reshape(N, 12).copy()
    g.drawpile = [np.int8(x) for x in sc["draw"]]
    g.discard_pile = [np.int8(x) for x in sc["disc"]]
    g.hand_card = sc.get("hand", 15) if sc.get("hand", 15) == 15 else np.int8(sc["hand"])
    force_turn(g, sc["player"], sc["phase"])
    if "np_seed" in sc:
        np.random.seed(sc["np_seed"])
    init = snapshot(g)
    per = {k: [] for k in ("obs", "mask", "over", "cards", "masked", "draw", "n_draw", "disc", "n_disc",
                           "hand", "player", "phase")}
    for a in sc["actions"]:
        pid = g.expected_action[0]
        obs, mask = g.collect_observation(pid)
        per["obs"].append(np.asarray(obs, dtype=np.int8))
        per["mask"].append(np.asarray(mask, dtype=np.int8))
        over = bool(g.act(pid, a))
        per["over"].append(int(over))
        for k, v in snapshot(g).items():
            per[k].append(v)
    obs_f, mask_f = g.collect_observation(g.expected_action[0])
    m = g.get_game_metrics()
    fs = np.asarray(m["final_score"], dtype=np.float64) if g.is_terminated else np.zeros(N)
    return init, per, obs_f, mask_f, fs, np.asarray(m["num_refunded"]), np.asarray(m["num_placed"]), g.is_terminated


def gen_scenarios():
    H, O, C = 2, 1, 0  # hidden / open / collapsed (skyjo.py:99,394,454)
    scs = []
    base_cards3 = [[8, 8, 3, 1, 2, 3, 4, 5, 6, 7, 9, 10], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11],
                   [5, 5, 5, -1, -2, 0, 1, 2, 3, 4, 12, 12]]
    m_open2 = [[O, O, H, H, H, H, H, H, H, H, H, H]] * 3
    # a: place an 8 onto the hidden third slot of a column with two open 8s -> collapse, [old,0,0,0] to discard
    scs.append(dict(name="collapse_place_hidden", N=3, cards=base_cards3, masked=m_open2, draw=[4, 5, 6, 8],
                    disc=[7], player=0, phase="draw", actions=[24, 2, 24, 12 + 5, 25, 0]))
    # b: reveal (12+slot) a hidden 8 completing the column
    c = [row[:] for row in base_cards3]
    c[0][2] = 8
    scs.append(dict(name="collapse_reveal", N=3, cards=c, masked=m_open2, draw=[4, 5, 6, 1], disc=[7],
                    player=0, phase="draw", actions=[24, 12 + 2, 24, 3]))
    # c: collapse on the last hidden slot, everybody gets one more turn, finisher's next draw ends the game
    c = [[8, 8, 8, 1, 2, 3, 4, 5, 6, 7, 9, 10], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
    m = [[O, O, H, O, O, O, O, O, O, O, O, O], [O, O, H, H, H, H, H, H, H, H, H, H]]
    scs.append(dict(name="finish_by_collapse", N=2, cards=c, masked=m, draw=[4, 5, 6, 1, 2], disc=[7],
                    player=0, phase="draw", actions=[24, 12 + 2, 24, 12 + 2, 24]))
    # d: discard goes empty after action 25 -> obs[17] == -3 during the place phase (skyjo.py:254)
    scs.append(dict(name="empty_discard", N=2, cards=c, masked=[[O, O, H, H, H, H, H, H, H, H, H, H]] * 2,
                    draw=[4, 5, 6], disc=[7], player=1, phase="draw", actions=[25, 4, 25, 12 + 7]))
    # e: forced mid-game reshuffle of the WHOLE discard pile incl. its top (skyjo.py:361-366)
    scs.append(dict(name="reshuffle_forced", N=2, cards=c, masked=[[O, O, H, H, H, H, H, H, H, H, H, H]] * 2,
                    draw=[], disc=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, -1, -2, 3, 3], player=0, phase="draw",
                    np_seed=777, actions=[24, 12 + 3, 24, 5, 24, 12 + 4]))
    scs.append(dict(name="reshuffle_small", N=1, cards=[c[1]], masked=[[O, O, H, H, H, H, H, H, H, H, H, H]],
                    draw=[], disc=[4, -2], player=0, phase="draw", np_seed=5, actions=[24, 0, 24, 1, 25, 12 + 2]))
    # f: scoring KATs.  notebook datum (notebooks/trainpettingzoo.ipynb:52745-52758): penalty on finisher 0
    nb = [[-1, 9, 7, -2, 4, 2, 0, 7, 4, 0, 3, 5], [0, 7, 1, 10, 7, 2, 0, 6, 1, -1, -1, 9],
          [-1, 6, 5, -2, 4, 2, 1, 4, -2, 3, -2, 3]]
    allopen = [[O] * 12] * 3
    scs.append(dict(name="score_notebook", N=3, cards=nb, masked=allopen, draw=[1, 2, 3], disc=[4], player=0,
                    phase="draw", actions=[24]))
    # finisher is winner -> no penalty; tie for minimum -> no penalty (skyjo.py:496-497)
    scs.append(dict(name="score_finisher_wins", N=3, cards=nb, masked=allopen, draw=[1, 2, 3], disc=[4], player=2,
                    phase="draw", actions=[24]))
    tie = [[1, 2, 3, 4, 5, 6, 0, 0, 1, 0, 0, 1], [1, 2, 3, 4, 5, 6, 0, 0, 1, 0, 0, 1]]
    scs.append(dict(name="score_tie_min", N=2, cards=tie, masked=[[O] * 12] * 2, draw=[1, 2], disc=[4], player=1,
                    phase="draw", actions=[25]))
    # negative finisher score that is not the minimum gets multiplied too; non-dyadic penalty
    neg = [[-2, -2, -1, -2, -1, -1, 0, 0, 1, -2, 0, -1], [-2, -2, -1, -2, -2, -1, -2, -1, -1, -2, -1, -1],
           [12, 12, 11, 12, 11, 10, 9, 9, 8, 5, 5, 6]]
    scs.append(dict(name="score_negative_penalty", N=3, cards=neg, masked=allopen, draw=[1], disc=[4], player=0,
                    phase="draw", penalty=1.3, actions=[24]))
    # hidden equal triple is skipped at scoring; -14 (collapsed) column skipped; other players still hidden
    hid = [[7, 7, 7, 1, 2, 3, -14, -14, -14, 4, 4, 5], [9, 9, 9, 1, 1, 1, 2, 3, 4, 12, 12, 12]]
    mh = [[O, O, O, O, O, O, C, C, C, O, O, O], [H, H, H, O, O, O, H, H, H, H, O, O]]
    scs.append(dict(name="score_hidden_triple", N=2, cards=hid, masked=mh, draw=[1], disc=[4], player=0,
                    phase="draw", actions=[25]))
    # open equal triple that was never collapsed (dealt that way) stays until the owner places: multi-collapse
    mc = [[3, 3, 3, 6, 6, 6, 1, 2, 9, 4, 4, 5], [9, 9, 9, 1, 1, 1, 2, 3, 4, 12, 12, 12]]
    mm = [[O, O, O, O, O, O, H, H, H, H, H, H], [H, H, H, O, O, O, H, H, H, H, O, O]]
    scs.append(dict(name="multi_collapse", N=2, cards=mc, masked=mm, draw=[1, 7, 8], disc=[4], player=0,
                    phase="draw", indirect=False, actions=[24, 12 + 6, 24, 12 + 0]))
    # direct observation mode counts open cards of every player in hist15 (skyjo.py:160,236-248)
    scs.append(dict(name="direct_mode_hist", N=3, cards=base_cards3, masked=m_open2, draw=[4, 5, 6, 8], disc=[7, 0, 0],
                    player=1, phase="draw", indirect=False, actions=[25, 0, 24, 12 + 4, 24, 2]))
    # N=1 with 127 clip candidates: all open 12s (sum 144 -> obs[0] clipped to 127, skyjo.py:182)
    big = [[12, 12, 11, 12, 11, 12, 12, 11, 12, 11, 12, 12]]
    scs.append(dict(name="clip127", N=1, cards=big, masked=[[O] * 11 + [H]], draw=[1, 2], disc=[4], player=0,
                    phase="draw", actions=[24, 12 + 11, 24]))
    out = {"names": np.asarray([s["name"] for s in scs])}
    for s in scs:
        init, per, obs_f, mask_f, fs, nr, npl, term = run_scenario(s)
        p = s["name"] + "/"
        out[p + "cfg"] = np.asarray([s["N"], int(s.get("indirect", True)), s.get("np_seed", -1)], dtype=np.int64)
        out[p + "penalty"] = np.asarray(s.get("penalty", 2.0), dtype=np.float64)
        out[p + "actions"] = np.asarray(s["actions"], dtype=np.int32)
        for k, v in init.items():
            out[p + "init_" + k] = np.asarray(v)
        for k, v in per.items():
            out[p + "step_" + k] = np.asarray(v)
        out[p + "final_obs"] = np.asarray(obs_f, dtype=np.int8)
        out[p + "final_mask"] = np.asarray(mask_f, dtype=np.int8)
        out[p + "final_score"] = fs
        out[p + "num_refunded"] = nr.astype(np.int32)
        out[p + "num_placed"] = npl.astype(np.int32)
        out[p + "terminated"] = np.asarray(int(term))
        if term:
            out[p + "rewards"] = np.stack([calc_rewards(list(fs), list(nr), mr, rr) for mr, rr in REWARD_CFGS])
    np.savez_compressed(os.path.join(OUT, "scenarios.npz"), **out)
    return len(scs)


def gen_env(cfg, seed, name, illegal_at=None, illegal_action=None, episodes=2):
    """G4/G5: env() wrapper stack (skyjo_env.py:19-26) on the pettingzoo stand-ins.

    Flow of tests/environment/test_skyjo_env_jit.py:17-42: env.seed(s); rng = default_rng(s);
    env.reset(); agent_iter/last/step.  Episode 2 re-uses env.reset() without re-seeding.
    """
    e = skyjo_env.env(**cfg)
    N = cfg["num_players"]
    e.seed(seed)
    rng = np.random.default_rng(seed)
    rows = {k: [] for k in ("agent", "done", "cum_reward", "action", "obs", "mask")}
    ep_start = [0]
    for ep in range(episodes):
        e.reset()
        n = 0
        for agent in e.agent_iter(max_iter=300 * N):
            obs, reward, done, info = e.last()
            rows["agent"].append(int(agent.split("_")[-1]))
            rows["done"].append(int(done))
            rows["cum_reward"].append(float(reward))
            rows["obs"].append(np.asarray(obs["observations"], dtype=np.int8))
            rows["mask"].append(np.asarray(obs["action_mask"], dtype=np.int8))
            if not done:
                a = int(policy_ra(obs["observations"], obs["action_mask"], rng=rng))
                if illegal_at is not None and ep == 0 and n == illegal_at:
                    mask = np.asarray(obs["action_mask"])
                    a = int(np.flatnonzero(mask == 0)[illegal_action % int(np.sum(mask == 0))])
                rows["action"].append(a)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    e.step(a)
                n += 1
            else:
                rows["action"].append(-1)
                e.step(None)
        ep_start.append(len(rows["agent"]))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), kind="env", wrapper_semantics="unpinned (pettingzoo stand-ins)",
        num_players=N, seed=seed, indirect=int(cfg["observe_other_player_indirect"]),
        score_penalty=float(cfg["score_penalty"]), mean_reward=float(cfg["mean_reward"]),
        reward_refunded=float(cfg["reward_refunded"]), ep_start=np.asarray(ep_start, dtype=np.int32),
        agent=np.asarray(rows["agent"], dtype=np.int32), done=np.asarray(rows["done"], dtype=np.int32),
        cum_reward=np.asarray(rows["cum_reward"], dtype=np.float64),
        action=np.asarray(rows["action"], dtype=np.int32), obs=np.asarray(rows["obs"], dtype=np.int8),
        mask=np.asarray(rows["mask"], dtype=np.int8))
    return len(rows["agent"])


def _np_state():
    st = np.random.get_state(legacy=True)
    return np.asarray(st[1], dtype=np.uint32).copy(), int(st[2])


def gen_global_core(N, np_seed, indirect, episodes, name, penalty=2.0):
    """sample_run (rlskyjo/game/sample_game.py:5-28) on the process-global stream: np.random.seed(np_seed); SkyjoGame(...)
    deals in its constructor (skyjo.py:49); every game starts with reset(); policy_ra(obs, mask) draws from np.random."""
    np.random.seed(np_seed)
    g = SkyjoGame(num_players=N, score_penalty=penalty, observe_other_player_indirect=indirect)
    rows = {k: [] for k in ("player", "phase", "action", "obs", "mask", "over", "n_draw")}
    ep_start, keys, poss, finals, deals = [0], [], [], [], []
    for ep in range(episodes):
        g.reset()
        deals.append(np.array(g.players_cards, dtype=np.int8).copy())
        while not g.is_terminated:
            pid, phase = g.expected_action
            obs, mask = g.collect_observation(pid)
            a = int(policy_ra(obs, mask))
            rows["player"].append(pid), rows["phase"].append(0 if phase == "draw" else 1), rows["action"].append(a)
            rows["obs"].append(np.asarray(obs, dtype=np.int8)), rows["mask"].append(np.asarray(mask, dtype=np.int8))
            rows["n_draw"].append(len(g.drawpile))
            rows["over"].append(int(bool(g.act(pid, a))))
        ep_start.append(len(rows["action"]))
        k, p = _np_state()
        keys.append(k), poss.append(p)
        finals.append(np.asarray(g.game_metrics["final_score"], dtype=np.float64))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), kind="global_core", num_players=N, np_seed=np_seed, indirect=int(indirect),
        score_penalty=float(penalty), ep_start=np.asarray(ep_start, dtype=np.int32),
        player=np.asarray(rows["player"], dtype=np.int32), phase=np.asarray(rows["phase"], dtype=np.int32),
        action=np.asarray(rows["action"], dtype=np.int32), obs=np.asarray(rows["obs"], dtype=np.int8),
        mask=np.asarray(rows["mask"], dtype=np.int8), over=np.asarray(rows["over"], dtype=np.int32),
        n_draw=np.asarray(rows["n_draw"], dtype=np.int32), deal_cards=np.asarray(deals, dtype=np.int8),
        final_score=np.asarray(finals, dtype=np.float64), end_key=np.asarray(keys, dtype=np.uint32),
        end_pos=np.asarray(poss, dtype=np.int32))
    return len(rows["action"])


def gen_global_env(cfg, np_seed, name, episodes=2):
    """simple_episode (rlskyjo/environment/vanilla_env_example.py:6-41) on the process-global stream, pettingzoo
    stand-ins underneath ("wrapper semantics unpinned"): np.random.seed(np_seed); env(**cfg) (its SkyjoGame deals in
    the constructor); reset(); agent_iter / last / step with policy_ra(obs, mask)."""
    np.random.seed(np_seed)
    e = skyjo_env.env(**cfg)
    N = cfg["num_players"]
    rows = {k: [] for k in ("agent", "done", "cum_reward", "action", "obs", "mask")}
    ep_start, keys, poss = [0], [], []
    for ep in range(episodes):
        e.reset()
        for agent in e.agent_iter(max_iter=300 * N):
            obs, reward, done, info = e.last()
            rows["agent"].append(int(agent.split("_")[-1])), rows["done"].append(int(done)), rows["cum_reward"].append(float(reward))
            rows["obs"].append(np.asarray(obs["observations"], dtype=np.int8)), rows["mask"].append(np.asarray(obs["action_mask"], dtype=np.int8))
            if not done:
                a = int(policy_ra(obs["observations"], obs["action_mask"]))
                rows["action"].append(a)
                e.step(a)
            else:
                rows["action"].append(-1)
                e.step(None)
        ep_start.append(len(rows["agent"]))
        k, p = _np_state()
        keys.append(k), poss.append(p)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), kind="global_env", wrapper_semantics="unpinned (pettingzoo stand-ins)",
        num_players=N, np_seed=np_seed, indirect=int(cfg["observe_other_player_indirect"]),
        score_penalty=float(cfg["score_penalty"]), mean_reward=float(cfg["mean_reward"]),
        reward_refunded=float(cfg["reward_refunded"]), ep_start=np.asarray(ep_start, dtype=np.int32),
        agent=np.asarray(rows["agent"], dtype=np.int32), done=np.asarray(rows["done"], dtype=np.int32),
        cum_reward=np.asarray(rows["cum_reward"], dtype=np.float64), action=np.asarray(rows["action"], dtype=np.int32),
        obs=np.asarray(rows["obs"], dtype=np.int8), mask=np.asarray(rows["mask"], dtype=np.int8),
        end_key=np.asarray(keys, dtype=np.uint32), end_pos=np.asarray(poss, dtype=np.int32))
    return len(rows["agent"])


def gen_global():
    n = gen_global_core(3, 7, True, 3, "global_core_N3_s7")
    n += gen_global_core(2, 11, False, 3, "global_core_N2_dir_s11")
    n += gen_global_core(12, 5, True, 1, "global_core_N12_s5")  # ~17 mid-game reshuffles between the policy's draws
    n += gen_global_env(dict(skyjo_env.DEFAULT_CONFIG), 7, "global_env_default_s7")
    return n


def render_strings(g):
    """Everything the reference can print about one state (skyjo.py:508-562)."""
    N = g.num_players
    return (g.render_table(), [g.render_player(p) for p in range(N)], [g.render_player(p, True) for p in range(N)])


def gen_render():
    """Render fixtures: strings produced by the reference's render utils (skyjo.py:508-602).

    "traj" cases replay a seeded game (set_seed + stored actions) and render at chosen steps incl. the terminated
    state; "state" cases render an injected state (a column already collapsed, empty discard pile, hidden cards next
    to refunded ones) after each action.  The consumer re-creates the state through its own API and compares `==`.
    """
    H, O, C = 2, 1, 0
    out = {}
    names = []

    def put(name, meta, tables, closed, opened):
        names.append(name)
        for k, v in meta.items():
            out[name + "/" + k] = np.asarray(v)
        out[name + "/table"] = np.asarray(tables, dtype=np.str_)
        out[name + "/player_closed"] = np.asarray(closed, dtype=np.str_)
        out[name + "/player_open"] = np.asarray(opened, dtype=np.str_)

    for N, seed, ind, pen in ((3, 42, True, 2.0), (2, 5, False, 1.0), (4, 9, True, 2.0)):
        g = SkyjoGame(N, pen, ind)
        g.set_seed(seed)
        rng = np.random.default_rng(seed)
        actions, at, T, Pc, Po = [], [], [], [], []
        t = 0
        while True:
            if t in (0, 1, 2, 17, 40) or g.is_terminated:
                tb, pc, po = render_strings(g)
                at.append(t), T.append(tb), Pc.append(pc), Po.append(po)
            if g.is_terminated:
                break
            pid = g.expected_action[0]
            obs, mask = g.collect_observation(pid)
            a = int(policy_ra(obs, mask, rng=rng))
            actions.append(a)
            g.act(pid, a)
            t += 1
        put("traj_N%d_s%d" % (N, seed), dict(kind="traj", cfg=[N, int(ind), seed], penalty=pen,
                                             actions=np.asarray(actions, dtype=np.int32),
                                             at=np.asarray(at, dtype=np.int32)), T, Pc, Po)

    cards = [[8, 8, 8, 1, 2, 3, -14, -14, -14, 7, 9, 10], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
    masked = [[O, O, H, O, H, H, C, C, C, H, O, O], [O, O, H, H, H, H, H, H, H, H, H, H]]
    for name, ind, draw, disc, player, phase, acts in (
            ("state_collapsed", True, [4, 5, 6, 12, -2], [7, 0, 0, 0], 0, "draw", [24, 12 + 2, 25, 0]),
            ("state_empty_discard", False, [4, 5, 6], [7], 1, "draw", [25, 4, 25, 12 + 4])):
        g = SkyjoGame(2, 2.0, ind)
        g.players_cards = np.asarray(cards, dtype=np.int8).copy()
        g.players_masked = np.asarray(masked, dtype=np.int8).copy()
        g.drawpile = [np.int8(x) for x in draw]
        g.discard_pile = [np.int8(x) for x in disc]
        g.hand_card = 15
        force_turn(g, player, phase)
        init = snapshot(g)
        T, Pc, Po = [], [], []
        for k in range(len(acts) + 1):
            tb, pc, po = render_strings(g)
            T.append(tb), Pc.append(pc), Po.append(po)
            if k < len(acts):
                g.act(g.expected_action[0], acts[k])
        meta = dict(kind="state", cfg=[2, int(ind), -1], penalty=2.0, actions=np.asarray(acts, dtype=np.int32),
                    at=np.arange(len(acts) + 1, dtype=np.int32))
        meta.update({"init_" + k: v for k, v in init.items()})
        put(name, meta, T, Pc, Po)
    out["names"] = np.asarray(names)
    out["explainer"] = np.asarray([SkyjoGame.render_action_explainer(a) for a in range(26)], dtype=np.str_)
    out["actions_help"] = np.asarray(SkyjoGame.render_actions(), dtype=np.str_)
    np.savez_compressed(os.path.join(OUT, "render.npz"), **out)
    return len(names)


def gen_policy_stats(games=6000):
    """Statistics of the reference's own loop: SkyjoGame + policy_ra, default settings (score_penalty 2, indirect
    observation - neither changes the play of a random policy), N = 2, 3, 4."""
    out = {}
    for N in (2, 3, 4):
        np.random.seed(1000 + N)              # deals and reshuffles: the process-global legacy stream (skyjo.py:81,101,135)
        rng = np.random.default_rng(77 + N)   # the policy's own generator (random_admissible_policy.py:24-25)
        g = SkyjoGame(num_players=N, score_penalty=2.0, observe_other_player_indirect=True)
        ep_len = np.zeros(games, dtype=np.int32)
        score = np.zeros((games, N), dtype=np.float64)
        refunded = np.zeros((games, N), dtype=np.int32)
        rank_counts = np.zeros((27, 26), dtype=np.int64)   # [number of legal actions][rank of the chosen one among them]
        for e in range(games):
            g.reset()
            steps = 0
            while True:
                pid, _ = g.expected_action
                obs, mask = g.collect_observation(pid)
                a = int(policy_ra(obs, mask, rng))
                m = np.asarray(mask) != 0
                assert m[a]
                rank_counts[int(m.sum()), int(m[:a].sum())] += 1
                steps += 1
                if g.act(pid, a):
                    break
            met = g.get_game_metrics()
            ep_len[e] = steps
            score[e] = np.asarray(met["final_score"], dtype=np.float64)
            refunded[e] = np.asarray(met["num_refunded"], dtype=np.int32)
        out[f"N{N}_ep_len"], out[f"N{N}_score"], out[f"N{N}_refunded"], out[f"N{N}_rank_counts"] = ep_len, score, refunded, rank_counts
    out["players"] = np.asarray([2, 3, 4], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "policy_stats.npz"), **out)
    return {f"N{N}": (float(out[f"N{N}_ep_len"].mean()), float(out[f"N{N}_refunded"].sum(1).mean())) for N in (2, 3, 4)}


def main():
    if "--policy-stats-only" in sys.argv:
        print("policy_stats", gen_policy_stats())
        return
    if "--render-only" in sys.argv:
        print("render", gen_render())
        return
    if "--global-only" in sys.argv:
        print("global", gen_global())
        return
    gen_rng_kat()
    total = 0
    for N in (1, 2, 3, 4):
        for seed in (0, 1, 42, 123456):
            for ind in (True, False):
                total += gen_traj(N, seed, ind)
    for N in (8, 12):
        for seed, ind in ((0, True), (42, False)):
            total += gen_traj(N, seed, ind, episodes=2)
    total += gen_traj(3, 7, True, penalty=1.0, tag="traj_N3_s7_ind_pen1")
    total += gen_traj(4, 9, False, penalty=1.3, tag="traj_N4_s9_dir_pen13")
    print("traj steps", total)
    print("dense", gen_dense(2, 3, True, [-2, 0, 5], 0.0, name="dense_N2_collapse_ind"))
    print("dense", gen_dense(3, 4, False, [0, 1], 0.0, name="dense_N3_collapse_dir"))
    print("dense", gen_dense(4, 5, True, [12, 12, 11], 0.3, name="dense_N4_collapse_stall"))
    print("dense", gen_dense(2, 6, True, list(range(-2, 13)), 0.97, draw_cut=6, name="dense_N2_reshuffle_ind"))
    print("dense", gen_dense(3, 8, False, list(range(-2, 13)), 0.95, draw_cut=3, name="dense_N3_reshuffle_dir"))
    print("dense", gen_dense(1, 2, True, [1, 1, 2], 0.9, draw_cut=2, name="dense_N1_mixed"))
    print("scenarios", gen_scenarios())
    print("env", gen_env(dict(skyjo_env.DEFAULT_CONFIG), 42, "env_default_s42"))
    print("env", gen_env(dict(num_players=2, score_penalty=1.0, observe_other_player_indirect=False,
                              mean_reward=-1, reward_refunded=0.01), 5, "env_N2_dir_s5"))
    print("env", gen_env(dict(num_players=4, score_penalty=2.0, observe_other_player_indirect=True,
                              mean_reward=0.0, reward_refunded=0.0), 11, "env_N4_ind_s11"))
    print("env", gen_env(dict(skyjo_env.DEFAULT_CONFIG), 3, "env_illegal_draw_s3", illegal_at=4, illegal_action=3))
    print("env", gen_env(dict(skyjo_env.DEFAULT_CONFIG), 3, "env_illegal_place_s3", illegal_at=7, illegal_action=25))
    print("render", gen_render())
    print("global", gen_global())
    print("policy_stats", gen_policy_stats())
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes", sz)


if __name__ == "__main__":
    main()
