"""CPU oracle (oracle/skyjo_oracle.c) vs golden vectors recorded from the real reference.

The fixtures in tests/golden were produced by oracle/gen_golden.py, which imports rlskyjo from
/root/reference in the build container.  These tests pin the oracle; the GPU parity tests then
compare the HIP path with the pinned oracle.
"""
import glob
import os

import numpy as np
import pytest

from oracle import skyjo_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")) + glob.glob(os.path.join(GOLDEN, "dense_*.npz")))


def test_rng_kat():
    """numpy legacy RandomState stream (skyjo.py:81,94,101,135)."""
    d = np.load(os.path.join(GOLDEN, "rng_kat.npz"))
    L = so.lib()
    for i, seed in enumerate(d["seeds"]):
        g = so.OracleGame(3)
        g.seed_legacy_raw(int(seed))
        raw = np.array([L.sko_rng_next(g.g.rng) for _ in range(700)], dtype=np.uint32)
        np.testing.assert_array_equal(raw, d["raw"][i])
        g.seed_legacy_raw(int(seed))
        a = np.arange(150, dtype=np.int32)
        L.sko_shuffle_i32(g.g.rng, a.ctypes.data, 150)
        np.testing.assert_array_equal(a, d["shuf150"][i])
        b = np.arange(114, dtype=np.int32)
        L.sko_shuffle_i32(g.g.rng, b.ctypes.data, 114)
        np.testing.assert_array_equal(b, d["shuf114"][i])
        for k in range(3):
            p = np.arange(12, dtype=np.int32)
            L.sko_shuffle_i32(g.g.rng, p.ctypes.data, 12)
            np.testing.assert_array_equal(p[:2], d["perm12"][i][k])


def _cmp_snapshot(snap, d, prefix, e):
    np.testing.assert_array_equal(snap["cards"], d[prefix + "cards"][e])
    np.testing.assert_array_equal(snap["masked"], d[prefix + "masked"][e])
    assert snap["n_draw"] == d[prefix + "n_draw"][e]
    assert snap["n_disc"] == d[prefix + "n_disc"][e]
    np.testing.assert_array_equal(snap["draw"], d[prefix + "draw"][e][: snap["n_draw"]])
    np.testing.assert_array_equal(snap["disc"], d[prefix + "disc"][e][: snap["n_disc"]])
    assert snap["hand"] == d[prefix + "hand"][e]
    assert snap["player"] == d[prefix + "player"][e]
    assert snap["phase"] == d[prefix + "phase"][e]


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[:-4] for p in TRAJ])
def test_trajectory(path):
    """Seeded games replayed action by action: obs, mask, piles, scores, rewards all bit-equal."""
    d = np.load(path)
    N, seed, ind = int(d["num_players"]), int(d["seed"]), bool(d["indirect"])
    dense = str(d["kind"]) == "dense"
    g = so.OracleGame(N, float(d["score_penalty"]), ind)
    g.set_seed(seed)
    E = len(d["ep_start"]) - 1
    for e in range(E):
        if e > 0:
            g.reset()
        if dense:  # injected deck: same RNG consumption, then overwrite the dealt state
            nd, ns = int(d["deal_n_draw"][e]), int(d["deal_n_disc"][e])
            g.set_state(d["deal_cards"][e], d["deal_masked"][e], d["deal_draw"][e][:nd], d["deal_disc"][e][:ns],
                        int(d["deal_hand"][e]), int(d["deal_player"][e]), int(d["deal_phase"][e]))
        _cmp_snapshot(g.snapshot(), d, "deal_", e)
        for t in range(int(d["ep_start"][e]), int(d["ep_start"][e + 1])):
            pid, phase = g.expected_action
            assert (pid, phase) == (d["player"][t], d["phase"][t]), t
            assert g.g.hand_card == d["hand"][t] and g.g.n_draw == d["n_draw"][t] and g.g.n_disc == d["n_disc"][t], t
            obs, mask = g.collect_observation(pid)
            np.testing.assert_array_equal(obs, d["obs"][t], err_msg=f"obs t={t}")
            np.testing.assert_array_equal(mask, d["mask"][t], err_msg=f"mask t={t}")
            obs_o, mask_o = g.collect_observation((pid + 1) % N)
            np.testing.assert_array_equal(obs_o, d["obs_other"][t])
            np.testing.assert_array_equal(mask_o, d["mask_other"][t])
            assert g.act(pid, int(d["action"][t])) == d["game_over"][t], t
        assert g.is_terminated
        _cmp_snapshot(g.snapshot(), d, "end_", e)
        np.testing.assert_array_equal(g.final_score(), d["final_score"][e])
        nr, npl = g.metrics()
        np.testing.assert_array_equal(nr, d["num_refunded"][e])
        np.testing.assert_array_equal(npl, d["num_placed"][e])
        for k, (mr, rr) in enumerate(d["reward_cfgs"]):
            np.testing.assert_array_equal(g.final_rewards(mr, rr), d["rewards"][e][k])
        # trap 18: acting on a terminated game returns True and changes nothing
        before = g.snapshot()
        assert g.act(g.expected_action[0], 24) == 1
        after = g.snapshot()
        for k in before:
            np.testing.assert_array_equal(before[k], after[k])


def test_scenarios():
    """Hand-built states covering the parity traps of SURVEY.md section 8.1."""
    d = np.load(os.path.join(GOLDEN, "scenarios.npz"))
    for name in d["names"]:
        p = str(name) + "/"
        N, ind, np_seed = (int(x) for x in d[p + "cfg"])
        g = so.OracleGame(N, float(d[p + "penalty"]), bool(ind))
        nd, ns = int(d[p + "init_n_draw"]), int(d[p + "init_n_disc"])
        g.set_state(d[p + "init_cards"], d[p + "init_masked"], d[p + "init_draw"][:nd], d[p + "init_disc"][:ns],
                    int(d[p + "init_hand"]), int(d[p + "init_player"]), int(d[p + "init_phase"]))
        if np_seed >= 0:
            g.seed_legacy_raw(np_seed)
        for t, a in enumerate(d[p + "actions"]):
            pid = g.expected_action[0]
            obs, mask = g.collect_observation(pid)
            np.testing.assert_array_equal(obs, d[p + "step_obs"][t], err_msg=f"{name} obs t={t}")
            np.testing.assert_array_equal(mask, d[p + "step_mask"][t], err_msg=f"{name} mask t={t}")
            assert g.act(pid, int(a)) == d[p + "step_over"][t], (name, t)
            s = g.snapshot()
            np.testing.assert_array_equal(s["cards"], d[p + "step_cards"][t], err_msg=f"{name} t={t}")
            np.testing.assert_array_equal(s["masked"], d[p + "step_masked"][t], err_msg=f"{name} t={t}")
            assert s["n_draw"] == d[p + "step_n_draw"][t] and s["n_disc"] == d[p + "step_n_disc"][t], (name, t)
            np.testing.assert_array_equal(s["draw"], d[p + "step_draw"][t][: s["n_draw"]], err_msg=f"{name} t={t}")
            np.testing.assert_array_equal(s["disc"], d[p + "step_disc"][t][: s["n_disc"]], err_msg=f"{name} t={t}")
            assert (s["hand"], s["player"], s["phase"]) == (
                d[p + "step_hand"][t], d[p + "step_player"][t], d[p + "step_phase"][t]), (name, t)
        obs, mask = g.collect_observation(g.expected_action[0])
        np.testing.assert_array_equal(obs, d[p + "final_obs"], err_msg=str(name))
        np.testing.assert_array_equal(mask, d[p + "final_mask"], err_msg=str(name))
        assert g.is_terminated == bool(d[p + "terminated"])
        nr, npl = g.metrics()
        np.testing.assert_array_equal(nr, d[p + "num_refunded"])
        np.testing.assert_array_equal(npl, d[p + "num_placed"])
        if g.is_terminated:
            np.testing.assert_array_equal(g.final_score(), d[p + "final_score"])
            cfgs = [(1.0, 0.001), (0.0, 0.0), (-1.0, 0.01), (1.0, 0.0)]
            for k, (mr, rr) in enumerate(cfgs):
                np.testing.assert_array_equal(g.final_rewards(mr, rr), d[p + "rewards"][k])


def test_notebook_score_kat():
    """The single known-answer datum in the reference: notebooks/trainpettingzoo.ipynb:52745-52758."""
    cards = np.array([[-1, 9, 7, -2, 4, 2, 0, 7, 4, 0, 3, 5], [0, 7, 1, 10, 7, 2, 0, 6, 1, -1, -1, 9],
                      [-1, 6, 5, -2, 4, 2, 1, 4, -2, 3, -2, 3]], dtype=np.int8)
    score = np.zeros(3)
    so.lib().sko_evaluate_game(cards.ctypes.data, 3, 0, 2.0, score.ctypes.data)
    np.testing.assert_array_equal(score, [76.0, 41.0, 21.0])


def test_act_assertions():
    """skyjo.py:310-334,399: the core's AssertionErrors are negative return codes, state untouched."""
    g = so.OracleGame(3, 2.0, True)
    g.set_seed(42)
    pid = g.expected_action[0]
    before = g.snapshot()
    assert g.act((pid + 1) % 3, 24) == -1
    assert g.act(pid, 26) == -2 and g.act(pid, -1) == -2
    assert g.act(pid, 3) == -4  # place without a hand card
    for k in before:
        np.testing.assert_array_equal(before[k], g.snapshot()[k])
    assert g.act(pid, 24) == 0
    assert g.act(pid, 25) == -3  # draw while holding a card
    open_slot = int(np.flatnonzero(before["masked"][pid] == 1)[0])
    assert g.act(pid, 12 + open_slot) == -5  # reveal an already open card
