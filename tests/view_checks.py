"""Checks of the drop-in views (skyjo_rl_amd/aec_env.py, game.py) that run on ANY engine: the CPU suite passes the
oracle-backed engine (tests/oracle_engine.py), the GPU suite the HIP engine - same assertions, same golden vectors
recorded from the reference (SimpleSkyjoEnv on pettingzoo stand-ins: third-party wrapper semantics are "unpinned")."""
import os
import warnings
from itertools import product

import numpy as np
import pytest

from skyjo_rl_amd import aec_env
from skyjo_rl_amd.game import SkyjoGame
from skyjo_rl_amd.policy import policy_ra

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_env(engine_factory, wrapped=True, **cfg):
    eng = engine_factory(1, auto_reset=False, **cfg)
    return aec_env.SimpleSkyjoEnv(engine=eng, wrapped=wrapped, **cfg)


def check_env_golden(engine_factory, path):
    """tests/environment/test_skyjo_env_jit.py flow: env.seed(s); rng=default_rng(s); reset; agent_iter/last/step."""
    d = np.load(path)
    cfg = dict(num_players=int(d["num_players"]), score_penalty=float(d["score_penalty"]),
               observe_other_player_indirect=bool(d["indirect"]), mean_reward=float(d["mean_reward"]),
               reward_refunded=float(d["reward_refunded"]))
    e = make_env(engine_factory, **cfg)
    seed = int(d["seed"])
    e.seed(seed)
    rng = np.random.default_rng(seed)
    illegal = "illegal" in path
    row = 0
    for ep in range(len(d["ep_start"]) - 1):
        e.reset()
        for agent in e.agent_iter(max_iter=300 * cfg["num_players"]):
            obs, reward, done, info = e.last()
            assert int(agent.split("_")[-1]) == d["agent"][row], row
            assert int(done) == d["done"][row], row
            assert float(reward) == d["cum_reward"][row], (row, reward, d["cum_reward"][row])
            np.testing.assert_array_equal(obs["observations"], d["obs"][row], err_msg=f"row {row}")
            np.testing.assert_array_equal(obs["action_mask"], d["mask"][row], err_msg=f"row {row}")
            assert obs["observations"].dtype == np.int8 and obs["action_mask"].dtype == np.int8
            if not done:
                a = int(policy_ra(obs["observations"], obs["action_mask"], rng=rng))
                if illegal:
                    a = int(d["action"][row])  # the fixture replaced one sampled action by an illegal one
                assert a == d["action"][row], row
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    e.step(a)
            else:
                e.step(None)
            row += 1
        assert row == d["ep_start"][ep + 1]
        assert e.agents == []
    return e


def check_reproducibility(engine_factory):
    """tests/environment/test_skyjo_env_jit.py:10-45: two seeded runs give identical observations and rewards."""
    runs = []
    for _ in range(2):
        e = make_env(engine_factory, **aec_env.DEFAULT_CONFIG)
        e.seed(42)
        rng = np.random.default_rng(42)
        e.reset()
        obs_list, rewards = [], []
        for agent in e.agent_iter(max_iter=300 * 3):
            obs, reward, done, info = e.last()
            if not done:
                obs_list.append(obs["observations"])
                e.step(policy_ra(obs["observations"], obs["action_mask"], rng=rng))
            else:
                e.step(None)
                rewards.append(reward)
        runs.append((obs_list, rewards))
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    # anchors observed when the reference itself is run this way
    assert len(runs[0][0]) == 89
    np.testing.assert_allclose(runs[0][1], [29.66666666666667, -70.33333333333333, 43.66666666666667], rtol=0, atol=0)


def check_config_sweep(engine_factory, every=1):
    """tests/environment/test_skyjo_env_nojit.py:11-48: simple_episode (vanilla_env_example.py:6-41) over the 288-config
    grid terminates cleanly; `every` thins the grid for the CPU suite."""
    grid = product(range(1, 13), [1.0, 2.0], [True, False], [-1, 0.0, 1.0], [0.0, 0.01])
    rng = np.random.default_rng(0)
    ran = 0
    for count, (n, pen, ind, mr, rr) in enumerate(grid):
        if count % every:
            continue
        cfg = dict(num_players=n, score_penalty=pen, observe_other_player_indirect=ind, mean_reward=mr,
                   reward_refunded=rr)
        e = make_env(engine_factory, **cfg)
        e.reset()
        total, steps = 0.0, 0
        for agent in e.agent_iter(max_iter=300 * n):
            obs, reward, done, info = e.last()
            if not done:
                assert e.observation_space(agent)["observations"].shape == obs["observations"].shape
                e.step(policy_ra(obs["observations"], obs["action_mask"], rng=rng))
                steps += 1
            else:
                total += reward
                e.step(None)
        if e.agents == []:  # finished inside max_iter (N=12 games can exceed 300*N iterations)
            bonus = rr * sum(e.table.game_metrics["num_refunded"]) if rr else 0.0
            assert abs(total - (n * mr + bonus)) < 1e-9  # skyjo_env.py:307-312: rewards sum to N*mean_reward (+bonus)
        if count % 48 == 0:
            assert "render board" in e.render()
        ran += 1
        if hasattr(e._engine, "close"):
            e._engine.close()
    return ran


def check_call_order(engine_factory):
    """OrderEnforcing / AssertOutOfBounds behaviour of the wrapper stack (skyjo_env.py:22-25)."""
    e = make_env(engine_factory, **aec_env.DEFAULT_CONFIG)
    with pytest.raises(AssertionError):
        e.step(24)  # before reset
    e.reset()
    with pytest.raises(AssertionError):
        e.step(26)
    with pytest.raises(AssertionError):
        e.step(None)  # agent is not done
    e.seed(3)
    with pytest.raises(AssertionError):
        e.observe("player_0")  # seed() requires a new reset
    e.reset()
    assert e.agent_selection in e.possible_agents and e.num_agents == 3 and e.max_num_agents == 3
    sp = e.action_space(e.agent_selection)
    assert sp.contains(25) and not sp.contains(26)


def check_core_view(engine_factory, num_envs=1, index=0):
    """rlskyjo/game/sample_game.py loop on the SkyjoGame view (game `index` of an engine with `num_envs` games); data
    from a golden trajectory.  With a shared engine the other games must not move."""
    d = np.load(os.path.join(GOLDEN, "traj_N3_s42_ind.npz"))
    eng = engine_factory(num_envs, num_players=3, observe_other_player_indirect=True, auto_reset=False)
    if num_envs > 1:
        eng.seed(None, 1000)
    g = SkyjoGame(3, 2.0, True, engine=eng, index=index)
    others = [i for i in range(num_envs) if i != index]
    before = {i: eng.get_state(i) for i in others[:3]}
    g.set_seed(42)
    assert g.obs_shape == (31,) and g.action_mask_shape == (26,)
    np.testing.assert_array_equal(g.players_cards, d["deal_cards"][0])
    t = 0
    for e in range(2):  # two episodes back to back: the game's private stream carries over like the reference's global one
        if e > 0:
            g.reset()
        while not g.is_terminated:
            pid, phase = g.expected_action
            assert phase in ("draw", "place")
            obs, mask = g.collect_observation(pid)
            np.testing.assert_array_equal(obs, d["obs"][t])
            np.testing.assert_array_equal(mask, d["mask"][t])
            if t % 9 == 0:
                oo, mo = g.collect_observation((pid + 1) % 3)
                np.testing.assert_array_equal(oo, d["obs_other"][t])
                np.testing.assert_array_equal(mo, d["mask_other"][t])
            assert g.act(pid, int(d["action"][t])) == bool(d["game_over"][t])
            t += 1
        assert t == d["ep_start"][e + 1]
        m = g.get_game_metrics()
        np.testing.assert_array_equal(m["final_score"], d["final_score"][e])
        np.testing.assert_array_equal(m["num_refunded"], d["num_refunded"][e])
        np.testing.assert_array_equal(m["num_placed"], d["num_placed"][e])
    for i, s in before.items():  # nobody else moved
        now = eng.get_state(i)
        for k in ("cards", "masked", "draw", "disc"):
            np.testing.assert_array_equal(now[k], s[k])
        assert (now["hand"], now["player"], now["phase"]) == (s["hand"], s["player"], s["phase"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert g.act(g.expected_action[0], 24) is True  # skyjo.py:316-321
        assert w
    # the core's assertions (skyjo.py:310-334)
    g.reset()
    pid = g.expected_action[0]
    with pytest.raises(AssertionError):
        g.act((pid + 1) % 3, 24)
    with pytest.raises(AssertionError):
        g.act(pid, 26)
    with pytest.raises(AssertionError):
        g.act(pid, 3)
    g.act(pid, 24)
    with pytest.raises(AssertionError):
        g.act(pid, 25)
    # render helpers produce text and never raise (tests/game/test_skyjo.py:12-18)
    assert "action ids 0-25" in g.render_actions()
    assert "Player 0" in g.render_player(0) and "u" in g.render_player(0, render_cards_open=True)
    assert [SkyjoGame.render_action_explainer(a) for a in range(26)][24] == "draw from drawpile"
    assert "next turn: place" in g.render_table()
    return eng


def check_render_golden(engine_factory):
    """The reference's render utils (rlskyjo/game/skyjo.py:508-602) string for string: tests/golden/render.npz holds what
    the reference printed for fresh, mid-game, collapsed-column, empty-discard and terminated states."""
    d = np.load(os.path.join(GOLDEN, "render.npz"))
    assert [SkyjoGame.render_action_explainer(a) for a in range(26)] == [str(x) for x in d["explainer"]]
    assert SkyjoGame.render_actions() == str(d["actions_help"])
    checked = 0
    for name in d["names"]:
        p = str(name) + "/"
        N, ind, seed = (int(x) for x in d[p + "cfg"])
        eng = engine_factory(1, num_players=N, score_penalty=float(d[p + "penalty"]), observe_other_player_indirect=bool(ind),
                             auto_reset=False)
        g = SkyjoGame(N, float(d[p + "penalty"]), bool(ind), engine=eng, seed=0)
        if str(d[p + "kind"]) == "traj":
            g.set_seed(seed)
        else:
            nd, ns = int(d[p + "init_n_draw"]), int(d[p + "init_n_disc"])
            eng.set_state(0, d[p + "init_cards"], d[p + "init_masked"], d[p + "init_draw"][:nd], d[p + "init_disc"][:ns],
                          int(d[p + "init_hand"]), int(d[p + "init_player"]), int(d[p + "init_phase"]))
            g.sync()
        at = [int(x) for x in d[p + "at"]]
        actions = [int(a) for a in d[p + "actions"]]
        for t in range(len(actions) + 1):
            if t in at:
                k = at.index(t)
                assert g.render_table() == str(d[p + "table"][k]), (name, t)
                for q in range(N):
                    assert g.render_player(q) == str(d[p + "player_closed"][k][q]), (name, t, q)
                    assert g.render_player(q, True) == str(d[p + "player_open"][k][q]), (name, t, q)
                checked += 1
            if t < len(actions):
                g.act(g.expected_action[0], actions[t])
        assert g.is_terminated == (str(d[p + "kind"]) == "traj")
        if hasattr(eng, "close"):
            eng.close()
    return checked


def check_global_core(engine_factory, path):
    """The reference's sample_run loop (rlskyjo/game/sample_game.py:5-28) on numpy's PROCESS-GLOBAL stream -
    np.random.seed(s); SkyjoGame(...) (deals in its constructor); reset() per game; policy_ra(obs, mask) WITHOUT a generator
    (random_admissible_policy.py:22-23) - against tests/golden/global_core_*.npz recorded from the reference: every
    observation, mask, action (the policy's draw comes out of the shared stream, so it pins the stream's position after
    every deal and reshuffle) and the stream's state (key and pos of np.random.get_state()) at every episode end."""
    d = np.load(path)
    N, ind = int(d["num_players"]), bool(d["indirect"])
    saved = np.random.get_state()
    try:
        np.random.seed(int(d["np_seed"]))
        eng = engine_factory(1, num_players=N, score_penalty=float(d["score_penalty"]), observe_other_player_indirect=ind,
                             auto_reset=False, no_bank=True)
        g = SkyjoGame(num_players=N, score_penalty=float(d["score_penalty"]), observe_other_player_indirect=ind, engine=eng,
                      global_rng=True)
        row = 0
        for ep in range(len(d["ep_start"]) - 1):
            g.reset()
            np.testing.assert_array_equal(g.players_cards, d["deal_cards"][ep], err_msg=f"deal of episode {ep}")
            while not g.is_terminated:
                pid, phase = g.expected_action
                obs, mask = g.collect_observation(pid)
                assert (pid, 0 if phase == "draw" else 1) == (d["player"][row], d["phase"][row]), row
                np.testing.assert_array_equal(obs, d["obs"][row], err_msg=f"row {row}")
                np.testing.assert_array_equal(mask, d["mask"][row], err_msg=f"row {row}")
                assert len(g.drawpile) == d["n_draw"][row], row
                a = int(policy_ra(obs, mask))
                assert a == d["action"][row], (row, a, d["action"][row])
                assert int(bool(g.act(pid, a))) == d["over"][row], row
                row += 1
            assert row == d["ep_start"][ep + 1]
            np.testing.assert_array_equal(np.asarray(g.game_metrics["final_score"], dtype=np.float64), d["final_score"][ep])
            st = np.random.get_state(legacy=True)
            assert int(st[2]) == d["end_pos"][ep], (ep, st[2], d["end_pos"][ep])
            np.testing.assert_array_equal(np.asarray(st[1], dtype=np.uint32), d["end_key"][ep], err_msg=f"stream after episode {ep}")
        return eng
    finally:
        np.random.set_state(saved)


def check_global_env(engine_factory, path):
    """The reference's simple_episode loop (rlskyjo/environment/vanilla_env_example.py:6-41) on numpy's process-global
    stream against tests/golden/global_env_*.npz (pettingzoo stand-ins underneath: wrapper semantics unpinned)."""
    d = np.load(path)
    cfg = dict(num_players=int(d["num_players"]), score_penalty=float(d["score_penalty"]),
               observe_other_player_indirect=bool(d["indirect"]), mean_reward=float(d["mean_reward"]),
               reward_refunded=float(d["reward_refunded"]))
    saved = np.random.get_state()
    try:
        np.random.seed(int(d["np_seed"]))
        eng = engine_factory(1, auto_reset=False, no_bank=True, **cfg)
        e = aec_env.SimpleSkyjoEnv(engine=eng, wrapped=True, global_rng=True, **cfg)
        row = 0
        for ep in range(len(d["ep_start"]) - 1):
            e.reset()
            for agent in e.agent_iter(max_iter=300 * cfg["num_players"]):
                obs, reward, done, info = e.last()
                assert int(agent.split("_")[-1]) == d["agent"][row] and int(done) == d["done"][row], row
                assert float(reward) == d["cum_reward"][row], (row, reward, d["cum_reward"][row])
                np.testing.assert_array_equal(obs["observations"], d["obs"][row], err_msg=f"row {row}")
                np.testing.assert_array_equal(obs["action_mask"], d["mask"][row], err_msg=f"row {row}")
                if not done:
                    a = int(policy_ra(obs["observations"], obs["action_mask"]))
                    assert a == d["action"][row], row
                    e.step(a)
                else:
                    e.step(None)
                row += 1
            assert row == d["ep_start"][ep + 1]
            st = np.random.get_state(legacy=True)
            assert int(st[2]) == d["end_pos"][ep]
            np.testing.assert_array_equal(np.asarray(st[1], dtype=np.uint32), d["end_key"][ep])
        return eng
    finally:
        np.random.set_state(saved)
