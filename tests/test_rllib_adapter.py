"""the RLlib-style multi-agent dict view over the AEC env (third-party contract restated from memory:
unpinned), driven by the CPU oracle engine.  Checked against the AEC env itself: same observations on turn, the final
rewards of every seat arrive exactly once, sum to N * mean_reward, and done["__all__"] closes the episode."""
import warnings

import numpy as np

from skyjo_rl_amd import aec_env
from skyjo_rl_amd.policy import policy_ra
from skyjo_rl_amd.rllib_adapter import PettingZooEnvAdapter
from tests.oracle_engine import OracleEngine


import pytest


def _make(factory=OracleEngine, **cfg):
    return aec_env.SimpleSkyjoEnv(engine=factory(1, auto_reset=False, **cfg), wrapped=True, **cfg)


def _hip(*a, **k):
    from skyjo_rl_amd import SkyjoVecEnv
    return SkyjoVecEnv(*a, **k)


def test_adapter_episode_matches_aec_env():
    _episode(OracleEngine)


@pytest.mark.gpu
def test_adapter_episode_matches_aec_env_on_hip_engine():
    _episode(_hip)


@pytest.mark.gpu
def test_adapter_illegal_action_ends_the_episode_on_hip_engine():
    _illegal(_hip)


def _episode(factory):
    cfg = dict(num_players=3, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.0)
    ref = _make(factory, **cfg)
    ad = PettingZooEnvAdapter(_make(factory, **cfg))
    ref.seed(7), ad.seed(7)
    ref.reset()
    obs = ad.reset()
    rng_a, rng_b = np.random.default_rng(3), np.random.default_rng(3)
    total = {f"player_{i}": 0.0 for i in range(3)}
    steps = 0
    while True:
        (agent, o), = obs.items()
        assert agent == ref.agent_selection
        ro, rr, rd, _ = ref.last()
        np.testing.assert_array_equal(o["observations"], ro["observations"])
        np.testing.assert_array_equal(o["action_mask"], ro["action_mask"])
        a = int(policy_ra(o["observations"], o["action_mask"], rng=rng_a))
        assert a == int(policy_ra(ro["observations"], ro["action_mask"], rng=rng_b))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref.step(a)
            obs_d, rew_d, done_d, info_d = ad.step({agent: a})
        steps += 1
        for k, v in rew_d.items():
            total[k] += v
        if done_d["__all__"]:
            assert all(done_d[k] for k in total)           # every seat reported done, once
            break
        obs = {k: v for k, v in obs_d.items() if not done_d[k]}
        assert len(obs) == 1
    assert steps > 40
    assert abs(sum(total.values()) - 3 * 1.0) < 1e-9        # skyjo_env.py:307-312
    for agent in ref.agent_iter(max_iter=10):                # the reference env, drained the AEC way, agrees seat by seat
        _, r, d, _ = ref.last()
        assert d and abs(r - total[agent]) < 1e-12
        ref.step(None)


def test_adapter_illegal_action_ends_the_episode():
    _illegal(OracleEngine)


def _illegal(factory):
    cfg = dict(num_players=2, score_penalty=2.0, observe_other_player_indirect=False, mean_reward=1.0, reward_refunded=0.0)
    ad = PettingZooEnvAdapter(_make(factory, **cfg))
    ad.seed(3)
    (agent, o), = ad.reset().items()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        obs_d, rew_d, done_d, _ = ad.step({agent: 0})        # a place action in the draw phase: masked out
    assert done_d["__all__"] and rew_d[agent] == -1.0        # TerminateIllegalWrapper(illegal_reward=-1), skyjo_env.py:23
    other = [k for k in rew_d if k != agent][0]
    assert rew_d[other] == 0.0
