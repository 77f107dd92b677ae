"""Stand-ins for pettingzoo.utils.wrappers used by rlskyjo/environment/skyjo_env.py:19-26."""
import contextlib
import io
import warnings

from ..aec import AECEnv, AECIterable


class BaseWrapper(AECEnv):
    def __init__(self, env):
        super().__init__()
        self.env = env
        self.possible_agents = env.possible_agents
        self.metadata = env.metadata
        try:
            self.infos = env.infos
        except AttributeError:
            pass

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def observation_space(self, agent):
        return self.env.observation_space(agent)

    def action_space(self, agent):
        return self.env.action_space(agent)

    def seed(self, seed=None):
        self.env.seed(seed)

    def close(self):
        self.env.close()

    def render(self, mode="human"):
        return self.env.render(mode)

    def _sync(self):
        self.agent_selection = self.env.agent_selection
        self.rewards = self.env.rewards
        self.dones = self.env.dones
        self.infos = self.env.infos
        self.agents = self.env.agents
        self._cumulative_rewards = self.env._cumulative_rewards

    def reset(self):
        self.env.reset()
        self._sync()

    def observe(self, agent):
        return self.env.observe(agent)

    def step(self, action):
        self.env.step(action)
        self._sync()


class CaptureStdoutWrapper(BaseWrapper):
    def render(self, mode="human"):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            super().render(mode)
        return buf.getvalue()


class TerminateIllegalWrapper(BaseWrapper):
    def __init__(self, env, illegal_reward):
        super().__init__(env)
        self._illegal_value = illegal_reward
        self._prev_obs = None

    def reset(self):
        self._terminated = False
        self._prev_obs = None
        super().reset()

    def observe(self, agent):
        obs = super().observe(agent)
        if agent == self.agent_selection:
            self._prev_obs = obs
        return obs

    def step(self, action):
        current_agent = self.agent_selection
        if self._prev_obs is None:
            self.observe(self.agent_selection)
        assert "action_mask" in self._prev_obs
        mask = self._prev_obs["action_mask"]
        self._prev_obs = None
        if self._terminated and self.dones[self.agent_selection]:
            self._was_done_step(action)
        elif not self.dones[self.agent_selection] and not mask[action]:
            warnings.warn("[WARNING]: Illegal move made, game terminating with current player losing.")
            self._cumulative_rewards[self.agent_selection] = 0
            self.dones = {d: True for d in self.dones}
            self._prev_obs = None
            self.rewards = {d: 0 for d in self.dones}
            self.rewards[current_agent] = float(self._illegal_value)
            self._accumulate_rewards()
            self._dones_step_first()
            self._terminated = True
        else:
            super().step(action)


class AssertOutOfBoundsWrapper(BaseWrapper):
    def step(self, action):
        assert (action is None and self.dones[self.agent_selection]) or self.action_space(
            self.agent_selection
        ).contains(action), "action is not in action space"
        super().step(action)


class OrderEnforcingWrapper(BaseWrapper):
    def __init__(self, env):
        self._has_reset = False
        super().__init__(env)

    def seed(self, seed=None):
        self._has_reset = False
        super().seed(seed)

    def step(self, action):
        if not self._has_reset:
            raise AssertionError("reset() needs to be called before step")
        if not self.agents:
            warnings.warn("step() called after all agents are done")
            return None
        super().step(action)

    def observe(self, agent):
        if not self._has_reset:
            raise AssertionError("reset() needs to be called before observe")
        return super().observe(agent)

    def agent_iter(self, max_iter=2 ** 63):
        if not self._has_reset:
            raise AssertionError("reset() needs to be called before agent_iter")
        return AECIterable(self, max_iter)

    def reset(self):
        self._has_reset = True
        super().reset()
