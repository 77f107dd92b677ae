"""RLlib-style multi-agent dict view of the AEC env, without Ray.

``rlskyjo/models/train_model_simple_rllib.py:22-59`` registers the reference env through
``ray.rllib.env.PettingZooEnv`` (ray==1.9.2, absent from this image).  This class restates that adapter's contract
from memory of ray 1.9.2 (last bullet - third-party semantics, unpinned):

* ``reset()``            -> ``{agent_selection: observe(agent_selection)}``
* ``step({agent: a})``   -> ``env.step(a)``; then the observation / reward / done / info of the NEW ``agent_selection`` is
  collected from ``env.last()``; while that agent is done its entry is recorded and it is drained with
  ``env.step(None)``; ``done["__all__"] = not env.agents``.

Rewards are the per-step values of ``last()`` (the AEC env's cumulative reward of the agent since it last acted), as
in the adapter.  It wraps ``skyjo_rl_amd.aec_env.env(**cfg)`` - the MI355X-backed drop-in for ``rlskyjo``'s
``skyjo_env.env`` - or any object with the same AEC surface.
"""


class PettingZooEnvAdapter:
    def __init__(self, aec_env):
        self.env = aec_env
        self.env.reset()
        self.agents = list(self.env.possible_agents)
        first = self.agents[0]
        self.observation_space = self.env.observation_space(first)
        self.action_space = self.env.action_space(first)
        self._agent_ids = set(self.agents)

    def seed(self, seed=None):
        self.env.seed(seed)

    def reset(self):
        self.env.reset()
        a = self.env.agent_selection
        return {a: self.env.observe(a)}

    def step(self, action_dict):
        assert len(action_dict) == 1 and self.env.agent_selection in action_dict, "one action, for the agent on turn"
        self.env.step(action_dict[self.env.agent_selection])
        obs_d, rew_d, done_d, info_d = {}, {}, {}, {}
        while self.env.agents:
            obs, rew, done, info = self.env.last()
            a = self.env.agent_selection
            obs_d[a], rew_d[a], done_d[a], info_d[a] = obs, rew, done, info
            if done:
                self.env.step(None)  # drain
            else:
                break
        done_d["__all__"] = not self.env.agents
        return obs_d, rew_d, done_d, info_d

    def render(self, mode="human"):
        return self.env.render(mode)

    def close(self):
        self.env.close()
