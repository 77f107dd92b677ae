"""PettingZoo-AEC compatible single-game view (drop-in for rlskyjo.environment.skyjo_env).

``env(**config)`` returns an object with the surface the reference builds from
``SimpleSkyjoEnv`` plus four PettingZoo wrappers (rlskyjo/environment/skyjo_env.py:19-26):
``reset / step / observe / last / agent_iter / seed / render / close``, the attributes
``agents, possible_agents, agent_selection, rewards, _cumulative_rewards, dones, infos,
num_agents`` and ``observation_space(agent) / action_space(agent)``.  PettingZoo itself is not a
dependency: the AEC bookkeeping of pettingzoo==1.14.0 (requirements.txt:4) that the reference
inherits - reward accumulation, done-agent draining, illegal-move termination, call-order checks -
is restated here so that ``vanilla_env_example.simple_episode``-style loops
(rlskyjo/environment/vanilla_env_example.py:6-41) and the seeded reproducibility flow
(tests/environment/test_skyjo_env_jit.py:10-45) run unchanged.  Those third-party semantics are
unpinned by the reference's tests; the game semantics underneath are the pinned HIP
engine.

All game logic runs in the engine (libskyjo_vec.so on the GPU); this file is host bookkeeping.
"""
import io
import warnings
from contextlib import redirect_stdout

import numpy as np

from ._lib import ST_ILLEGAL
from .game import SkyjoGame
from .spaces import Box, Dict, Discrete

# rlskyjo/environment/skyjo_env.py:10-16
DEFAULT_CONFIG = {
    "num_players": 3,
    "score_penalty": 2.0,
    "observe_other_player_indirect": True,
    "mean_reward": 1.0,
    "reward_refunded": 0.001,
}


class _AgentIter:
    def __init__(self, env, max_iter):
        self._env, self._left = env, max_iter

    def __iter__(self):
        return self

    def __next__(self):
        if not self._env.agents or self._left <= 0:
            raise StopIteration
        self._left -= 1
        return self._env.agent_selection


class SimpleSkyjoEnv:
    """SimpleSkyjoEnv(AECEnv) of skyjo_env.py:29-334 on top of a batched engine with one game."""

    metadata = {
        "render.modes": ["human"],
        "name": "skyjo",
        "is_parallelizable": False,
        "video.frames_per_second": 1,
    }

    def __init__(self, num_players=2, score_penalty: float = 2.0, observe_other_player_indirect: bool = False,
                 mean_reward: float = 1.0, reward_refunded: float = 0.0, engine=None, device=0, wrapped=False,
                 global_rng=False):
        """``global_rng=True``: deals and reshuffles draw from numpy's process-global stream like the reference's
        (game.py has the contract), so ``np.random.seed(s)`` + ``policy_ra(obs, mask)`` loops replay the reference."""
        self.num_players = num_players
        self.mean_reward = mean_reward
        self.reward_refunded = reward_refunded
        if engine is None:
            from .vec_env import SkyjoVecEnv

            engine = SkyjoVecEnv(1, num_players=num_players, score_penalty=float(score_penalty),
                                 observe_other_player_indirect=observe_other_player_indirect,
                                 mean_reward=float(mean_reward), reward_refunded=float(reward_refunded),
                                 device=device, auto_reset=False, no_bank=bool(global_rng))
        self._engine = engine
        self.table = SkyjoGame(num_players, score_penalty=score_penalty,
                               observe_other_player_indirect=observe_other_player_indirect, engine=engine,
                               global_rng=global_rng)
        # wrapper behaviour of env() (skyjo_env.py:22-25) folded into this object when wrapped=True
        self._wrapped = wrapped
        self._has_reset = False
        self._illegal_terminated = False
        self._skip_agent_selection = None

        self.agents = [f"player_{i}" for i in range(num_players)]
        self.possible_agents = self.agents[:]
        self.agent_selection = self._expected_agentname_and_action()[0]
        self.dones = self._convert_to_dict([False for _ in range(self.num_agents)])
        self.infos = {i: {} for i in self.agents}
        # skyjo_env.py:125-151
        self._observation_spaces = self._convert_to_dict([
            Dict({
                "observations": Box(low=-24, high=127, shape=self.table.obs_shape, dtype=self.table.card_dtype),
                "action_mask": Box(low=0, high=1, shape=self.table.action_mask_shape, dtype=np.int8),
            })
            for _ in self.possible_agents
        ])
        self._action_spaces = self._convert_to_dict(
            [Discrete(self.table.action_mask_shape[0]) for _ in self.possible_agents])

    # ---- PettingZoo AECEnv surface ------------------------------------------------------------------
    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    @property
    def unwrapped(self):
        return self

    @property
    def observation_spaces(self):
        return self._observation_spaces

    @property
    def action_spaces(self):
        return self._action_spaces

    def observation_space(self, agent):
        return self._observation_spaces[agent]

    def action_space(self, agent):
        return self._action_spaces[agent]

    def observe(self, agent: str):
        """{"observations", "action_mask"} of `agent` (skyjo_env.py:199-214)."""
        if self._wrapped and not self._has_reset:
            raise AssertionError("reset() needs to be called before observe")
        obs, action_mask = self.table.collect_observation(self._name_to_player_id(agent))
        return {"observations": obs, "action_mask": action_mask}

    def last(self, observe=True):
        agent = self.agent_selection
        if agent is None:
            raise ValueError("no agent selected")
        observation = self.observe(agent) if observe else None
        return observation, self._cumulative_rewards[agent], self.dones[agent], self.infos[agent]

    def agent_iter(self, max_iter=2 ** 63):
        if self._wrapped and not self._has_reset:
            raise AssertionError("reset() needs to be called before agent_iter")
        return _AgentIter(self, max_iter)

    def step(self, action) -> None:
        """skyjo_env.py:216-252, preceded by the checks of the wrapper stack when built by env()."""
        if self._wrapped:
            if not self._has_reset:
                raise AssertionError("reset() needs to be called before step")
            if not self.agents:
                warnings.warn("step() called after all agents are done")
                return None
            done_now = self.dones[self.agent_selection]
            # AssertOutOfBoundsWrapper
            assert (action is None and done_now) or self.action_space(self.agent_selection).contains(action), \
                "action is not in action space"
            if self._illegal_terminated and done_now:
                return self._was_done_step(action)
            if not done_now and not self.observe(self.agent_selection)["action_mask"][action]:
                return self._terminate_illegal()  # TerminateIllegalWrapper(illegal_reward=-1)
        current_agent = self.agent_selection
        player_id = self._name_to_player_id(current_agent)
        if self.dones[current_agent]:
            return self._was_done_step(action)
        game_is_over = self.table.act(player_id, action_int=action)
        self.agent_selection = self._expected_agentname_and_action()[0]
        if game_is_over:
            self.rewards = self._convert_to_dict(self._final_rewards())
            self.dones = {i: True for i in self.agents}
        self._accumulate_rewards()
        self._clear_rewards()
        self._dones_step_first()

    def reset(self) -> None:
        """skyjo_env.py:254-267"""
        self.table.reset()
        self._has_reset = True
        self._illegal_terminated = False
        self._skip_agent_selection = None
        self.agents = self.possible_agents[:]
        self.agent_selection = self._expected_agentname_and_action()[0]
        self.rewards = self._convert_to_dict([0 for _ in range(self.num_agents)])
        self._cumulative_rewards = self._convert_to_dict([0 for _ in range(self.num_agents)])
        self.dones = self._convert_to_dict([False for _ in range(self.num_agents)])
        self.infos = {i: {} for i in self.agents}

    def render(self, mode="human"):
        if mode == "human":
            if self._wrapped:  # CaptureStdoutWrapper: return the text instead of printing it
                buf = io.StringIO()
                with redirect_stdout(buf):
                    print(self.table.render_table())
                return buf.getvalue()
            print(self.table.render_table())

    def close(self) -> None:
        pass

    def seed(self, seed: int = None) -> None:
        """skyjo_env.py:280-290; like OrderEnforcingWrapper a reset() is required afterwards."""
        if self._wrapped:
            self._has_reset = False
        if seed is not None:
            self.table.set_seed(seed)

    # ---- AEC bookkeeping of pettingzoo 1.14.0 ----------------------------------
    def _accumulate_rewards(self):
        for agent, reward in self.rewards.items():
            self._cumulative_rewards[agent] += reward

    def _clear_rewards(self):
        for agent in self.rewards:
            self.rewards[agent] = 0

    def _dones_step_first(self):
        order = [a for a in self.agents if self.dones[a]]
        if order:
            self._skip_agent_selection = self.agent_selection
            self.agent_selection = order[0]
        return self.agent_selection

    def _was_done_step(self, action):
        if action is not None:
            raise ValueError("when an agent is done, the only valid action is None")
        agent = self.agent_selection
        assert self.dones[agent], "an agent that was not done as attempted to be removed"
        del self.dones[agent], self.rewards[agent], self._cumulative_rewards[agent], self.infos[agent]
        self.agents.remove(agent)
        order = [a for a in self.agents if self.dones[a]]
        if order:
            if self._skip_agent_selection is None:
                self._skip_agent_selection = self.agent_selection
            self.agent_selection = order[0]
        else:
            if self._skip_agent_selection is not None:
                self.agent_selection = self._skip_agent_selection
            self._skip_agent_selection = None
        self._clear_rewards()

    def _terminate_illegal(self):
        """TerminateIllegalWrapper semantics: offender -1, others 0, everybody done (skyjo_env.py:23)."""
        warnings.warn("[WARNING]: Illegal move made, game terminating with current player losing.")
        current = self.agent_selection
        self._cumulative_rewards[current] = 0
        self.dones = {a: True for a in self.dones}
        self.rewards = {a: 0 for a in self.dones}
        self.rewards[current] = float(-1)
        self._accumulate_rewards()
        self._dones_step_first()
        self._illegal_terminated = True

    # ---- utils (skyjo_env.py:293-332) -------------------------------------------------------------------
    def _final_rewards(self):
        """_calc_final_rewards (skyjo_env.py:293-312): computed by the engine at game end, float64."""
        rewards, _, _ = self._engine.rewards_host()
        return np.array(rewards[0], dtype=np.float64)

    def _calc_final_rewards(self, final_score, num_refunded, **kwargs):
        """``SimpleSkyjoEnv._calc_final_rewards`` (skyjo_env.py:293-312) for caller-supplied results: reward relative to the
        mean score, + ``mean_reward``, + ``reward_refunded`` per refunded column.  Computed on the device
        (``skyjo_vec_calc_final_rewards``), float64 in numpy's operation order."""
        import ctypes as C

        from . import _lib

        score = np.ascontiguousarray(final_score, dtype=np.float64).ravel()
        ref = np.ascontiguousarray(num_refunded, dtype=np.int32).ravel()
        assert score.shape == ref.shape and 0 < score.shape[0] <= 12
        out = np.zeros_like(score)
        dev = getattr(self._engine, "device_index", 0)
        _lib.check(_lib.load().skyjo_vec_calc_final_rewards(int(dev), 1, int(score.shape[0]), score.ctypes.data_as(C.c_void_p),
                                                        ref.ctypes.data_as(C.c_void_p), float(self.mean_reward),
                                                        float(self.reward_refunded), out.ctypes.data_as(C.c_void_p)))
        return out

    @staticmethod
    def _name_to_player_id(name: str) -> int:
        return int(name.split("_")[-1])

    def _convert_to_dict(self, list_of_list):
        return dict(zip(self.possible_agents, list_of_list))

    def _expected_agentname_and_action(self):
        a = self.table.get_expected_action()
        return f"player_{a[0]}", a[1]


def env(**kwargs):
    """Factory with the behaviour of the reference's wrapper stack (skyjo_env.py:19-26)."""
    return SimpleSkyjoEnv(wrapped=True, **kwargs)
