class AECIterable:
    def __init__(self, env, max_iter):
        self.env, self.max_iter = env, max_iter

    def __iter__(self):
        return AECIterator(self.env, self.max_iter)


class AECIterator:
    def __init__(self, env, max_iter):
        self.env, self.iters_til_term = env, max_iter

    def __next__(self):
        if not self.env.agents or self.iters_til_term <= 0:
            raise StopIteration
        self.iters_til_term -= 1
        return self.env.agent_selection

    def __iter__(self):
        return self


class AECEnv:
    def __init__(self):
        pass

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    def seed(self, seed=None):
        pass

    def close(self):
        pass

    def _dones_step_first(self):
        order = [a for a in self.agents if self.dones[a]]
        if order:
            self._skip_agent_selection = self.agent_selection
            self.agent_selection = order[0]
        return self.agent_selection

    def _clear_rewards(self):
        for a in self.rewards:
            self.rewards[a] = 0

    def _accumulate_rewards(self):
        for a, r in self.rewards.items():
            self._cumulative_rewards[a] += r

    def agent_iter(self, max_iter=2 ** 63):
        return AECIterable(self, max_iter)

    def last(self, observe=True):
        agent = self.agent_selection
        if agent is None:
            raise ValueError("no agent selected")
        observation = self.observe(agent) if observe else None
        return observation, self._cumulative_rewards[agent], self.dones[agent], self.infos[agent]

    def _was_done_step(self, action):
        if action is not None:
            raise ValueError("when an agent is done, the only valid action is None")
        agent = self.agent_selection
        assert self.dones[agent], "an agent that was not done as attempted to be removed"
        del self.dones[agent]
        del self.rewards[agent]
        del self._cumulative_rewards[agent]
        del self.infos[agent]
        self.agents.remove(agent)
        order = [a for a in self.agents if self.dones[a]]
        if order:
            if getattr(self, "_skip_agent_selection", None) is None:
                self._skip_agent_selection = self.agent_selection
            self.agent_selection = order[0]
        else:
            if getattr(self, "_skip_agent_selection", None) is not None:
                self.agent_selection = self._skip_agent_selection
            self._skip_agent_selection = None
        self._clear_rewards()
