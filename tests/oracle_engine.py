"""Oracle-backed stand-in for SkyjoVecEnv's host-style API (TEST INFRASTRUCTURE).

Lets the CPU test-suite exercise the host-side logic of skyjo_rl_amd (AEC bookkeeping, SkyjoGame
view, sharding / statistics gather) without a GPU.  It is never importable from the product: the
product's engine is the HIP library and nothing else.
"""
import numpy as np

from oracle import skyjo_oracle as so
from skyjo_rl_amd.vec_env import Obs


class OracleEngine:
    def __init__(self, num_envs, num_players=3, score_penalty=2.0, observe_other_player_indirect=True,
                 mean_reward=1.0, reward_refunded=0.001, rng_mode=so.RNG_MT19937, auto_reset=True, game_id0=0,
                 no_bank=False, **_):
        self.no_bank = bool(no_bank)  # (the oracle never deals ahead: its stream is numpy's as it stands)
        self.v = so.OracleVec(num_envs=num_envs, num_players=num_players, score_penalty=score_penalty,
                              observe_other_player_indirect=observe_other_player_indirect, mean_reward=mean_reward,
                              reward_refunded=reward_refunded, rng_mode=rng_mode, auto_reset=auto_reset,
                              game_id0=game_id0)
        self.num_envs, self.num_players = num_envs, num_players
        self.obs_dim = self.v.obs_dim
        self.obs_shape = (self.obs_dim,)
        self.action_mask_shape = (26,)
        self.game_id0 = game_id0

    def _obs(self, players=None):
        obs, mask, agent, phase = self.v.observe(players)
        ep = np.array([self.v.v.contents.ep_len[i] for i in range(self.num_envs)], dtype=np.uint16)
        return Obs(obs, mask, agent, phase, self.v.dones, self.v.status, ep)

    ACTION_SKIP = so.ACTION_SKIP

    def seed(self, seeds=None, base_seed=0):
        self.v.seed(seeds, base_seed)

    def seed_one(self, game, value):
        self.v.seed_one(game, value)

    def set_state(self, game, cards, masked, draw, disc, hand=15, player=0, phase=0, **_):
        self.v.set_state(game, cards, masked, draw, disc, hand, player, phase)

    def reset_host(self, mask=None):
        self.v.reset(mask)
        return self._obs()

    def step_host(self, actions):
        self.v.step(actions)
        return self._obs()

    def observe_host(self, players=None):
        return self._obs(players)

    def rewards_host(self):
        N = self.num_players
        sc = np.array([[self.v.game(i).final_score[p] for p in range(N)] for i in range(self.num_envs)])
        return self.v.rewards, sc, self.v.dones

    def close(self):
        pass

    def rng_set(self, game, key, pos):
        r = self.v.game(game).rng
        for k, w in enumerate(np.asarray(key, dtype=np.uint32).tolist()):
            r.mt[k] = w
        r.idx = int(pos)

    def rng_get(self, game):
        r = self.v.game(game).rng
        return np.array(list(r.mt), dtype=np.uint32), int(r.idx)

    def rollout_host(self, iters, policy_seed=0):
        self.v.rollout(iters, policy_seed)

    def counters(self):
        c = self.v.counters()
        c["iters"] = c.pop("iter")
        return c

    def get_state(self, i):
        g, N = self.v.game(i), self.num_players
        return dict(
            cards=np.array([list(g.players_cards[p]) for p in range(N)], dtype=np.int8),
            masked=np.array([list(g.players_masked[p]) for p in range(N)], dtype=np.int8),
            draw=np.array(list(g.drawpile[: g.n_draw]), dtype=np.int8), n_draw=g.n_draw,
            disc=np.array(list(g.discard_pile[: g.n_disc]), dtype=np.int8), n_disc=g.n_disc,
            hand=g.hand_card, player=g.exp_player, phase=g.exp_phase, is_terminated=bool(g.is_terminated),
            done=bool(self.v.dones[i]), num_refunded=np.array(g.num_refunded[:N]),
            num_placed=np.array(g.num_placed[:N]), final_score=np.array(g.final_score[:N]),
            rewards=self.v.rewards[i])
