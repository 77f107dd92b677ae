"""PPO-style rollout collection for config 5, all on the GPU and all behind ONE call of the C ABI
(``skyjo_vec_model_rollout``): per lockstep iteration one launch evaluates the policy branch with its masked categorical draw
and the value branch on the matrix cores, and the step kernel writes the next records straight into the buffer and - the lane
that ends an episode - the episode-end flag and the final rewards: two launches per iteration, no torch kernel, no host code
between them.  ``collect_stepwise`` is the same loop made one launch at a time from Python (``FusedNet.act(value_net=...)``,
``skyjo_vec_step_collect``): same bits, used by the tests.  What a learner needs per step of the acting seat
(``rlskyjo/models/train_model_simple_rllib.py:22-59`` has RLlib collect the same columns): observation / action mask (inside the
records), action, log-probability, value estimate, the acting agent, done flags and - at episode ends - the final rewards of
skyjo_env.py:293-312 for every seat.  ``RolloutBuffer.valid`` tells a learner which rows are transitions at all.
"""
import ctypes as C

import torch

from . import _lib


class RolloutBuffer:
    """Columns of T lockstep iterations x B games, preallocated once and refilled by ``collect``."""

    def __init__(self, env, T):
        dev = torch.device("cuda", env.device_index)
        B, N = env.num_envs, env.num_players
        self.T, self.B, self.N = T, B, N
        self._env = env
        # records[t] = what the actor of step t saw; [T] = bootstrap.  With the engine's layout 'tile-planar-all' the buffer is tile-planar
        # ([T + 1, tiles, P, 64, 16]: the step kernel writes it from registers, the nets read it in place) and ``views`` copies
        self.planar = env.record_layout == "tile-planar-all"
        self.records = env.new_planar_records(T + 1) if self.planar else env.new_records(T + 1)
        self.actions = torch.empty((T, B), dtype=torch.int32, device=dev)
        self.logp = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.values = torch.empty((T + 1, B, 1), dtype=torch.float32, device=dev)
        self.final_rewards = torch.zeros((T, B, N), dtype=torch.float64, device=dev)  # non-zero rows where episode_end[t]
        self.episode_end = torch.zeros((T, B), dtype=torch.uint8, device=dev)          # 1 where the step ended the episode

    def views(self, env=None):
        """Zero-copy column views of the stored records: observations int8 [T+1, B, D], action_mask int8 [T+1, B, 26],
        agent / phase / done / status uint8 [T+1, B]."""
        e = env or self._env
        return e.split(e.rows_from_planar(self.records) if self.planar else self.records)

    @property
    def valid(self):
        """bool [T, B]: row t of game b is a transition.  Where ``records[t]`` already shows ``done`` the step only re-deals
        the game (auto-reset: the action, log-probability and value stored for it were ignored) - mask those rows out."""
        return self.views().done[: self.T] == 0


def _first(env, buf, first_records):
    if first_records is None:
        env.observe(out=buf.records[0])
    else:
        buf.records[0].copy_(first_records)


@torch.no_grad()
def collect(env, policy, value, buf, seed=0, first_ticket=0, first_records=None, no_masking=False):
    """Fill ``buf`` with T steps of the current policy in one native call.  ``policy`` / ``value``: ``FusedNet`` of the
    model's two branches.  ``first_records``: the records the rollout starts from (default: ``env.observe()``)."""
    L = _lib.load()
    vp = lambda t: t.data_ptr()
    _first(env, buf, first_records)
    b = _lib.RolloutBuffers(vp(buf.records), vp(buf.actions), vp(buf.logp), vp(buf.values), vp(buf.final_rewards), vp(buf.episode_end))
    _lib.check(L.skyjo_vec_model_rollout(env._h, policy._h, value._h, buf.T, int(seed), int(first_ticket), 1 if no_masking else 0,
                                         C.byref(b), env._stream()))
    return buf


@torch.no_grad()
def collect_stepwise(env, policy, value, buf, seed=0, first_ticket=0, first_records=None):
    """The same rollout, one launch at a time from Python (the net, then ``skyjo_vec_step_collect``): bit-identical columns."""
    L = _lib.load()
    vp = lambda t: C.c_void_p(t.data_ptr())
    _first(env, buf, first_records)
    for t in range(buf.T):
        policy.act(env, buf.records[t], seed=seed, ticket=first_ticket + t, actions=buf.actions[t], logp=buf.logp[t],
                   value_net=value, values=buf.values[t], planar=buf.planar)
        # a game that ends in this step has its flag and its final rewards written by the step kernel itself
        _lib.check(L.skyjo_vec_step_collect(env._h, vp(buf.actions[t]), vp(buf.records[t + 1]), vp(buf.final_rewards[t]),
                                            vp(buf.episode_end[t]), env._stream()))
    value(buf.records[buf.T], out=buf.values[buf.T], planar=buf.planar)
    return buf
