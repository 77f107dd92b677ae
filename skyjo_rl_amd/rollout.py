"""PPO-style rollout collection for config 5, all on the GPU: per lockstep iteration the policy and value
nets run as MFMA kernels over the engine's records (``FusedNet``), the masked categorical draw is the fused HIP pass
(``SkyjoVecEnv.sample_actions``) and the env step writes the next records straight into the buffer - no tensor of the
rollout is ever touched by a torch kernel or the host.  What a learner needs per step of the acting seat
(``rlskyjo/models/train_model_simple_rllib.py:22-59`` has RLlib collect the same columns): observation / action mask
(inside the records), action, log-probability, value estimate, the acting agent, done flags and - at episode ends -
the final rewards of skyjo_env.py:293-312 for every seat.
"""
import torch


class RolloutBuffer:
    """Columns of T lockstep iterations x B games, preallocated once and refilled by ``collect``."""

    def __init__(self, env, T):
        dev = torch.device("cuda", env.device_index)
        B, N = env.num_envs, env.num_players
        self.T, self.B, self.N = T, B, N
        self.records = env.new_records(T + 1)                    # records[t] = what the actor of step t saw; [T] = bootstrap
        self.actions = torch.empty((T, B), dtype=torch.int32, device=dev)
        self.logp = torch.empty((T, B), dtype=torch.float32, device=dev)
        self.values = torch.empty((T + 1, B, 1), dtype=torch.float32, device=dev)
        self.final_rewards = torch.zeros((T, B, N), dtype=torch.float64, device=dev)  # non-zero rows where episode_end[t]
        self.episode_end = torch.zeros((T, B), dtype=torch.bool, device=dev)

    def views(self, env):
        """Zero-copy column views of the stored records: observations int8 [T+1, B, D], action_mask int8 [T+1, B, 26],
        agent / phase / done / status uint8 [T+1, B]."""
        return env.split(self.records)


@torch.no_grad()
def collect(env, policy, value, buf, seed=0, first_ticket=0, first_records=None):
    """Fill ``buf`` with T steps of the current policy.  ``policy`` / ``value``: ``FusedNet`` of the model's two
    branches.  ``first_records``: the records the rollout starts from (default: ``env.observe()``).  Returns ``buf``."""
    T = buf.T
    if first_records is None:
        env.observe(out=buf.records[0])
    else:
        buf.records[0].copy_(first_records)
    for t in range(T):
        rec = buf.records[t]
        policy.act(env, rec, seed=seed, ticket=first_ticket + t, actions=buf.actions[t], logp=buf.logp[t])
        value(rec, out=buf.values[t])
        env.step(buf.actions[t], out=buf.records[t + 1])
        v = env.split(buf.records[t + 1])
        # a game that has just ended shows done = 1 in the record written by this step; its rewards stay valid until the
        # reset that the next step performs
        buf.episode_end[t] = v.done.bool() & (v.status != 3)
        buf.final_rewards[t] = env.rewards_tensor() * buf.episode_end[t].unsqueeze(-1)
    value(buf.records[T], out=buf.values[T])
    return buf
