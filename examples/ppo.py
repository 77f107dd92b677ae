"""EXAMPLE, not part of the package (SURVEY C5 - RLlib training - is out of scope).  Learner side of the rollout hand-off: a minimal clipped-PPO update that consumes the columns of
``rollout.RolloutBuffer`` - what ``rlskyjo/models/train_model_simple_rllib.py:22-59`` has RLlib's PPO trainer do with the
sample batches of its rollout workers.  The trainer itself (Ray, tune, checkpoints) stays out of scope; this file shows
that the buffer is sufficient for a learner and closes the loop  collect (HIP kernels) -> update (torch autograd on
``ActionMaskModel``) -> re-pack the weights for the matrix cores -> collect.

Rewards in SkyJo arrive once, at the end of an episode, for every seat (skyjo_env.py:293-312); with RLlib's multi-agent
set-up every seat is its own agent, so the return of a step is the final reward of the seat that acted (gamma = 1 inside
an episode).  Steps of episodes that did not finish inside the buffer carry no return and are masked out.
"""
import torch

from skyjo_rl_amd.action_mask_model import FLOAT_MIN, FusedNet


@torch.no_grad()
def compute_returns(buf):
    """returns float32 [T, B]: final reward (skyjo_env.py:293-312) of the acting seat's episode, and mask bool [T, B]:
    the row is a transition (``buf.valid``) whose episode ended inside the buffer."""
    v = buf.views()
    T, B, N = buf.T, buf.B, buf.N
    agent = v.agent[:T].long()                       # the seat that acted at step t
    carry = torch.zeros((B, N), dtype=torch.float64, device=buf.actions.device)
    known = torch.zeros((B,), dtype=torch.bool, device=buf.actions.device)
    returns = torch.zeros((T, B), dtype=torch.float32, device=buf.actions.device)
    mask = torch.zeros((T, B), dtype=torch.bool, device=buf.actions.device)
    valid = buf.valid
    for t in range(T - 1, -1, -1):
        end = buf.episode_end[t].bool()
        carry = torch.where(end.unsqueeze(-1), buf.final_rewards[t], carry)
        known = known | end
        returns[t] = carry.gather(1, agent[t].unsqueeze(1)).squeeze(1).to(torch.float32)
        mask[t] = known & valid[t]
    return returns, mask


def ppo_update(model, buf, optimizer, epochs=2, minibatch=1 << 15, clip=0.3, vf_coef=1.0, seed=0):
    """Clipped-surrogate PPO epochs over the buffer (RLlib defaults: clip_param 0.3, vf_loss_coeff 1.0).  Returns the mean
    losses of the first and the last epoch."""
    v = buf.views()
    T = buf.T
    returns, mask = compute_returns(buf)
    idx = mask.reshape(-1).nonzero().squeeze(1)
    obs = v.observations[:T].reshape(-1, v.observations.shape[-1])
    am = v.action_mask[:T].reshape(-1, 26)
    act = buf.actions.reshape(-1).long()
    logp_old = buf.logp.reshape(-1)
    val_old = buf.values[:T].reshape(-1)
    ret = returns.reshape(-1)
    adv_all = ret - val_old
    mean, std = adv_all[idx].mean(), adv_all[idx].std().clamp_min(1e-6)
    gen = torch.Generator(device=idx.device).manual_seed(seed)
    stats = []
    for ep in range(epochs):
        perm = idx[torch.randperm(idx.numel(), device=idx.device, generator=gen)]
        tot = {"policy_loss": 0.0, "vf_loss": 0.0, "kl": 0.0, "n": 0}
        for k in range(0, perm.numel(), minibatch):
            j = perm[k:k + minibatch]
            x = obs[j].to(torch.float32)
            logits = model.policy(x) + torch.clamp(torch.log(am[j].to(torch.float32)), min=FLOAT_MIN)  # action_mask_model.py:70-71
            logp = torch.log_softmax(logits, -1).gather(1, act[j].unsqueeze(1)).squeeze(1)
            value = model.value(x).squeeze(-1)
            ratio = torch.exp(logp - logp_old[j])
            adv = (adv_all[j] - mean) / std
            pl = -torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv).mean()
            vl = ((value - ret[j]) ** 2).mean()
            loss = pl + vf_coef * vl
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            n = j.numel()
            tot["policy_loss"] += float(pl.detach()) * n
            tot["vf_loss"] += float(vl.detach()) * n
            tot["kl"] += float((logp_old[j] - logp).mean().detach()) * n
            tot["n"] += n
        stats.append({k: tot[k] / max(tot["n"], 1) for k in ("policy_loss", "vf_loss", "kl")})
    return {"first": stats[0], "last": stats[-1], "transitions": int(idx.numel())}


def repack(model, device=0):
    """The updated weights as MFMA fragments for the next rollout: (policy FusedNet, value FusedNet)."""
    return FusedNet(model.policy, device=device), FusedNet(model.value, device=device)
