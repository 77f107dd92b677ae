"""Action-mask policy model: the second caller of the env hot path (config 5 of BASELINE.json).

Restates ``TorchActionMaskModel`` (rlskyjo/models/action_mask_model.py:13-77) without Ray: RLlib's
default fully connected net (two tanh layers of 256 units, separate value branch) on
``obs["observations"]``, and ``logits + clamp(log(action_mask), min=FLOAT_MIN)`` as in :58-74.  It
consumes the zero-copy views of the engine's record tensor directly on the GPU (``SkyjoVecEnv.split``),
so a PPO-style rollout never leaves the device.  This is caller code, not part of the accelerated
path; MFMA work (the three small GEMMs) is left to PyTorch-ROCm / hipBLASLt.  The masking + categorical
draw has a fused HIP form behind the C ABI (``skyjo_vec_sample_actions``, ``sample_actions_fused`` below);
``forward`` / ``sample_actions`` are the plain-torch statement of the same arithmetic and its test reference.
"""
import torch
from torch import nn

FLOAT_MIN = torch.finfo(torch.float32).min  # ray.rllib.utils.torch_utils.FLOAT_MIN


class ActionMaskModel(nn.Module):
    def __init__(self, obs_dim=31, num_outputs=26, hiddens=(256, 256), no_masking=False):
        super().__init__()
        self.no_masking = no_masking  # action_mask_model.py:53-56
        layers, d = [], obs_dim
        for h in hiddens:
            layers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        self.policy = nn.Sequential(*layers, nn.Linear(d, num_outputs))
        vlayers, d = [], obs_dim
        for h in hiddens:
            vlayers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        self.value = nn.Sequential(*vlayers, nn.Linear(d, 1))
        self._last_obs = None

    def forward(self, obs):
        """obs: {"observations": int8/float [B, D], "action_mask": int8/float [B, 26]} -> masked logits [B, 26]."""
        x = obs["observations"].to(torch.float32)
        self._last_obs = x
        logits = self.policy(x)
        if self.no_masking:
            return logits
        inf_mask = torch.clamp(torch.log(obs["action_mask"].to(torch.float32)), min=FLOAT_MIN)
        return logits + inf_mask

    def value_function(self):
        return self.value(self._last_obs).squeeze(-1)


@torch.no_grad()
def sample_actions(model, obs, generator=None):
    """Categorical sample from the masked logits -> int32 actions for SkyjoVecEnv.step."""
    probs = torch.softmax(model(obs), dim=-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1).to(torch.int32)


@torch.no_grad()
def sample_actions_fused(model, env, records, seed=0, ticket=0, logp=None):
    """The same draw with the masking, softmax and sampling fused into one HIP pass over the engine's records
    (``SkyjoVecEnv.sample_actions``): only the policy net's three GEMMs run in torch."""
    v = env.split(records)
    x = v.observations.to(torch.float32)
    model._last_obs = x
    logits = model.policy(x)
    return env.sample_actions(logits, records, seed=seed, ticket=ticket, no_masking=model.no_masking, logp=logp)
