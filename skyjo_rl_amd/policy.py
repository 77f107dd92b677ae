"""Random admissible policy: the caller of the env hot path (rlskyjo/models/random_admissible_policy.py:6-28).

``policy_ra`` keeps the reference's signature and RNG consumption (one ``choice`` with
``p = mask / sum(mask)`` on ``np.random`` or on the Generator passed in), so seeded runs that
drive this package's AEC view reproduce the reference action for action.
``random_admissible_actions`` is the batched torch equivalent for device-resident masks; the
fully fused variant lives in the rollout kernel (skyjo_device.h: policy_pick).
"""
import numpy as np


def policy_ra(observation, action_mask, rng=None):
    source = np.random if rng is None else rng
    mask = np.asarray(action_mask)
    return source.choice(np.arange(len(mask)), p=mask / np.sum(mask))


def random_admissible_actions(action_mask, generator=None):
    """Uniform choice among legal actions for a batch of masks [B, 26] (torch, any device) -> int32 [B]."""
    import torch

    probs = action_mask.to(torch.float32)
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1).to(torch.int32)
