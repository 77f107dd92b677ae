"""CPU check of the action-mask model's masking rule (rlskyjo/models/action_mask_model.py:58-74)."""
import torch

from skyjo_rl_amd.action_mask_model import FLOAT_MIN, ActionMaskModel, sample_actions


def test_masking_rule_and_sampling():
    torch.manual_seed(0)
    m = ActionMaskModel(obs_dim=31)
    obs = torch.randint(-2, 13, (64, 31), dtype=torch.int8)
    mask = torch.zeros((64, 26), dtype=torch.int8)
    mask[:, 24:] = 1
    mask[::2, :24] = (torch.rand(32, 24) < 0.5).to(torch.int8)
    out = m({"observations": obs, "action_mask": mask})
    raw = m.policy(obs.float())
    # fp32 tolerance 0: adding 0 keeps the logit, adding FLOAT_MIN saturates to the most negative float
    assert torch.equal(out[mask.bool()], raw[mask.bool()])
    assert bool((out[~mask.bool()] <= FLOAT_MIN / 2).all())
    a = sample_actions(m, {"observations": obs, "action_mask": mask})
    assert a.dtype == torch.int32 and bool(mask.gather(1, a.long().unsqueeze(1)).eq(1).all())
    assert m.value_function().shape == (64,)
    m.no_masking = True
    assert torch.equal(m({"observations": obs, "action_mask": mask}), raw)
