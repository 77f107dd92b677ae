"""Config 5 plumbing: 4-player games stepped by the action-mask model on the GPU, records consumed zero-copy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_action_mask_model_drives_env_without_illegal_moves():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, sample_actions

    torch.manual_seed(0)
    B, N = 8192, 4
    env = SkyjoVecEnv(B, num_players=N)  # DEFAULT_CONFIG: indirect observation, D = 31
    env.seed(None, 3)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    rec = env.reset()
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(400):
        v = env.split(rec)
        obs = {"observations": v.observations, "action_mask": v.action_mask}
        assert v.observations.data_ptr() == rec.data_ptr()  # a view into the record tensor, no copy
        actions = sample_actions(model, obs, gen)
        # the masked logits never select a masked-out action (action_mask_model.py:70-71)
        assert bool(v.action_mask.gather(1, actions.long().unsqueeze(1)).squeeze(1).eq(1).all())
        rec = env.step(actions)
    c = env.counters()
    assert c["illegal"] == 0 and c["episodes"] > B and c["steps"] + c["resets"] == 400 * B + B  # (+B: the explicit reset above)
    assert model.value_function().shape == (B,)
    # masking switched off reproduces the reference's warning scenario: illegal moves end games with -1
    model.no_masking = True
    for t in range(4):
        v = env.split(rec)
        rec = env.step(sample_actions(model, {"observations": v.observations, "action_mask": v.action_mask}, gen))
    assert env.counters()["illegal"] > 0
    rew = env.rewards_tensor()
    st = env.split(rec).status
    bad = (st == 1).nonzero().flatten()
    assert len(bad) > 0 and bool((rew[bad].min(dim=1).values == -1.0).all())
    env.close()
