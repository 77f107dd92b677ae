"""SURVEY 8f.1 (config 5 caller): the fused masking + categorical draw (skyjo_vec_sample_actions) against the plain
torch float32 statement of rlskyjo/models/action_mask_model.py:58-74.  Floating point: probabilities must agree with
torch.softmax within 1e-6 absolute, log-probabilities within 1e-5; the drawn action must be the inverse-CDF pick of
torch's own probabilities for the kernel's uniform, except where that uniform lies within 1e-5 of a CDF step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLOAT_MIN = np.finfo(np.float32).min


@pytest.mark.parametrize("N,ind,B", [(4, True, 8192), (3, False, 1000), (2, True, 257)])
def test_fused_sampler_matches_torch_reference(N, ind, B):
    import torch

    from skyjo_rl_amd import SkyjoVecEnv

    env = SkyjoVecEnv(B, num_players=N, observe_other_player_indirect=ind)
    env.seed(None, 11)
    rec = env.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    for step in range(6):  # both phases: draw (2 legal actions) and place (up to 24)
        v = env.split(rec)
        logits = (torch.randn((B, 26), generator=g, device="cuda") * 3.0).contiguous()
        logp = torch.empty(B, dtype=torch.float32, device="cuda")
        uni = torch.empty(B, dtype=torch.float32, device="cuda")
        act = env.sample_actions(logits, rec, seed=77, ticket=step, logp=logp, uniform=uni)
        act2 = env.sample_actions(logits, rec, seed=77, ticket=step)
        assert torch.equal(act, act2)  # same (seed, ticket) -> same draws
        mask = v.action_mask.to(torch.float32)
        masked = logits + torch.clamp(torch.log(mask), min=float(FLOAT_MIN))  # action_mask_model.py:70-71
        p = torch.softmax(masked, dim=-1)
        a = act.long()
        assert bool((uni >= 0).all()) and bool((uni < 1).all())
        assert bool(v.action_mask.gather(1, a[:, None]).eq(1).all())  # never a masked-out action
        ref_logp = torch.log_softmax(masked, dim=-1).gather(1, a[:, None]).squeeze(1)
        assert float((logp - ref_logp).abs().max()) < 1e-5
        cdf = torch.cumsum(p.double(), dim=-1)
        lo = torch.where(a > 0, cdf.gather(1, (a - 1).clamp(min=0)[:, None]).squeeze(1), torch.zeros_like(cdf[:, 0]))
        hi = cdf.gather(1, a[:, None]).squeeze(1)
        u = uni.double()
        assert bool(((u >= lo - 1e-5) & (u <= hi + 1e-5)).all())  # inverse CDF of torch's probabilities
        # the implied probabilities themselves: exp(logp) of the drawn action vs torch.softmax, 1e-6 absolute
        assert float((torch.exp(logp) - p.gather(1, a[:, None]).squeeze(1)).abs().max()) < 1e-6
        rec = env.step(act)
    assert env.counters()["illegal"] == 0
    env.close()


def test_fused_sampler_distribution_and_no_masking():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv

    B = 65536
    env = SkyjoVecEnv(B, num_players=3)
    env.seed(None, 3)
    rec = env.reset()
    rec = env.step(torch.full((B,), 24, dtype=torch.int32, device="cuda"))  # everybody draws: place phase next
    row = torch.linspace(-2, 2, 26, device="cuda")
    logits = row.repeat(B, 1).contiguous()
    act = env.sample_actions(logits, rec, seed=1, ticket=0, no_masking=True)
    freq = torch.bincount(act.long(), minlength=26).double() / B
    p = torch.softmax(row.double(), dim=0)
    # 65 536 draws: 5 sigma of a binomial frequency is below 0.004 for every p <= 0.15
    assert float((freq - p).abs().max()) < 0.004
    masked = env.sample_actions(logits, rec, seed=1, ticket=0)
    assert bool(env.split(rec).action_mask.gather(1, masked.long()[:, None]).eq(1).all())
    env.close()


def test_a_masked_action_is_never_drawn_in_two_hundred_million_draws():
    """The draw sums its exponentials in blocks of four (csrc/skyjo_draw.h) and a block's CDF starts from a total that was rounded on
    another path than the running sum of the block before - one ulp apart now and then.  Without the "non-zero probability"
    condition a masked action at the head of a block was drawn about once in 10^7 draws (found by the every-record test of config
    5).  65 536 games in the place phase (hidden cards revealed as the games go on: masked actions in every block), fixed logits, 3 000
    tickets: 2 x 10^8 draws, none of them masked out."""
    import torch

    from skyjo_rl_amd import SkyjoVecEnv

    B = 65536
    env = SkyjoVecEnv(B, num_players=4)
    env.seed(None, 17)
    rec = env.reset()
    g = torch.Generator(device="cuda").manual_seed(9)
    logits = (torch.randn((B, 26), generator=g, device="cuda") * 2.0).contiguous()
    for t in range(24):  # a few rounds in, so that the masks differ from game to game
        rec = env.step(env.sample_actions(logits, rec, seed=2, ticket=t))
    rec = env.step(torch.full((B,), 24, dtype=torch.int32, device="cuda"))  # (those in the draw phase draw: place phase next)
    mask = env.split(rec).action_mask
    assert int((mask.sum(1) > 2).sum()) > B // 2  # mostly place-phase masks
    act = torch.empty(B, dtype=torch.int32, device="cuda")
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for ticket in range(3000):
        env.sample_actions(logits, rec, seed=5, ticket=ticket, actions=act)
        bad += (mask.gather(1, act.long()[:, None]).squeeze(1) == 0).sum()
    assert int(bad) == 0
    # the two-lane form of the same draw, in the policy net's own launch (65 million more)
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(4)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    with torch.no_grad():
        model.policy[-1].weight.mul_(20.0)
    pol = FusedNet(model.policy, precision="bf16")
    for ticket in range(1000):
        pol.act(env, rec, seed=6, ticket=ticket, actions=act)
        bad += (mask.gather(1, act.long()[:, None]).squeeze(1) == 0).sum()
    assert int(bad) == 0
    pol.close()
    env.close()


def test_model_loop_with_fused_sampler():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, sample_actions_fused

    torch.manual_seed(0)
    B = 4096
    env = SkyjoVecEnv(B, num_players=4)
    env.seed(None, 3)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    rec = env.reset()
    for t in range(300):
        rec = env.step(sample_actions_fused(model, env, rec, seed=9, ticket=t))
    c = env.counters()
    assert c["illegal"] == 0 and c["episodes"] > 0
    env.close()
