"""SURVEY 8f.1, the hand-off the verdict of round 1 asked for: rollout buffer (HIP kernels) -> PPO update (torch) -> re-packed
weights -> rollout.  Checks the returns the learner derives from the buffer against the engine's own rewards, that an update
on the buffer fits the value branch (loss falls) without moving the policy far (clipped), and that the loop closes."""
import pytest

pytestmark = pytest.mark.gpu


def test_rollout_to_learner_and_back():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel
    from examples.ppo import compute_returns, ppo_update, repack
    from skyjo_rl_amd.rollout import RolloutBuffer, collect

    torch.manual_seed(0)
    B, N, T = 4096, 3, 320
    env = SkyjoVecEnv(B, num_players=N)
    env.seed(None, 9)
    env.reset()
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = repack(model)
    buf = RolloutBuffer(env, T)
    collect(env, pol, val, buf, seed=1, first_ticket=0)

    returns, mask = compute_returns(buf)
    v = buf.views()
    ends = buf.episode_end.bool()
    assert int(ends.sum()) > B  # every game finished at least one episode inside 320 steps ... most of them two
    # the return of the step that ended an episode is the final reward of the seat that made it (the finisher's draw)
    t, b = ends.nonzero()[:64].T
    seat = v.agent[t, b].long()
    want = buf.final_rewards[t, b, seat].float()
    assert torch.equal(returns[t, b], want)
    # inside an episode every step of a seat carries that seat's final reward
    tt, bb = int(t[0]), int(b[0])
    k = tt
    while k > 0 and not bool(ends[k - 1, bb]) and bool(buf.valid[k - 1, bb]):
        k -= 1
        s = int(v.agent[k, bb])
        assert float(returns[k, bb]) == float(buf.final_rewards[tt, bb, s].float())  # (returns are float32)
    assert bool((mask <= buf.valid).all()) and float(mask.float().mean()) > 0.5

    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    before = [p.detach().clone() for p in model.policy.parameters()]
    out = ppo_update(model, buf, opt, epochs=3)
    assert out["transitions"] == int(mask.sum())
    assert all(torch.isfinite(torch.tensor([out[k][j] for k in ("first", "last") for j in ("policy_loss", "vf_loss", "kl")])))
    assert out["last"]["vf_loss"] < out["first"]["vf_loss"]          # the value branch fits the returns
    assert abs(out["last"]["kl"]) < 0.05                             # the clipped policy stays close
    assert any(float((p.detach() - q).abs().max()) > 0 for p, q in zip(model.policy.parameters(), before))

    pol.close(), val.close()
    pol, val = repack(model)                                          # updated weights back onto the matrix cores
    collect(env, pol, val, buf, seed=1, first_ticket=T)
    assert env.counters()["illegal"] == 0 and int(buf.episode_end.sum()) > 0
    # the re-packed net is the updated torch module (float32-grade tolerance of tests/test_gpu_policy_net.py)
    with torch.no_grad():
        ref = model.policy(buf.views().observations[5].to(torch.float32))
    assert float((pol(buf.records[5]) - ref).abs().max()) < 1e-4
    pol.close(), val.close(), env.close()
