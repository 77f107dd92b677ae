"""the on-device rollout collector (skyjo_rl_amd/rollout.py) - columns are consistent with each other and
with the engine: recorded actions are legal under the recorded masks, recorded log-probabilities equal the masked
log-softmax of the (recomputed) logits at the recorded action within 1e-5, episode ends carry the rewards of
skyjo_env.py:293-312 (per game they sum to N * mean_reward + refund bonus)."""
import pytest

pytestmark = pytest.mark.gpu


def test_rollout_buffer_columns_are_consistent():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
    from skyjo_rl_amd.rollout import RolloutBuffer, collect

    torch.manual_seed(0)
    B, N, T = 2048, 4, 160
    env = SkyjoVecEnv(B, num_players=N, reward_refunded=0.0)
    env.seed(None, 21)
    env.reset()
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy), FusedNet(model.value)
    buf = RolloutBuffer(env, T)
    for rnd in range(3):
        collect(env, pol, val, buf, seed=5, first_ticket=rnd * T)
        v = buf.views(env)
        a = buf.actions.long()
        assert bool(v.action_mask[:T].gather(2, a.unsqueeze(-1)).eq(1).all())      # legal under the stored masks
        assert bool((v.agent[:T] < N).all())
        # stored log-probabilities against a recomputation from the stored records
        t = T // 2
        logits = pol(buf.records[t])
        mask = v.action_mask[t].float()
        ref = torch.log_softmax(logits + torch.clamp(torch.log(mask), min=torch.finfo(torch.float32).min), -1)
        assert float((ref.gather(1, a[t].unsqueeze(1)).squeeze(1) - buf.logp[t]).abs().max()) < 1e-5
        assert float((val(buf.records[t]) - buf.values[t]).abs().max()) == 0.0
        ends = buf.episode_end.bool()
        # rows whose record already showed done are re-deals, not transitions: exactly the steps after an episode end
        valid = buf.valid
        assert bool((~valid[1:] == ends[:-1]).all())
        assert bool((v.status[1:T + 1][~valid] == 3).all())  # the step of an invalid row only re-dealt the game
        if rnd == 2:
            assert int(ends.sum()) > 0
        rw = buf.final_rewards[ends]                                               # [episodes, N]
        assert bool(((rw.sum(-1) - N * 1.0).abs() < 1e-9).all())                    # skyjo_env.py:307-312
        assert bool((buf.final_rewards[~ends] == 0).all())
    assert env.counters()["illegal"] == 0
    env.close()


def test_native_rollout_call_equals_the_stepwise_loop_and_the_separate_episode_end_kernel():
    """skyjo_vec_model_rollout (T iterations behind one call, two launches each) against the same loop made one launch at a
    time from Python, and the step kernel's episode-end columns against skyjo_vec_episode_ends run on the stored records:
    every column bit for bit.  Two engines seeded alike play the two forms."""
    import ctypes as C

    import torch

    from skyjo_rl_amd import SkyjoVecEnv, _lib
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
    from skyjo_rl_amd.rollout import RolloutBuffer, collect, collect_stepwise

    torch.manual_seed(4)
    B, N, T = 3000, 4, 150
    model = ActionMaskModel(obs_dim=31).cuda()
    pol, val = FusedNet(model.policy), FusedNet(model.value)
    cols = []
    for form in (collect, collect_stepwise):
        env = SkyjoVecEnv(B, num_players=N)
        env.seed(None, 33)
        buf = RolloutBuffer(env, T)
        for rnd in range(2):
            form(env, pol, val, buf, seed=8, first_ticket=rnd * T, first_records=None if rnd == 0 else buf.records[T].clone())
        cols.append([x.clone() for x in (buf.records, buf.actions, buf.logp, buf.values, buf.final_rewards, buf.episode_end)])
        if form is collect:
            L = _lib.load()
            vp = lambda t: C.c_void_p(t.data_ptr())
            fr, ee = torch.empty_like(buf.final_rewards[0]), torch.empty_like(buf.episode_end[0])
            hits = 0
            for t in range(T - 1):  # (the rewards of an ended game stay in place until the next step re-deals it)
                if not bool(buf.episode_end[t].any()):
                    continue
                # re-derive from the records of step t; the engine's reward array still holds step T's state, so only the
                # flags can be compared for earlier steps
                _lib.check(L.skyjo_vec_episode_ends(env._h, vp(buf.records[t + 1]), vp(fr), vp(ee), env._stream()))
                assert torch.equal(ee, buf.episode_end[t])
                hits += 1
            assert hits > 0
            _lib.check(L.skyjo_vec_episode_ends(env._h, vp(buf.records[T]), vp(fr), vp(ee), env._stream()))
            assert torch.equal(ee, buf.episode_end[T - 1]) and torch.equal(fr, buf.final_rewards[T - 1])
        assert env.counters()["illegal"] == 0 and int(buf.episode_end.sum()) > 0
        env.close()
    for a, b in zip(*cols):
        assert torch.equal(a, b)

