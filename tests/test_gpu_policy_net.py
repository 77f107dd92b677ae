"""the policy net of config 5 on the matrix cores (skyjo_vec_mlp_forward, csrc/skyjo_policy.hip)
against torch.  The reference's TorchFC is float32 (rlskyjo/models/action_mask_model.py:43-49, 58-74); north_star
states no tolerance, the ones used here are written out:

  precision "fp32" (the default of FusedNet; every operand a bf16 pair, three MFMAs per product, float32 accumulation):
  * against the plain float32 module: max |diff| of logits / values <= 1e-4, mean <= 2e-5, and
    mean KL(softmax fp32 || softmax kernel) <= 1e-7 - at 65 536 x 4 as well (test_config5_full_size_...).
  precision "bf16" (single bf16 weights and inter-layer activations - the fast mode):
  * against a torch float32 emulation of exactly that arithmetic (_emulate: bf16-rounded weights - the hidden layers' as the
    kernel packs them, times 2 / ln 2 -, r = (1 - tanh) / 2 rounded to bf16 after each activation): max |diff| of the outputs < 2e-2 and mean |diff| < 2e-3 (what is left is the fast tanh, the order of
    the float32 sums and the bf16 roundings that flip because of them);
  * against the plain float32 module: max |diff| < 8e-2, mean < 1e-2, mean KL < 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


SCALE = 2.8853900817779268  # 2 / ln 2 (SKP_SCALE, csrc/skyjo_policy.h)


def _emulate(seq, x):
    """The kernel's bf16 arithmetic in float32 torch: the two hidden layers are packed times 2 / ln 2 BEFORE the rounding to
    bf16 (so that their accumulators are the exponent of tanh(x) = 1 - 2 / (2^(x 2 / ln 2) + 1) as they stand), layer 1's bias
    rides in the product (bf16 as well), layer 2's is the accumulator's initial value (float32); what is rounded to bf16 after
    the activation is r = 1 / (2^y + 1) = (1 - tanh) / 2, which the next layer takes with - 2 W as weights and b + W 1 as bias
    (round 6: the activation is v_exp, v_add, v_rcp); the output layer is bf16 weights + float32 bias."""
    import torch
    from torch import nn

    h, from_r = x, False
    lins = [m for m in seq if isinstance(m, nn.Linear)]
    for i, lin in enumerate(lins):
        sc = SCALE if i < 2 else 1.0
        w = (lin.weight.detach().float() * sc).bfloat16().float()
        b = lin.bias.detach().float() * sc
        if i == 0:
            b = b.bfloat16().float()
        if from_r:
            h = h @ (-2.0 * w).t() + (b.double() + w.double().sum(1)).float()
        else:
            h = h @ w.t() + b
        if i < 2:
            h = (1.0 / (torch.exp2(h) + 1.0)).bfloat16().float()
            from_r = True
    return h


TOL = {"fp32": dict(max=1e-4, mean=2e-5, kl=1e-7), "bf16": dict(max=8e-2, mean=1e-2, kl=1e-3)}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("N,B,branch", [(4, 8192, "policy"), (3, 1000, "policy"), (2, 33, "value"), (4, 65536, "value"),
                                        (4, 65536, "policy")])
def test_policy_net_matches_torch(N, B, branch, precision):
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(1)
    env = SkyjoVecEnv(B, num_players=N)
    env.seed(None, 5)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    with torch.no_grad():  # weights well away from zero so that every hidden unit matters
        for p in model.parameters():
            p.mul_(1.5)
    seq = model.policy if branch == "policy" else model.value
    net = FusedNet(seq, precision=precision)
    rec = env.reset()
    for t in range(5):
        rec = env.step(env.sample_actions(torch.zeros((B, 26), device="cuda"), rec, seed=1, ticket=t))
    x = env.split(rec).observations.to(torch.float32)
    got = net(rec)
    with torch.no_grad():
        emu = _emulate(seq, x)
        ref = seq(x)
    assert got.shape == ref.shape
    tol = TOL[precision]
    if precision == "bf16":
        d = (got - emu).abs()
        assert float(d.max()) < 2e-2 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))
    d32 = (got - ref).abs()
    assert float(d32.max()) <= tol["max"] and float(d32.mean()) <= tol["mean"], (precision, float(d32.max()), float(d32.mean()))
    if branch == "policy":
        ref64, got64 = ref.double(), got.double()  # (a KL of 1e-7 is below float32's resolution of the log-softmax)
        kl = (torch.softmax(ref64, -1) * (torch.log_softmax(ref64, -1) - torch.log_softmax(got64, -1))).sum(-1)
        assert float(kl.mean()) <= tol["kl"], float(kl.mean())
    net.close()
    env.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_policy_and_value_net_in_one_launch_over_many_ragged_batches(precision):
    """A launch is ONE round of workgroups: a workgroup keeps the 256 x 256 layer in LDS and walks through several batches of 256
    games (csrc/skyjo_policy.hip, `passes`).  140 100 two-player games are 548 batches of both nets over 256 compute units - five
    per workgroup, the last workgroup's fifth beyond the last game, the last batch partly empty - against the float32 module."""
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(3)
    B = 140100
    env = SkyjoVecEnv(B, num_players=2)
    env.seed(None, 11)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy, precision=precision), FusedNet(model.value, precision=precision)
    rec = env.reset()
    for t in range(3):
        rec = env.step(env.sample_actions(torch.zeros((B, 26), device="cuda"), rec, seed=2, ticket=t))
    x = env.split(rec).observations.to(torch.float32)
    logits = torch.empty((B, 26), device="cuda")
    values = torch.empty((B, 1), device="cuda")
    act = pol.act(env, rec, seed=5, ticket=0, logits=logits, value_net=val, values=values)
    with torch.no_grad():
        ref, vref = model.policy(x), model.value(x)
    tol = TOL[precision]
    assert float((logits - ref).abs().max()) <= tol["max"] and float((values - vref).abs().max()) <= tol["max"]
    assert torch.equal(pol(rec), logits) and torch.equal(val(rec), values)  # the single-net launches (other `passes`) give the same bits
    assert bool(env.split(rec).action_mask.gather(1, act.long().unsqueeze(1)).squeeze(1).eq(1).all())
    pol.close(), val.close(), env.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("obs_dim,out_dim,rec_bytes,n", [(17, 5, 48, 777), (1, 32, 32, 64), (31, 1, 112, 4100)])
def test_net_on_caller_records_of_other_shapes(obs_dim, out_dim, rec_bytes, n, precision):
    """skyjo_vec_mlp_forward takes any records of record_bytes >= 32 (a multiple of 16) whose first obs_dim <= 31 bytes are int8
    features: the bytes from obs_dim on - here random - must not enter (the kernel clears them word-wise and puts the constant 1 that
    carries layer 1's bias into byte 31)."""
    import torch
    from torch import nn

    from skyjo_rl_amd.action_mask_model import FusedNet

    torch.manual_seed(obs_dim)
    seq = nn.Sequential(nn.Linear(obs_dim, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh(), nn.Linear(256, out_dim)).cuda()
    net = FusedNet(seq, precision=precision)
    rec = torch.randint(0, 256, (n, rec_bytes), dtype=torch.uint8, device="cuda")
    x = rec.view(torch.int8)[:, :obs_dim].to(torch.float32) * 0.25  # (smaller features: the net is not saturated)
    rec8 = rec.clone()
    rec8.view(torch.int8)[:, :obs_dim] = (x * 1.0).to(torch.int8)  # features in [-32, 31]
    x = rec8.view(torch.int8)[:, :obs_dim].to(torch.float32)
    got = net(rec8)
    with torch.no_grad():
        ref = seq(x)
    tol = TOL[precision]
    d = (got - ref).abs()
    assert got.shape == (n, out_dim) and float(d.max()) <= 4 * tol["max"] and float(d.mean()) <= 4 * tol["mean"], (float(d.max()), float(d.mean()))
    net.close()


def test_fused_policy_loop_plays_legal_games():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(0)
    B = 4096
    env = SkyjoVecEnv(B, num_players=4)
    env.seed(None, 3)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy), FusedNet(model.value)
    rec = env.reset()
    for t in range(300):
        logits = pol(rec)
        rec = env.step(env.sample_actions(logits, rec, seed=4, ticket=t))
    assert val(rec).shape == (B, 1)
    c = env.counters()
    assert c["illegal"] == 0 and c["episodes"] > 0
    env.close()


@pytest.mark.parametrize("precision,N,no_masking", [("bf16", 4, False), ("fp32", 3, False), ("bf16", 2, True)])
def test_pair_draw_equals_one_lane_draw_on_sharp_logits(precision, N, no_masking):
    """The draw made in the net's launch (two lanes per game, csrc/skyjo_draw.h: sk_draw_action_pair) against skyjo_vec_sample_actions
    on the logits the same launch wrote (one lane per game): same actions, same log-probabilities, bit for bit - on logits spread over
    +- 40 (last layer times 60: most exponentials underflow to 0, the fall-back to the last possible action is taken, the CDF has
    long flat stretches), over 20 011 games, both phases, with and without the mask.  Every drawn action is legal under the mask."""
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(5)
    B = 20011
    env = SkyjoVecEnv(B, num_players=N)
    env.seed(None, 21)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    with torch.no_grad():
        model.policy[-1].weight.mul_(60.0)
    pol = FusedNet(model.policy, precision=precision)
    rec = env.reset()
    seen = torch.zeros(26, dtype=torch.int64, device="cuda")
    for t in range(30):
        lg = torch.empty((B, 26), device="cuda")
        lp_a = torch.empty(B, device="cuda")
        a = pol.act(env, rec, seed=13, ticket=t, no_masking=no_masking, logp=lp_a, logits=lg)
        if t == 0:
            assert float(lg.abs().max()) > 20.0
        lp_b = torch.empty(B, device="cuda")
        a_ref = env.sample_actions(lg, rec, seed=13, ticket=t, no_masking=no_masking, logp=lp_b)
        assert torch.equal(a, a_ref)
        assert torch.equal(lp_a, lp_b)
        mask = env.split(rec).action_mask
        legal = mask.gather(1, a.long()[:, None]).squeeze(1).eq(1)
        if not no_masking:
            assert bool(legal.all())
        seen += torch.bincount(a.long(), minlength=26)
        # (without the mask an illegal pick is possible: play the masked draw on, so that both phases keep coming)
        rec = env.step(a if not no_masking else env.sample_actions(lg, rec, seed=13, ticket=1000 + t))
    assert int((seen > 0).sum()) >= 20  # draws all over the action range, both halves of the lane pair
    env.close()


def test_act_equals_forward_then_sample_bit_for_bit():
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet

    torch.manual_seed(2)
    B = 5000  # (not a multiple of 32: the last wavefront is partly empty)
    env = SkyjoVecEnv(B, num_players=3)
    env.seed(None, 8)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol = FusedNet(model.policy)
    rec = env.reset()
    for t in range(40):
        logits = pol(rec)
        lp_a = torch.empty(B, device="cuda")
        a_ref = env.sample_actions(logits, rec, seed=3, ticket=t, logp=lp_a)
        lp_b = torch.empty(B, device="cuda")
        lg = torch.empty((B, 26), device="cuda")
        a = pol.act(env, rec, seed=3, ticket=t, logp=lp_b, logits=lg)
        assert torch.equal(a, a_ref) and torch.equal(lp_a, lp_b) and torch.equal(lg, logits)
        rec = env.step(a)
    assert env.counters()["illegal"] == 0
    env.close()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config5_full_size_policy_loop_against_float32_module(precision):
    """BASELINE config 5 at full size - 65 536 four-player games, every action drawn by the action-mask model on the matrix
    cores (policy + value branch in ONE launch, skyjo_vec_mlp_act_value) - against the float32 torch module on the same
    records with the same weights.  Tolerances vs float32 (north_star states none; TOL above): "fp32" - logits and values
    max |diff| <= 1e-4, mean <= 2e-5, mean KL(float32 || kernel) of the masked action distributions <= 1e-7; "bf16" - 8e-2 /
    1e-2 / 1e-3.  The drawn action is legal everywhere, its stored log-probability equals the masked log-softmax of the
    kernel's own logits within 1e-5, no illegal move in 300 iterations, episodes end and the two-launch form gives the same
    bits."""
    import torch

    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import FLOAT_MIN, ActionMaskModel, FusedNet

    torch.manual_seed(7)
    B, N, T = 65536, 4, 300
    env = SkyjoVecEnv(B, num_players=N)
    env.seed(None, 17)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy, precision=precision), FusedNet(model.value, precision=precision)
    tol = TOL[precision]
    rec = env.reset()
    act = torch.empty(B, dtype=torch.int32, device="cuda")
    logp = torch.empty(B, device="cuda")
    logits = torch.empty((B, 26), device="cuda")
    values = torch.empty((B, 1), device="cuda")
    for t in range(T):
        pol.act(env, rec, seed=6, ticket=t, actions=act, logp=logp, logits=logits, value_net=val, values=values)
        if t % 60 == 0:
            v = env.split(rec)
            x = v.observations.to(torch.float32)
            mask = v.action_mask.to(torch.float32)
            with torch.no_grad():
                ref = model.policy(x)
                vref = model.value(x)
            d = (logits - ref).abs()
            assert float(d.max()) <= tol["max"] and float(d.mean()) <= tol["mean"], (t, float(d.max()), float(d.mean()))
            assert float((values - vref).abs().max()) <= tol["max"]
            inf = torch.clamp(torch.log(mask), min=FLOAT_MIN)
            r64, l64, i64 = ref.double(), logits.double(), inf.double()
            kl = (torch.softmax(r64 + i64, -1) * (torch.log_softmax(r64 + i64, -1) - torch.log_softmax(l64 + i64, -1))).sum(-1)
            assert float(kl.mean()) <= tol["kl"], float(kl.mean())
            live = v.done == 0
            assert bool(v.action_mask.gather(1, act.long().unsqueeze(1)).squeeze(1).eq(1)[live].all())
            own = torch.log_softmax(logits + inf, -1).gather(1, act.long().unsqueeze(1)).squeeze(1)
            assert float((own - logp).abs().max()) < 1e-5
            # the same through two launches: bit for bit
            a2 = pol.act(env, rec, seed=6, ticket=t)
            assert torch.equal(a2, act) and torch.equal(val(rec), values)
        rec = env.step(act, out=rec)
    c = env.counters()
    assert c["illegal"] == 0 and c["episodes"] > B and c["steps"] + c["resets"] == T * B + B
    pol.close(), val.close(), env.close()
