"""Output error of the policy / value net kernels against the float32 torch module and its float64 evaluation (65 536 x 4 players, weights x 1.5 as in
tests/test_gpu_policy_net.py):  [SKYJO_LIB=build_exp/lib...so] python tools/dev/net_error.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from skyjo_rl_amd import SkyjoVecEnv
from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
torch.manual_seed(1)
B = 65536
env = SkyjoVecEnv(B, num_players=4); env.seed(None, 5)
model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
with torch.no_grad():
    for p in model.parameters(): p.mul_(1.5)
rec = env.reset()
for t in range(5):
    rec = env.step(env.sample_actions(torch.zeros((B, 26), device="cuda"), rec, seed=1, ticket=t))
x = env.split(rec).observations.to(torch.float32)
for prec in ("fp32", "bf16"):
    for name, seq in (("policy", model.policy), ("value", model.value)):
        net = FusedNet(seq, precision=prec)
        got = net(rec)
        with torch.no_grad():
            ref = seq(x); ref64 = seq.double()(x.double()); seq.float()
        d = (got - ref).abs(); d64 = (got.double() - ref64).abs()
        print(prec, name, "vs f32 module: max %.3g mean %.3g | vs f64: max %.3g mean %.3g" % (float(d.max()), float(d.mean()), float(d64.max()), float(d64.mean())))
        net.close()
