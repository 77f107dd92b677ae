"""Kernel time of skyjo_vec_mlp_forward (policy net on the matrix cores) at 65 536 records, torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
B = 65536
env = SkyjoVecEnv(B, num_players=4); env.seed(None, 1); rec = env.reset()
model = ActionMaskModel(obs_dim=31).cuda(); net = FusedNet(model.policy)
out = torch.empty((B, 26), device="cuda")
for _ in range(10): net(rec, out=out)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): net(rec, out=out)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 10
flops = 2 * B * (32 * 256 + 256 * 256 + 256 * 32)
print("mlp forward %.1f us per 65536 records  %.1f TFLOP/s (bf16 MFMA, dense peak 2500)" % (us, flops / us / 1e6))
x = env.split(rec).observations.float()
for _ in range(10): model.policy(x)
a.record()
for _ in range(100): model.policy(x)
b.record(); torch.cuda.synchronize()
print("torch fp32 policy net %.1f us" % (a.elapsed_time(b) * 10))
