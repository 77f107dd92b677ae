"""Diagnostic: how the self-adapting dealing interval behaves for 2, 3 and 4 players (python tools/probe_interval.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
for N in (3, 2, 4):
    eng = SkyjoVecEnv(65536, num_players=N); eng.seed(None, 0)
    rec = eng.new_records(80); act = torch.empty((80, 65536), dtype=torch.int32, device="cuda")
    out = []
    for r in range(100):
        eng.rollout(80, 1, records=rec, actions=act)
        if r % 10 == 9: out.append(eng.deal_interval())
    print("N", N, "interval over time", out, "waits", eng.counters()["waits"])
    eng.close()
