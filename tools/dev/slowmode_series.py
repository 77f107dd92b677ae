"""Time series of the dealing kernel's duration inside ONE process (does the slow mode set in over time?).
usage: python tools/dev/slowmode_series.py [chunks] [launches_per_chunk]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = 65536
eng = SkyjoVecEnv(B, num_players=3)
eng.seed(None, 0)
it = eng.deal_interval()
rec = eng.new_records(it)
for _ in range(10): eng.rollout(it, 1, records=rec)
torch.cuda.synchronize()
series = []
for c in range(chunks):
    eng.profile(1)
    for _ in range(per): eng.rollout(it, 1, records=rec)
    torch.cuda.synchronize()
    p = eng.profile(0)
    series.append((p["k_deal_ms"] / max(p["k_deal_launches"], 1) * 1e3, p["k_step_ms"] / max(p["k_step_launches"], 1) * 1e3))
print("k_deal us:", " ".join("%.0f" % a for a, b in series))
print("k_step us:", " ".join("%.0f" % b for a, b in series))
