"""A short fused-rollout run for rocprofv3's PC sampling (tools/dev/pcs.sh): 65 536 x 3, eight dealing cycles per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
B = int(os.environ.get("PCS_B", "65536")); N = int(os.environ.get("PCS_N", "3"))
eng = SkyjoVecEnv(B, num_players=N)
eng.seed(None, 0)
eng.set_deal_interval(eng.deal_interval())
K = eng.deal_interval() * 8
if os.environ.get("PCS_LAYOUT", "tile-planar") == "tile-planar":
    eng.set_record_layout("tile-planar"); rec = eng.new_planar_records(K)
else:
    rec = eng.new_records(K)
for _ in range(int(os.environ.get("PCS_LAUNCHES", "160"))):
    eng.rollout(K, policy_seed=1, records=rec)
torch.cuda.synchronize()
print("done", eng.counters()["steps"])
