import sys, os
sys.path.insert(0, os.getcwd())
import torch
from skyjo_rl_amd import SkyjoVecEnv
eng = SkyjoVecEnv(65536, num_players=3); eng.seed(None, 0)
rec = eng.new_records(16); act = torch.empty((16, 65536), dtype=torch.int32, device="cuda")
for _ in range(40): eng.rollout(16, 1, records=rec, actions=act)
c0 = eng.counters()
for _ in range(40): eng.rollout(16, 1, records=rec, actions=act)
c1 = eng.counters()
d = {k: c1[k] - c0[k] for k in c0 if isinstance(c0[k], (int, float))}
print({k: v for k, v in d.items() if v})
it = 640; waves = 1024
print("per wave-iter: reshuffles %.4f resets %.4f waits %.5f episodes %.4f" % (d["reshuffles"]/it/waves, d["resets"]/it/waves, d["waits"]/it/waves, d["episodes"]/it/waves))
