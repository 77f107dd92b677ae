import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyjo_rl_amd import SkyjoVecEnv
B, N = 4096, 3
eng = SkyjoVecEnv(B, num_players=N)
eng.seed(None, 0)
for r in range(10):
    rec = eng.new_records(16)
    eng.rollout(16, policy_seed=1, records=rec)
    torch.cuda.synchronize()
    v = eng.split(rec[15])
    rew, sc, done = eng.rewards_host()
    print(r, "rec done", int(v.done.sum()), "P.done", int(done.sum()), "state done", sum(eng.get_state(g)["done"] for g in range(64)))
