"""Diagnostic: k_step time per lockstep iteration while NO game can end (the first 60 iterations after seeding: the
shortest episode is 61 steps) vs steady state (resets in ~45 % of the wavefront-iterations), with and without records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from skyjo_rl_amd import SkyjoVecEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = SkyjoVecEnv(B, num_players=3)
eng.set_overlap(False)
rec = eng.new_records(80)
for records in (rec, None):
    res = []
    for rep in range(3):
        eng.seed(None, rep)
        eng.profile(1)
        eng.rollout(60, 1, records=records[:60] if records is not None else None)
        p = eng.profile(0)
        res.append(p["step_ms"] * 1e3 / 60)
    eng.rollout(800, 1)
    eng.profile(1)
    for _ in range(40):
        eng.rollout(80, 1, records=records)
    p = eng.profile(0)
    print("records" if records is not None else "no records", "no-reset us/iter", ["%.3f" % x for x in res],
          "steady us/iter %.3f" % (p["step_ms"] * 1e3 / 3200), "deal us %.1f" % (p["deal_ms"] * 1e3 / max(p["deal_launches"], 1)))
