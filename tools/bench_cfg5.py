"""Config 5 of BASELINE.json on its own (the same measurement bench.py reports under other_configs): 65 536 parallel
4-player games stepped by the action-mask policy model (rlskyjo/models/action_mask_model.py:13-77 restated without Ray: RLlib's
default 256-256 tanh net, random weights), policy + value branch in ONE launch on the matrix cores with the masked draw in its
epilogue, collected by skyjo_vec_model_rollout (two launches per lockstep iteration, rollout columns written).
    python tools/bench_cfg5.py [B] [T] [rounds] [fp32|bf16]        (one JSON line; used under rocprofv3 by tools/refresh_profiles.sh)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
only = sys.argv[4] if len(sys.argv) > 4 else None
out = {"config": f"{B} x 4 players, action-mask model in the loop", "T": T, "rounds": ROUNDS}
for prec in ("fp32", "bf16"):
    if only and prec != only:
        continue
    out[prec] = bench.side_model_config(prec, B, 4, T, ROUNDS, 0)
print(json.dumps(out))
