"""One-off soak: the fused rollout at the headline size, eight dealing cycles per launch, every record of every iteration against
the oracle over many episodes per game.   python tools/dev/soak_parity.py [launches] [players] [games] [row-major|tile-planar]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import skyjo_oracle as so
from skyjo_rl_amd import SkyjoVecEnv
L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001, rng_mode=0, auto_reset=True)
eng = SkyjoVecEnv(B, **cfg); ora = so.OracleVec(num_envs=B, **cfg)
eng.seed(None, 123); ora.seed(None, 123)
K = eng.deal_interval() * 8
PLANAR = (sys.argv[4] if len(sys.argv) > 4 else "tile-planar") == "tile-planar"
if PLANAR:
    eng.set_record_layout("tile-planar")
rec = eng.new_planar_records(K) if PLANAR else eng.new_records(K)
t0 = time.time()
for r in range(L):
    eng.rollout(K, policy_seed=77, records=rec)
    oact, obs, mask, meta, eplen = ora.rollout(K, 77, threads=min(32, os.cpu_count() or 1), record_obs=True)
    v = eng.split(eng.rows_from_planar(rec) if PLANAR else rec)
    for name, a, b in (("action", v.action, oact.astype(np.int8)), ("obs", v.observations, obs), ("mask", v.action_mask, mask), ("agent", v.agent, meta[..., 0]),
                       ("phase", v.phase, meta[..., 1]), ("done", v.done, meta[..., 2]), ("status", v.status, meta[..., 3])):
        assert np.array_equal(a.cpu().numpy(), b), (name, r)
    assert np.array_equal(v.episode_steps.cpu().numpy().astype(np.uint16), eplen), r
    print("launch", r, "ok", K, "iterations", round(time.time() - t0, 1), "s", flush=True)
c, oc = eng.counters(), ora.counters()
for k in ("steps", "episodes", "resets", "sum_len"):
    assert c[k] == oc[k], (k, c[k], oc[k])
print("SOAK OK", eng.dealing_form(), "episodes", c["episodes"], "waits", c["waits"], "reshuffles", c["reshuffles"])
