"""what the host-mapped fast path of the *_host calls costs / buys for mid-size batches.
   python tools/dev/step_host_ab.py      (one line per batch size and variant: microseconds per step_host call)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from skyjo_rl_amd import SkyjoVecEnv
B = int(sys.argv[1])
eng = SkyjoVecEnv(B, num_players=3)
eng.seed(None, 0)
def once():
    o = eng.observe_host()
    return np.argmax(o.action_mask, axis=1).astype(np.int32)
acts = once()
for _ in range(50): eng.step_host(acts); acts = once()
t_obs = time.perf_counter()
for _ in range(200): once()
t_obs = (time.perf_counter() - t_obs) / 200
t0 = time.perf_counter()
n = 300
for _ in range(n):
    eng.step_host(acts); acts = once()
dt = (time.perf_counter() - t0) / n - t_obs
t1 = time.perf_counter()
for _ in range(200): eng.get_state(B - 1)
gs = (time.perf_counter() - t1) / 200
print("%%d step_host %%.1f us  observe_host %%.1f us  get_state %%.1f us" %% (B, dt * 1e6, t_obs * 1e6, gs * 1e6))
''' % ROOT
for B in (64, 256, 1024, 4096):
    for name, env in (("fast path (raw export up to 4 tiles)", {}), ("fast path, no raw export", {"SKYJO_NO_RAW_EXPORT": "1"}), ("hipMemcpy path", {"SKYJO_NO_FAST_HOST": "1"})):
        out = subprocess.run([sys.executable, "-c", CHILD, str(B)], env=dict(os.environ, **env), capture_output=True, text=True)
        print("%-40s %s" % (name, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
