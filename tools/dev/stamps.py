"""Diagnostic: where does a k_step wavefront spend its cycles?  Needs a -DSK_STAMPS build:
   hipcc ... -DSK_STAMPS -o /tmp/libskyjo_stamps.so ; SKYJO_LIB=/tmp/libskyjo_stamps.so python tools/stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyjo_rl_amd import SkyjoVecEnv, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = SkyjoVecEnv(B, num_players=3)
eng.seed(None, 0)
eng.set_deal_interval(eng.deal_interval())
ITERS = eng.deal_interval() * int(os.environ.get("CYCLES", "8"))  # one k_cycle launch of that many dealing cycles
mode = sys.argv[2] if len(sys.argv) > 2 else "rec"
act = None
if mode == "planar":
    eng.set_record_layout("tile-planar"); rec = eng.new_planar_records(ITERS)
elif mode == "norec":
    rec = None
else:
    rec = eng.new_records(ITERS)
print("mode", mode, "iterations per launch", ITERS, "form", eng.dealing_form())
for _ in range(20): eng.rollout(ITERS, 1, records=rec, actions=act)
torch.cuda.synchronize()
out = np.zeros(16, dtype=np.uint64)
_lib.check(eng._L.skyjo_vec_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
for _ in range(10): eng.rollout(ITERS, 1, records=rec, actions=act)
torch.cuda.synchronize()
_lib.check(eng._L.skyjo_vec_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
names = ["tile_load", "reset commit", "top: philox, spare issue, vis read (+pick)", "draw path (+finish)", "place: legality", "place A: reads", "place B: update (+draw join)", "emit+stores (+tile_store)"] if os.environ.get("FINE") else ["tile_load", "philox+reset path", "vis row load", "policy_pick", "apply: legality+draw(+finish)", "apply: place", "emit+stores", "tile_store+counters"]
dn = ["scan+compact", "-", "stream open", "deal_into_lds (shuffles etc.)", "tile_store", "MT refill (wave-synchronous)", "-", "-"]
dt = float(out[8:].sum()); dwaves = (B + 255) // 256 * 10
for n, v in zip(dn, out[8:]):
    print("deal %-25s %6.1f%%  %9.0f cycles/wave/launch" % (n, 100 * v / max(dt, 1), v / dwaves))
tot = float(out[:8].sum()); waves = (B + 63) // 64 * 10
for n, v in zip(names, out[:8]):
    print("%-30s %6.1f%%  %9.0f cycles/wave/launch  %8.0f /iter" % (n, 100 * v / tot, v / waves, v / waves / ITERS))
print("total cycles/wave/launch %.0f" % (tot / waves))
