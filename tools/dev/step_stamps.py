"""Where does a caller-action k_step wavefront (config 5's step kernel: ONE lockstep iteration per launch) spend its cycles?
-DSK_STAMPS build:  python -c "from skyjo_rl_amd import build; build.build(force=True, extra=['-DSK_STAMPS'], out='build_exp/libskyjo_vec_envstamps.so')"
                    SKYJO_LIB=build_exp/libskyjo_vec_envstamps.so python tools/dev/step_stamps.py [games] [players]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyjo_rl_amd import SkyjoVecEnv, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = SkyjoVecEnv(B, num_players=N)
eng.seed(None, 0)
rec = eng.reset()
zeros = torch.zeros((B, 26), device="cuda")
def loop(n, t0):
    global rec
    for t in range(n):
        rec = eng.step(eng.sample_actions(zeros, rec, seed=1, ticket=t0 + t), out=rec)
loop(200, 0)
torch.cuda.synchronize()
out = np.zeros(16, dtype=np.uint64)
_lib.check(eng._L.skyjo_vec_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
L = 200
loop(L, 200)
torch.cuda.synchronize()
_lib.check(eng._L.skyjo_vec_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
names = ["tile_load (+ wait)", "philox+reset path", "vis row load", "policy_pick", "apply: legality+draw(+finish)", "apply: place", "emit+stores", "tile_store+counters"]
waves = (B + 63) // 64 * L
tot = float(out[:8].sum())
for n, v in zip(names, out[:8]):
    print("%-32s %6.1f%%  %8.0f cycles/wave/launch" % (n, 100 * v / max(tot, 1), v / waves))
print("total cycles/wave/launch %.0f" % (tot / waves))
