"""When do the wavefronts of a caller-action step launch (config 5's step kernel: ONE lockstep iteration per launch) start and end?
-DSK_TRACE build:  python -c "from skyjo_rl_amd import build; build.build(force=True, extra=['-DSK_TRACE'], out='build_exp/libskyjo_vec_trace.so')"
                   SKYJO_LIB=build_exp/libskyjo_vec_trace.so python tools/dev/step_trace.py [games] [players]
s_memrealtime (100 MHz) at the first and last instruction of every wavefront of the last two launches: start skew, wavefront life, the
span from the first start to the last end - to hold against the kernel's duration in a rocprofv3 trace (10.9 us at 65 536 x 4)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyjo_rl_amd import SkyjoVecEnv, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = SkyjoVecEnv(B, num_players=N)
eng.seed(None, 0)
rec = eng.reset()
zeros = torch.zeros((B, 26), device="cuda")
for t in range(301):
    rec = eng.step(eng.sample_actions(zeros, rec, seed=1, ticket=t), out=rec)
torch.cuda.synchronize()
tiles = (B + 63) // 64
tr = np.zeros((4, tiles, 8), dtype=np.uint64)
_lib.check(eng._L.skyjo_vec_debug_trace(eng._h, tr.ctypes.data_as(C.c_void_p)))
out = {"games": B, "players": N}
for slot in (0, 1):
    rows = tr[slot].astype(np.int64)
    if not rows[:, 3].any():
        continue
    t0, t1, cyc = rows[:, 2], rows[:, 3], rows[:, 5]
    hw, xcc = rows[:, 0], rows[:, 1] & 0xf
    cu = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    out["launch_slot_%d" % slot] = {
        "start_skew_us": float(t0.max() - t0.min()) / 100.0, "wave_life_us_median": float(np.median(t1 - t0)) / 100.0,
        "wave_life_us_max": float((t1 - t0).max()) / 100.0, "span_first_start_to_last_end_us": float(t1.max() - t0.min()) / 100.0,
        "start_offsets_us_percentiles_50_90_99": [float(np.percentile(t0 - t0.min(), p)) / 100.0 for p in (50, 90, 99)],
        "wave_life_us_percentiles_10_50_90_99_100": [float(np.percentile(t1 - t0, p)) / 100.0 for p in (10, 50, 90, 99, 100)],
        "wave_life_us_median_by_xcc": [float(np.median((t1 - t0)[xcc == x])) / 100.0 for x in range(8) if (xcc == x).any()],
        "wave_end_us_percentiles_50_90_99_100": [float(np.percentile(t1 - t0.min(), p)) / 100.0 for p in (50, 90, 99, 100)],
        "slowest_tiles": [int(i) for i in np.argsort(t1 - t0)[-8:]],
        "clock_GHz_median": float(np.median(cyc / np.maximum(1, t1 - t0))) / 10.0, "compute_units_used": int(len(per_cu)),
        "waves_per_cu_max": int(per_cu.max())}
print(json.dumps(out, indent=1))
