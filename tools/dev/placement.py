"""Diagnostic (needs a -DSK_TRACE build: SKYJO_LIB=build_exp/lib_trace.so): where and when the wavefronts of the last two
k_step / k_deal launches ran - per CU / SIMD placement, start offsets (a second round?), duration against the number of
dealing wavefronts that shared the SIMD / the CU.   python tools/dev/placement.py [games] [launches]"""
import collections, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from skyjo_rl_amd import SkyjoVecEnv, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = int(sys.argv[2]) if len(sys.argv) > 2 else 61
rng = 1 if os.environ.get("PHILOX") else 0
eng = SkyjoVecEnv(B, num_players=3, rng_mode=rng)
eng.seed(None, 0)
K = int(os.environ.get("PLACE_K", "0")) or eng.deal_interval()
rec = eng.new_records(K)
for _ in range(L):
    eng.rollout(K, 1, records=rec)
torch.cuda.synchronize()
tiles = (B + 63) // 64
tr = np.zeros((4, tiles, 8), dtype=np.uint64)
_lib.check(eng._L.skyjo_vec_debug_trace(eng._h, tr.ctypes.data_as(C.c_void_p)))


def decode(rows):
    hw, xcc = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64) & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    global WAITS
    WAITS = rows[:, 6].astype(np.int64)
    return cu_key, cu_key * 4 + simd, rows[:, 2].astype(np.int64), rows[:, 3].astype(np.int64), int(rows[0, 4]), rows[:, 5].astype(np.int64)


out = {"games": B, "interval": K, "overlap": eng.overlap(), "rng": "philox" if rng else "mt19937"}
launches = {}
for kind, name in ((0, "k_step"), (1, "k_deal")):
    for slot in (0, 1):
        rows = tr[2 * kind + slot]
        if not rows[:, 3].any():
            continue
        launches[(name, slot)] = decode(rows)
        out.setdefault("barrier_wait_cycles_mean", {})[f"{name}[{slot}]"] = float(WAITS.mean())
        if kind == 1:
            out.setdefault("deal_barrier_wait_cycles_total_mean (all launches so far)", {})[f"{name}[{slot}]"] = float(rows[:, 7].astype(np.int64).mean())
t_ref = min(v[2].min() for v in launches.values())
for (name, slot), (cu, simd, t0, t1, tag, cyc) in sorted(launches.items()):
    d = (t1 - t0) / 100.0  # us
    ghz = cyc / np.maximum(t1 - t0, 1) * 0.1
    out[f"{name}[{slot}]"] = {"tag": tag, "start_us": (t0.min() - t_ref) / 100.0, "end_us": (t1.max() - t_ref) / 100.0,
                              "launch_us": (t1.max() - t0.min()) / 100.0, "wave_us_mean": float(d.mean()),
                              "shader_clock_GHz_p10_p50_p90": [float(np.percentile(ghz, q)) for q in (10, 50, 90)], "wave_cycles_mean": float(cyc.mean()), "wave_us_p10_p50_p90_max": [float(np.percentile(d, q)) for q in (10, 50, 90, 100)],
                              "start_offset_us_p50_p90_p99_max": [float(np.percentile((t0 - t0.min()) / 100.0, q)) for q in (50, 90, 99, 100)],
                              "cus_used": int(len(np.unique(cu))), "waves_per_cu_hist": dict(collections.Counter(collections.Counter(cu.tolist()).values())),
                              "waves_per_simd_hist": dict(collections.Counter(collections.Counter(simd.tolist()).values()))}
# the k_step launch that ran beside a dealing launch: pair by time overlap
for ks in [k for k in launches if k[0] == "k_step"]:
    cu_s, simd_s, s0, s1, _, _c = launches[ks]
    for kd in [k for k in launches if k[0] == "k_deal"]:
        cu_d, simd_d, d0, d1, _, _c2 = launches[kd]
        ov = min(s1.max(), d1.max()) - max(s0.min(), d0.min())
        if ov <= 0:
            continue
        # per step wavefront: dealing wavefronts on its SIMD / its CU whose life overlaps its own by more than half
        per_simd, per_cu = collections.defaultdict(list), collections.defaultdict(list)
        for i in range(len(simd_d)):
            per_simd[int(simd_d[i])].append((d0[i], d1[i]))
            per_cu[int(cu_d[i])].append((d0[i], d1[i]))
        def share(table, key, a, b):
            return sum(1 for (x, y) in table.get(int(key), ()) if min(b, y) - max(a, x) > 0.5 * (b - a))
        n_simd = np.array([share(per_simd, simd_s[i], s0[i], s1[i]) for i in range(len(simd_s))])
        n_cu = np.array([share(per_cu, cu_s[i], s0[i], s1[i]) for i in range(len(cu_s))])
        dur = (s1 - s0) / 100.0
        out[f"pair {ks[0]}[{ks[1]}] x {kd[0]}[{kd[1]}]"] = {
            "overlap_us": ov / 100.0,
            "step_wave_us_by_dealing_waves_on_its_simd": {int(n): [int((n_simd == n).sum()), float(dur[n_simd == n].mean())] for n in np.unique(n_simd)},
            "step_wave_us_by_dealing_waves_on_its_cu": {int(n): [int((n_cu == n).sum()), float(dur[n_cu == n].mean())] for n in np.unique(n_cu)}}
print(json.dumps(out, indent=1, default=str))
