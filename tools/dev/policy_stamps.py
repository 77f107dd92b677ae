"""Per-phase cycle counts of the policy net's wavefronts (diagnostic build -DSKP_STAMPS of csrc/skyjo_policy.hip):
    python -m skyjo_rl_amd.build   # the product
    python tools/dev/policy_stamps.py build        # -> build_exp/libskyjo_vec_stamps.so  (CPU box)
    SKYJO_LIB=build_exp/libskyjo_vec_stamps.so python tools/dev/policy_stamps.py [bf16|fp32]   (GPU box)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from skyjo_rl_amd import build
    print(build.build(force=True, verbose=True, extra=["-DSKP_STAMPS"], out=os.path.join(ROOT, "build_exp", "libskyjo_vec_stamps.so")))
    sys.exit(0)
import numpy as np
import torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
r = bench.side_model_config(prec, 65536, 4, 16, 2, 0)
# (the last launch of a rollout is the value net alone - model_tail -: look at the waves of a two-net launch only by running
# one more two-net launch at the end is not possible through bench; instead the tail launch has 1024 waves: ids 0..1023 are overwritten)
torch.cuda.synchronize()
from skyjo_rl_amd import _lib
L = _lib.load()
NS, NW = 32, 2048
buf = np.zeros(NW * NS, dtype=np.uint64)
rc = L.skyjo_debug_policy_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
assert rc == 0, rc
st = buf.reshape(NW, NS).astype(np.int64)
names = ["start", "loads", "dma-issued", "layer1", "arrived01"] + ["pre-stage%d" % u for u in range(8)] + ["stages-done", "finish"]
out = {"precision": prec, "kernel_us": 1e3 * r["dominant_kernel_ms"], "waves": NW}
rows = []
def seg(a, b, label):
    d = st[:, b] - st[:, a]
    rows.append((label, float(np.median(d)), float(d.min()), float(d.max())))
seg(0, 1, "pass0 record waited, loads issued (fp32: + first two tiles staged)")
if prec == "bf16":
    seg(0, 2, "  pass0: own two KiB of layer 1 arrived and written")
    seg(2, 16, "  pass0: barrier (layer 1 in LDS)")
    seg(16, 1, "  pass0: loads issued, fragments read, record unpacked")
seg(1, 3, "pass0 layer 1 (+ staging writes)")
if prec == "bf16":
    seg(3, 4, "pass0 barrier (weights in LDS)")
    seg(4, 5, "pass0 chain(0)")
else:
    seg(3, 5, "pass0 barrier before stage 0")
for u in range(8):
    seg(5 + u, 6 + u, "pass0 stage %d (+ wait)" % u)
seg(13, 14, "pass0 finish (draw / stores)")
seg(14, 15, "pass1 record waited, loads issued")
seg(15, 17, "pass1 layer 1")
seg(17, 19, "pass1 chain(0) / barrier")
for u in range(8):
    seg(19 + u, 20 + u, "pass1 stage %d" % u)
seg(27, 28, "pass1 finish")
seg(0, 28, "whole wavefront")
pol = st[:1024]; val = st[1024:]
out["whole_policy_med"] = float(np.median(pol[:, 28] - pol[:, 0])); out["whole_value_med"] = float(np.median(val[:, 28] - val[:, 0]))
out["span_all_waves_us"] = float(st[:, 30].max() - st[:, 29].min()) / 100.0
out["wave_life_us_median"] = float(np.median(st[:, 30] - st[:, 29])) / 100.0
out["start_skew_us"] = float(st[:, 29].max() - st[:, 29].min()) / 100.0
out["cycles_per_us_median"] = float(np.median((st[:, 28] - st[:, 0]) / np.maximum(1, (st[:, 30] - st[:, 29]) / 100.0)))
for lab, med, mn, mx in rows:
    print("%-40s median %8.0f  min %8.0f  max %8.0f" % (lab, med, mn, mx))
print(json.dumps(out))
