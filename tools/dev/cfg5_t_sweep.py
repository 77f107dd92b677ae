import json, sys
sys.path.insert(0, ".")
import bench
for T in (4, 16, 64, 256):
    r = bench.side_model_config("bf16", 65536, 4, T, max(3, 256 // T), 0)
    print("T=%3d  iter %.1f us  mlp %.1f us  step %.1f us" % (T, 1e3 * r["ms_per_iteration"], 1e3 * r["dominant_kernel_ms"], 1e3 * r["step_kernel_ms"]))
