"""cfg5 A/B over environment settings: python tools/dev/cfg5_ab.py "K=V" ...  (prints the other_configs cfg5 lines of bench.py's helpers)"""
import json, os, subprocess, sys
code = r'''
import json, sys
sys.path.insert(0, ".")
import bench
out = {}
for prec in ("bf16", "fp32"):
    r = bench.side_model_config(prec, 65536, 4, 64, 4, 0)
    out[prec] = {k: r[k] for k in ("value", "ms_per_iteration", "dominant_kernel_ms", "step_kernel_ms")}
print(json.dumps(out))
'''
for v in (sys.argv[1:] or [""]):
    v = v.replace("LIB=", "SKYJO_LIB=")
    env = dict(os.environ)
    for kv in v.split():
        k, val = kv.split("=", 1)
        env[k] = val
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    try:
        d = json.loads(o.stdout.strip().splitlines()[-1])
        print("%-40s" % repr(v), " | ".join("%s: %.3g steps/s, iter %.1f us, mlp %.1f us, step %.1f us" % (p, d[p]["value"], 1e3 * d[p]["ms_per_iteration"], 1e3 * d[p]["dominant_kernel_ms"], 1e3 * d[p]["step_kernel_ms"]) for p in d))
    except Exception:
        print(repr(v), "FAILED", o.stderr[-500:])
