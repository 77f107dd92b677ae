"""A/B of environment settings (and library builds) on the headline bench: python tools/dev/env_ab.py "K=V K2=V2" "K=V" ...
Each argument is one variant (space-separated environment assignments, "" = defaults; LIB=path selects a build)."""
import json, os, subprocess, sys
variants = sys.argv[1:] or [""]
rounds = int(os.environ.get("ROUNDS", "2"))
extra = os.environ.get("BENCH_ARGS", "--steps 40 --warmup 10 --blocks 3").split()
res = {v: [] for v in variants}
for rnd in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=", 1)
            env["SKYJO_LIB" if k == "LIB" else k] = os.path.abspath(val) if k == "LIB" else val
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs"] + extra, env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:
            print(repr(v), "FAILED", out.stderr[-600:])
            continue
        res[v].append((d["value"] / 1e9, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["waits"], d["config"]["iterations_per_step"], d["blocks"]["min"] / 1e9, d["blocks"]["max"] / 1e9))
for v in variants:
    for r in res[v]:
        print("%-60s %.2f G steps/s (blocks %.2f..%.2f) k_step %.1f us k_deal %.1f us waits %d interval %d" % (repr(v), r[0], r[5], r[6], r[1], r[2], r[3], r[4]))
