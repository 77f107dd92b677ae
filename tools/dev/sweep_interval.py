"""Sweep the dealing interval (lockstep iterations between dealing runs): bench value, k_deal time, slow-path deals."""
import json, os, subprocess, sys
for iv in [int(x) for x in sys.argv[1:]] or [64, 80, 96, 112, 128]:
    env = dict(os.environ, SKYJO_DEAL_INTERVAL=str(iv))
    vals = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "bench.py", "--steps", "1920", "--warmup", "192", "--no-cpu-baseline"], env=env,
                             capture_output=True, text=True).stdout
        d = json.loads(out)
        vals.append((d["value"] / 1e9, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["waits"]))
    print("interval %4d: " % iv + " | ".join("%.2f G/s k_step %.1f us k_deal %.1f us waits %d" % v for v in vals), flush=True)
