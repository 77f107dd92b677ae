"""How often does a process land in the dealing kernel's slow mode?  usage: python tools/dev/slowmode_probe.py [runs]
Each run is a fresh process: 30 launches of the headline configuration, prints k_deal's average time."""
import json, os, subprocess, sys
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
vals = []
for i in range(runs):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        vals.append(d["roofline"]["deal_kernel_avg_ms"] * 1e3)
    except Exception:
        vals.append(-1.0)
print("k_deal us per process:", " ".join("%.0f" % v for v in vals))
print("slow (> 80 us): %d of %d" % (sum(v > 80 for v in vals), len(vals)))
