"""A/B of library builds on long runs (banks settled): python tools/ab_long.py lib1.so lib2.so ...  [env: STEPS]"""
import json, os, subprocess, sys
libs = sys.argv[1:]
steps = os.environ.get("STEPS", "7040")
for l in libs:
    env = dict(os.environ, SKYJO_LIB=os.path.abspath(l))
    out = subprocess.run([sys.executable, "bench.py", "--steps", steps, "--warmup", "400", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True).stdout
    d = json.loads(out)
    print("%-20s %.2fe9 steps/s  k_step %.1f us  k_deal %.1f us  waits %d" % (
        os.path.basename(l), d["value"] / 1e9, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["waits"]))
