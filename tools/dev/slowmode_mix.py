"""Does the dealing kernel's slow mode show when processes of different shapes alternate?  usage: slowmode_mix.py lib.so [rounds]"""
import json, os, subprocess, sys
lib = os.path.abspath(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
env = dict(os.environ, SKYJO_LIB=lib)
others = [32768, 131072, 16384, 98304, 4096, 49152]
vals = []
for r in range(rounds):
    subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--num-envs", str(others[r % len(others)])],
                   env=dict(env, SKYJO_BENCH_ALLOW_WAITS="1"), capture_output=True, text=True)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1]); vals.append(d["roofline"]["deal_kernel_avg_ms"] * 1e3)
    except Exception:
        vals.append(-1.0)
print(os.path.basename(lib), "k_deal us at 65536 games:", " ".join("%.0f" % v for v in vals), "| slow:", sum(v > 80 for v in vals))
