"""A/B several builds of libskyjo_vec.so in one process-per-variant loop (interleaved rounds, same GPU)."""
import json, os, subprocess, sys
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, SKYJO_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "640", "--warmup", "64", "--no-cpu-baseline"], env=env,
                             capture_output=True, text=True).stdout
        d = json.loads(out)
        res[l].append((d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["value"] / 1e9))
for l in libs:
    ks = sorted(x[0] for x in res[l]); dl = sorted(x[1] for x in res[l]); v = sorted(x[2] for x in res[l])
    print("%-22s k_step us median %.1f min %.1f | k_deal us median %.1f | G steps/s median %.2f" % (os.path.basename(l), ks[1], ks[0], dl[1], v[1]))
