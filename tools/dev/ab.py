"""A/B several builds of libskyjo_vec.so in one process-per-variant loop (interleaved rounds, same GPU).
   usage: python tools/ab.py [--env K=V ...] lib1.so lib2.so ...   (a lib may be given as path@K=V to set an env var for it)"""
import json, os, subprocess, sys
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
extra = [a for a in sys.argv[1:] if a.startswith("--")]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        path, *kv = l.split("@")
        env = dict(os.environ, SKYJO_LIB=os.path.abspath(path))
        for x in kv:
            k, v = x.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"] + extra, env=env,
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(l, "FAILED", out.stderr[-400:])
            continue
        res[l].append((d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["value"] / 1e9, d["waits"]))
for l in libs:
    if not res[l]:
        continue
    ks = sorted(x[0] for x in res[l]); dl = sorted(x[1] for x in res[l]); v = sorted(x[2] for x in res[l])
    m = len(ks) // 2
    print("%-40s k_step us median %.1f min %.1f | k_deal us median %.1f | G steps/s median %.2f max %.2f | waits %s" % (
        l[-40:], ks[m], ks[0], dl[m], v[m], v[-1], res[l][0][3]))
