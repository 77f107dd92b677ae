#!/bin/bash
# usage (GPU box, repo root): bash tools/dev/cfg5_traffic.sh  -> gpurun_out/cfg5traffic/{fetch,write}/  + a digest on stdout
# HBM-side bytes per launch of config 5's kernels (the caller-action step above all): FETCH_SIZE and WRITE_SIZE in separate passes.
set -u
root=$PWD; out=$root/gpurun_out/cfg5traffic; rm -rf "$out"; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- python3 "$root/tools/bench_cfg5.py" 65536 64 2 > "$out/$c.json" 2> "$out/$c.err"; echo "$c rc=$?"
done
cd "$root"
python3 - <<'P'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/cfg5traffic/%s/**/*counter_collection.csv" % c, recursive=True)
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            agg[k] += float(r["Counter_Value"]); n[k] += 1
    for k in agg:
        if n[k] > 50: print("%-11s %-62s %6d dispatches  %10.1f KiB per dispatch" % (c, k, n[k], agg[k] / n[k]))
P
