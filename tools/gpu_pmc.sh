#!/bin/bash
# usage: bash tools/gpu_pmc.sh <tag> "<counters>" [bench args]   (PMC pass: kernel-trace only, no other tracing)
set -u
tag=$1; ctrs=$2; shift; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -- python3 "$OLDPWD/bench.py" --steps 30 --warmup 10 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "rc=$?"
cd "$OLDPWD"
f=$(find "$out" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(agg[k].items()):
        print("   %-28s total %.4g  per-dispatch %.4g" % (c, v, v / len(n[k])))
PY
