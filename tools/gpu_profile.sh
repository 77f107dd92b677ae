#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag> [bench args...]
# kernel-trace + stats of a short bench run; summaries end up in gpurun_out/prof_<tag>/
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$OLDPWD/bench.py" --steps 100 --warmup 10 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
echo "rc=$?"
cd "$OLDPWD"
find "$out" -name "*kernel_stats*.csv" | head -3
f=$(find "$out" -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
cat "$out/bench.json"
