#!/bin/bash
# usage (GPU box, repo root): bash tools/dev/pcs.sh <tag> [stochastic|host_trap]   - PC sampling of the fused rollout, nothing else traced
set -u
tag=${1:-pcs}; method=${2:-stochastic}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
if [ "$method" = stochastic ]; then unit=cycles; iv=${PCS_INTERVAL:-1048576}; else unit=time; iv=${PCS_INTERVAL:-100}; fi
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
  --output-format csv -d "$out" -- python3 "$root/tools/dev/pcs_run.py" > "$out/run.log" 2> "$out/run.err"
echo "rc=$?"
cd "$root"
find "$out" -type f | head; tail -3 "$out/run.err"
f=$(find "$out" -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && { head -3 "$f"; wc -l "$f"; }
