#!/bin/bash
# usage (on a node with N MI355X, from the repo root):  bash tools/scale_run.sh N [bench args...]
#
# One scaling point of BASELINE.json's headline metric with rocprofv3 evidence per rank (north_star: "throughput
# evidenced by rocprof achieved-HBM-GB/s vs peak at 1/2/4/8 GPUs"): starts the N ranks of `bench.py --gpus N` as N separate
# commands, each under its own `rocprofv3 --kernel-trace --stats`, and digests the N traces with tools/scale_collect.py into
# gpurun_out/scale/N<N>/scale_point.json (per rank: k_cycle calls / average us / achieved GB/s against the 8 TB/s peak; the node:
# bench.py's own line).  All four points:  for n in 1 2 4 8; do bash tools/scale_run.sh $n; done; python3 tools/scale_collect.py --table
#
# The program itself follows `--` (python3 bench.py ...): no env / bash -c / torchrun hop under the profiler (its preloaded
# library has initialised the GPU by then, and an exec from such a process is refused on this pool); the rank's environment is
# set by the shell on the rocprofv3 command itself.  bench.py does not start its own ranks when WORLD_SIZE is set.
# NOT run from inside a round (one-GPU boxes; the 8-GPU run is the driver's): staged for the day a node is at hand.
set -u
N=${1:?usage: bash tools/scale_run.sh N [bench args]}; shift
root=$PWD
out=$root/gpurun_out/scale/N$N
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
port=$(python3 -c 'import socket; s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])')
cd /tmp
pids=()
for ((r = 0; r < N; r++)); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$port \
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rank$r" -- \
    python3 "$root/bench.py" --gpus "$N" --steps 100 --warmup 10 --blocks 3 --no-cpu-baseline --no-other-configs "$@" \
    > "$out/rank$r.json" 2> "$out/rank$r.err" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
cd "$root"
echo "ranks done, rc=$rc"
find "$out" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
python3 tools/scale_collect.py "$N" || rc=$?
exit $rc
