#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c4; mkdir -p "$out"
SKYJO_MERGED=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_full_batch.py -m gpu -x -q -k "cfg3 or cfg2 or cfg4 or philox or inplace_deals_interval1000" > "$out/pytest_merged.log" 2>&1; echo "pytest merged rc=$?"
tail -n 8 "$out/pytest_merged.log"
ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "" "SKYJO_MERGED=1" "SKYJO_MERGED=1 SKYJO_DEAL_INTERVAL=88" > "$out/ab.txt" 2>&1; cat "$out/ab.txt"
SKYJO_MERGED=1 SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 65536 61 > "$out/place_merged.json" 2> "$out/place_merged.err"; echo "place rc=$?"
