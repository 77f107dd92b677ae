#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c8; mkdir -p "$out"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 12 "$out/pytest.log"
SKYJO_MERGED=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_full_batch.py tests/test_gpu_parity.py tests/test_gpu_shard_snapshot.py -m gpu -x -q -k "not config5" > "$out/pytest_merged.log" 2>&1; echo "pytest merged rc=$?"
tail -n 12 "$out/pytest_merged.log"
ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "" "SKYJO_DEAL_INTERVAL=64" "SKYJO_DEAL_INTERVAL=72" "SKYJO_DEAL_INTERVAL=76" "SKYJO_DEAL_INTERVAL=84" > "$out/ab.txt" 2>&1; cat "$out/ab.txt"
