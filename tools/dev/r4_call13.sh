#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c13; mkdir -p "$out"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py -m gpu -x -q -k "not config5" > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 6 "$out/pytest.log"
ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_BENCH_CYCLES=1" "SKYJO_BENCH_CYCLES=2" "SKYJO_BENCH_CYCLES=4" "SKYJO_BENCH_CYCLES=8" > "$out/ab.txt" 2>&1; cat "$out/ab.txt"
