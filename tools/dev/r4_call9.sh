#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c9; mkdir -p "$out"
for cfg in "--num-envs 32768" "--num-envs 4096 --num-players 2" "--num-envs 49152" "--num-envs 16384" "--num-envs 32768 --num-players 4" "--num-envs 65536 --num-players 2"; do
  echo "== $cfg"
  ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3 $cfg" SKYJO_BENCH_ALLOW_WAITS=1 timeout -k 10 300 python3 tools/dev/env_ab.py "" "SKYJO_MERGED=1" "SKYJO_OVERLAP=0" 2>&1 | tail -3
done > "$out/ab_sizes.txt" 2>&1
cat "$out/ab_sizes.txt"
SKYJO_MERGED=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_full_batch.py tests/test_gpu_parity.py tests/test_gpu_shard_snapshot.py -m gpu -x -q -k "not config5" > "$out/pytest_merged.log" 2>&1; echo "pytest merged rc=$?"
tail -n 6 "$out/pytest_merged.log"
