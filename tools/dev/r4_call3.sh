#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c3; mkdir -p "$out"
ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" SKYJO_BENCH_ALLOW_WAITS=1 timeout -k 10 600 python3 tools/dev/env_ab.py "" "SKYJO_OVERLAP=1" \
  "LIB=build_exp/lib_nols.so" "LIB=build_exp/lib_nols.so SKYJO_OVERLAP=1" "LIB=build_exp/lib_nol.so SKYJO_OVERLAP=1" "LIB=build_exp/lib_nos.so SKYJO_OVERLAP=1" \
  "LIB=build_exp/lib_nop.so SKYJO_OVERLAP=1" "LIB=build_exp/lib_nol.so" "LIB=build_exp/lib_nos.so" > "$out/ab_vmem.txt" 2>&1; echo "ab rc=$?"
cat "$out/ab_vmem.txt"
timeout -k 10 900 python3 -m pytest tests/test_gpu_policy_stats.py tests/test_gpu_ppo_handoff.py tests/test_gpu_rollout_buffer.py tests/test_gpu_sampler.py tests/test_gpu_scoring_helpers.py tests/test_gpu_shard_snapshot.py tests/test_gpu_views.py -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 25 "$out/pytest.log"
