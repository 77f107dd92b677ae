#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c10; mkdir -p "$out"
{
echo "== 65536 x 3"; ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_DEAL_INTERVAL=48" "SKYJO_DEAL_INTERVAL=56" "SKYJO_DEAL_INTERVAL=60" "SKYJO_DEAL_INTERVAL=64" "SKYJO_DEAL_INTERVAL=68" 2>&1 | tail -10
echo "== 65536 x 2"; ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3 --num-players 2" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_DEAL_INTERVAL=40" "SKYJO_DEAL_INTERVAL=48" "SKYJO_DEAL_INTERVAL=56" "SKYJO_DEAL_INTERVAL=64" 2>&1 | tail -4
echo "== 32768 x 3"; ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3 --num-envs 32768" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_DEAL_INTERVAL=56" "SKYJO_DEAL_INTERVAL=64" "SKYJO_DEAL_INTERVAL=72" "SKYJO_DEAL_INTERVAL=80" "SKYJO_DEAL_INTERVAL=88" 2>&1 | tail -5
echo "== 4096 x 2"; ROUNDS=1 BENCH_ARGS="--steps 80 --warmup 10 --blocks 3 --num-envs 4096 --num-players 2" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_DEAL_INTERVAL=40" "SKYJO_DEAL_INTERVAL=48" "SKYJO_DEAL_INTERVAL=56" "SKYJO_DEAL_INTERVAL=64" 2>&1 | tail -4
echo "== 32768 x 4"; ROUNDS=1 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3 --num-envs 32768 --num-players 4" timeout -k 10 400 python3 tools/dev/env_ab.py "SKYJO_DEAL_INTERVAL=64" "SKYJO_DEAL_INTERVAL=80" "SKYJO_DEAL_INTERVAL=96" "SKYJO_DEAL_INTERVAL=112" 2>&1 | tail -4
} > "$out/ab_interval.txt" 2>&1
cat "$out/ab_interval.txt"
