#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c7; mkdir -p "$out"
ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "" "SKYJO_MERGED=1" "SKYJO_MERGED=1 SKYJO_CYCLE_SPLIT=1" "SKYJO_MERGED=1 SKYJO_DEAL_INTERVAL=88" "SKYJO_MERGED=1 SKYJO_DEAL_INTERVAL=72" > "$out/ab.txt" 2>&1; cat "$out/ab.txt"
SKYJO_MERGED=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_full_batch.py tests/test_gpu_parity.py -m gpu -x -q -k "not generic and not direct and not config5" > "$out/pytest_merged.log" 2>&1; echo "pytest merged rc=$?"
tail -n 8 "$out/pytest_merged.log"
SKYJO_MERGED=1 SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 65536 61 > "$out/place_merged.json" 2> "$out/place_merged.err"; echo "place rc=$?"
python3 - "$out/place_merged.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and "wave_cycles_mean" in v: print(k, "wave_us", round(v["wave_us_mean"],1), "launch_us", round(v["launch_us"],1), "p10/50/90", [round(x,1) for x in v["wave_us_p10_p50_p90_max"]], "simd_hist", v["waves_per_simd_hist"])
    elif isinstance(v,dict): print(k, v)
PY
