#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c5; mkdir -p "$out"
for v in "inline:SKYJO_OVERLAP=0" "overlap:SKYJO_OVERLAP=1" "merged:SKYJO_MERGED=1"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 65536 61 > "$out/place_$tag.json" 2> "$out/place_$tag.err"; echo "place $tag rc=$?"
done
env PHILOX=1 SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 98304 61 > "$out/place_98k_philox.json" 2> "$out/place_98k.err"; echo "place 98k rc=$?"
grep -h "shader_clock\|wave_us_mean\|^ \"k_\|games\|rng" "$out"/place_*.json | head -80
