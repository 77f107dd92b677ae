#!/bin/bash
# placement traces (SK_TRACE build) + the GPU suite
set -u
root=$PWD; out=$root/gpurun_out/r4c2; mkdir -p "$out"
for v in "ov0:SKYJO_OVERLAP=0" "ov1:SKYJO_OVERLAP=1" "ov1_philox:SKYJO_OVERLAP=1 PHILOX=1"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 65536 61 > "$out/place_$tag.json" 2> "$out/place_$tag.err"; echo "place $tag rc=$?"
done
env PHILOX=1 SKYJO_LIB=$root/build_exp/lib_trace.so timeout -k 10 200 python3 tools/dev/placement.py 98304 61 > "$out/place_98k_philox.json" 2> "$out/place_98k.err"; echo "place 98k rc=$?"
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 25 "$out/pytest.log"
