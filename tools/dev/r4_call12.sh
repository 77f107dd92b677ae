#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c12; mkdir -p "$out"
timeout -k 10 600 python3 -m pytest tests/test_gpu_rollout_buffer.py tests/test_gpu_model_loop.py tests/test_gpu_ppo_handoff.py tests/test_gpu_policy_net.py -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 8 "$out/pytest.log"
for v in "" "SKYJO_NO_FUSED_MODEL=1"; do
  env $v timeout -k 10 200 python3 tools/bench_cfg5.py 65536 64 8 > "$out/cfg5_${v:-fused}.json" 2> "$out/cfg5.err"; echo "cfg5 [$v] rc=$?"
  python3 -c "
import json
d=json.load(open('$out/cfg5_${v:-fused}.json'))
for k,x in d.items(): print(k, round(x['value']/1e9,3), 'G steps/s', round(x['ms_per_iteration']*1e3,2), 'us/iter  net', round(x['dominant_kernel_ms']*1e3,2), 'step', round(x['step_kernel_ms']*1e3,2))
"
done
