mkdir -p gpurun_out/exp48
timeout -k 10 900 python -m pytest tests/test_gpu_policy_net.py tests/test_gpu_rollout_buffer.py tests/test_gpu_ppo_handoff.py -m gpu -x -q > gpurun_out/exp48/pytest.log 2>&1 || { tail -40 gpurun_out/exp48/pytest.log; exit 1; }
tail -2 gpurun_out/exp48/pytest.log
timeout -k 10 600 python tools/bench_cfg5.py 65536 300 > gpurun_out/exp48/cfg5.json 2> gpurun_out/exp48/cfg5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/exp48/cfg5.json").read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print(k, "%.3e steps/s %.1f us/it" % (v["env_steps_per_s"], v["us_per_iteration"]), v.get("illegal"))
PY
