#!/bin/bash
# bench.py over the other BASELINE configurations (numbers of DESIGN.md section 6); outputs under gpurun_out/configs/
mkdir -p gpurun_out/configs
run() { name=$1; shift; python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/configs/$name.json 2> gpurun_out/configs/$name.err || tail -2 gpurun_out/configs/$name.err; }
run cfg3_default
run cfg2_4096x2 --num-envs 4096 --num-players 2
run cfg4_shard_32768x3 --num-envs 32768
run b16384x3 --num-envs 16384
run b65536x2 --num-players 2
run b65536x4 --num-players 4
run b65536x3_direct --direct-obs
run b65536x3_philox --rng philox
run b131072x3 --num-envs 131072
run b262144x3 --num-envs 262144
run b65536x3_actions_array --actions-array
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %.3e steps/s  k_step %.1f us/launch of %d it  k_deal %.1f us  dealing %s  eplen %.1f" % (
            os.path.basename(f)[:-5], d["value"], d["roofline"]["avg_launch_ms"] * 1e3, d["config"]["iterations_per_step"],
            d["roofline"]["deal_kernel_avg_ms"] * 1e3, d["config"]["dealing"], d["mean_episode_len"]))
    except Exception as e:
        print(f, "ERR", e)
PY
