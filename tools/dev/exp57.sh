for n in 4096:2 16384:3 32768:3 40960:3 49152:3; do
  B=${n%%:*}; N=${n##*:}
  for ov in 0 1; do
    SKYJO_OVERLAP=$ov python bench.py --steps 100 --warmup 10 --no-cpu-baseline --num-envs $B --num-players $N > gpurun_out/exp57.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/exp57.json").read().strip().splitlines()[-1])
print("B=$B N=$N overlap=$ov  %.3e steps/s  it/step %d  k_step %.1f k_deal %.1f  waits %d" % (d["value"], d["config"]["iterations_per_step"], d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["deal_kernel_avg_ms"]*1e3, d["waits"]))
PY
  done
done
