mkdir -p gpurun_out/exp60
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp60/pytest.log 2>&1 || { tail -40 gpurun_out/exp60/pytest.log; exit 1; }
tail -2 gpurun_out/exp60/pytest.log
for n in 4096:2 32768:3 49152:3; do
  B=${n%%:*}; N=${n##*:}
  for pp in 0 1; do
    SKYJO_PIPELINED=$pp python bench.py --steps 100 --warmup 10 --no-cpu-baseline --num-envs $B --num-players $N > gpurun_out/exp60.json 2>gpurun_out/exp60.err || tail -3 gpurun_out/exp60.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/exp60.json").read().strip().splitlines()[-1])
print("B=$B N=$N piped=$pp  %.3e steps/s  it/step %d  wall %.1f k_step %.1f k_deal %.1f  waits %d episodes %d" % (d["value"], d["config"]["iterations_per_step"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["deal_kernel_avg_ms"]*1e3, d["waits"], d["episodes"]))
PY
  done
done
