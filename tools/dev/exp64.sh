mkdir -p gpurun_out/exp64
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp64/pytest.log 2>&1 || { tail -40 gpurun_out/exp64/pytest.log; exit 1; }
tail -2 gpurun_out/exp64/pytest.log
export SKYJO_BENCH_ALLOW_WAITS=1
run() { B=$1; N=$2; ov=$3
  SKYJO_OVERLAP=$ov python bench.py --steps 200 --warmup 10 --no-cpu-baseline --num-envs $B --num-players $N > gpurun_out/exp61.json 2>gpurun_out/exp61.err || tail -3 gpurun_out/exp61.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/exp61.json").read().strip().splitlines()[-1])
print("B=$B N=$N overlap=$ov  %.3e steps/s  it/step %d  wall %.1f k_step %.1f k_deal %.1f  waits %d" % (d["value"], d["config"]["iterations_per_step"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["deal_kernel_avg_ms"]*1e3, d["waits"]))
PY
}
run 57344 3 1; run 57344 3 0; run 65536 3 1; run 65536 3 0; run 53248 3 1; run 53248 3 0
