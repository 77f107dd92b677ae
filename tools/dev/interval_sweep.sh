export SKYJO_BENCH_ALLOW_WAITS=1
run() { B=$1; N=$2; pp=$3; iv=$4
  SKYJO_PIPELINED=$pp SKYJO_DEAL_INTERVAL=$iv python bench.py --steps 100 --warmup 10 --no-cpu-baseline --num-envs $B --num-players $N > gpurun_out/exp61.json 2>gpurun_out/exp61.err || tail -3 gpurun_out/exp61.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/exp61.json").read().strip().splitlines()[-1])
print("B=$B N=$N piped=$pp iv=$iv  %.3e steps/s  it/step %d  wall %.1f k_step %.1f k_deal %.1f  waits %d" % (d["value"], d["config"]["iterations_per_step"], d["ms_per_step"]*1e3, d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["deal_kernel_avg_ms"]*1e3, d["waits"]))
PY
}
for iv in 64 72 80; do run 32768 3 1 $iv; done
run 32768 3 0 64
for iv in 48 56 64; do run 4096 2 1 $iv; done
run 4096 2 0 48
for iv in 64 72 80; do run 49152 3 1 $iv; done
