for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/exp65.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/exp65.json").read().strip().splitlines()[-1])
print("run $i  %.3e  k_step %.1f k_deal %.1f" % (d["value"], d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["deal_kernel_avg_ms"]*1e3))
PY
done
