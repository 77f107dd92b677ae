for f in 0 1; do
SKYJO_FUSED_SCAN=$f python bench.py --no-cpu-baseline > gpurun_out/exp52_$f.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/exp52_$f.json").read().strip().splitlines()[-1])
print("fused=$f value %.4e wall %.2f us kernels %s episodes %d waits %d" % (d["value"], d["ms_per_step"]*1e3, {k: round(v*1e3,1) for k,v in d["roofline_path"]["kernel_ms_per_step"].items()}, d["episodes"], d["waits"]))
PY
done
