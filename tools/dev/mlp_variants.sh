for v in base gt2 ug4 ug1gt2; do
SKYJO_LIB=$PWD/build_exp/mlp_$v.so timeout -k 10 300 python tools/bench_cfg5.py 65536 200 mfma > gpurun_out/mlpv.json 2>gpurun_out/mlpv.err || tail -3 gpurun_out/mlpv.err
python - <<PY
import json
d=json.loads(open("gpurun_out/mlpv.json").read().strip().splitlines()[-1])
print("$v", {k:(round(v['env_steps_per_s']/1e9,3), round(v['us_per_iteration'],1)) for k,v in d.items() if isinstance(v,dict)})
PY
done
