mkdir -p gpurun_out/exp12
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/exp12/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/exp12/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/exp12/bench_driver.json 2>gpurun_out/exp12/err.txt && python -c "
import json; d=json.load(open('gpurun_out/exp12/bench_driver.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['config']['iterations_per_step'], d['roofline']['frac'], d['roofline_path']['frac_wall'])"
python tools/bench_cfg5.py > gpurun_out/exp12/cfg5.json 2>>gpurun_out/exp12/err.txt; cat gpurun_out/exp12/cfg5.json
