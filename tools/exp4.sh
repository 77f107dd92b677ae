mkdir -p gpurun_out/exp4
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/exp4/pytest.log 2>&1; rc=$?
tail -25 gpurun_out/exp4/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/exp4/bench_driver.json 2>gpurun_out/exp4/err.txt && cat gpurun_out/exp4/bench_driver.json
python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/exp4/bench_2ranks_shared.json 2>>gpurun_out/exp4/err.txt; echo "rc=$?"; cat gpurun_out/exp4/bench_2ranks_shared.json; tail -5 gpurun_out/exp4/err.txt
