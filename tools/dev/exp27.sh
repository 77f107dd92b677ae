mkdir -p gpurun_out/exp27
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_views.py -m gpu -x -q > gpurun_out/exp27/pytest.log 2>&1 || { tail -40 gpurun_out/exp27/pytest.log; exit 1; }
tail -2 gpurun_out/exp27/pytest.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --direct-obs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('direct', d['value'], d['roofline']['avg_launch_ms'])"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['roofline']['avg_launch_ms'])"
python -c "import __graft_entry__ as g; g.smoke()"
