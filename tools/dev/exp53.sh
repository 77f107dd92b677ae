mkdir -p gpurun_out/exp53
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/exp53/pytest.log 2>&1 || { tail -40 gpurun_out/exp53/pytest.log; exit 1; }
tail -2 gpurun_out/exp53/pytest.log
timeout -k 10 600 python tools/dev/ab.py build_exp/final3.so build_exp/early.so
