mkdir -p gpurun_out/exp34
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/exp34/pytest.log 2>&1 || { tail -40 gpurun_out/exp34/pytest.log; exit 1; }
tail -2 gpurun_out/exp34/pytest.log
python tools/dev/ab.py build_exp/head.so build_exp/nobr.so build_exp/peel.so
