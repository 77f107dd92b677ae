mkdir -p gpurun_out/exp7
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/exp7/pytest.log 2>&1 || { tail -30 gpurun_out/exp7/pytest.log; exit 1; }
tail -2 gpurun_out/exp7/pytest.log
python tools/ab.py build_exp/base.so build_exp/trust.so 2>&1 | tee gpurun_out/exp7/ab.txt
