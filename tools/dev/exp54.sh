mkdir -p gpurun_out/exp54
SKYJO_LIB=$PWD/build_exp/early3.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or cadence or headline" > gpurun_out/exp54/pytest.log 2>&1 || { tail -40 gpurun_out/exp54/pytest.log; exit 1; }
tail -2 gpurun_out/exp54/pytest.log
timeout -k 10 600 python tools/dev/ab.py build_exp/final3.so build_exp/early3.so
