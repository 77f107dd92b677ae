mkdir -p gpurun_out/exp50
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp50/pytest.log 2>&1 || { tail -40 gpurun_out/exp50/pytest.log; exit 1; }
tail -2 gpurun_out/exp50/pytest.log
timeout -k 10 600 python tools/dev/ab.py build_exp/fused.so@SKYJO_FUSED_SCAN=0 build_exp/fused.so@SKYJO_FUSED_SCAN=1
