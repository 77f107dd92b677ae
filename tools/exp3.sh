mkdir -p gpurun_out/exp3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cadence or rollout or headline" > gpurun_out/exp3/pytest.log 2>&1 || { tail -30 gpurun_out/exp3/pytest.log; exit 1; }
tail -3 gpurun_out/exp3/pytest.log
python tools/ab.py build_exp/base.so build_exp/deck8.so build_exp/deck8.so@SKYJO_OVERLAP=1 build_exp/deck8_prio.so@SKYJO_OVERLAP=1 > gpurun_out/exp3/ab.txt 2>gpurun_out/exp3/err.txt
cat gpurun_out/exp3/ab.txt
python tools/ab.py --num-envs=32768 build_exp/deck8.so build_exp/deck8.so@SKYJO_OVERLAP=1 > gpurun_out/exp3/ab32k.txt 2>>gpurun_out/exp3/err.txt
cat gpurun_out/exp3/ab32k.txt
