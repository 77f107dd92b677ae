mkdir -p gpurun_out/exp26
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard_snapshot.py tests/test_gpu_views.py -m gpu -x -q > gpurun_out/exp26/pytest.log 2>&1 || { tail -40 gpurun_out/exp26/pytest.log; exit 1; }
tail -2 gpurun_out/exp26/pytest.log
python tools/dev/ab.py build_exp/head.so build_exp/gm.so
bash tools/gpu_pmc.sh fetch2 "FETCH_SIZE" 2>&1 | grep -A2 "k_step\|k_deal"
bash tools/gpu_pmc.sh write2 "WRITE_SIZE" 2>&1 | grep -A2 "k_step\|k_deal"
