mkdir -p gpurun_out/exp6
python tools/ab.py build_exp/base.so build_exp/base.so@SKYJO_DEAL_INTERVAL=96@SKYJO_BENCH_CHUNK=96 build_exp/base.so@SKYJO_DEAL_INTERVAL=88@SKYJO_BENCH_CHUNK=88 2>&1 | tee gpurun_out/exp6/ab.txt
