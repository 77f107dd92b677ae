mkdir -p gpurun_out/exp8
python tools/ab.py build_exp/trust.so build_exp/wrap.so build_exp/cached.so build_exp/wrapcached.so 2>&1 | tee gpurun_out/exp8/ab.txt
