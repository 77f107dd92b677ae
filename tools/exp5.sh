mkdir -p gpurun_out/exp5
for v in base noscore; do echo $v; SKYJO_LIB=$PWD/build_exp/$v.so python tools/first60.py 2>/dev/null; done | tee gpurun_out/exp5/first60.txt
