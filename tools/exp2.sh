mkdir -p gpurun_out/exp2
python tools/ab.py build_exp/base.so build_exp/nodma.so build_exp/nodrain.so > gpurun_out/exp2/ab.txt 2>gpurun_out/exp2/err.txt
cat gpurun_out/exp2/ab.txt
SKYJO_LIB=$PWD/build_exp/base.so python tools/first60.py > gpurun_out/exp2/first60.txt 2>>gpurun_out/exp2/err.txt
cat gpurun_out/exp2/first60.txt
SKYJO_LIB=$PWD/build_exp/stamps.so python tools/stamps.py > gpurun_out/exp2/stamps.txt 2>>gpurun_out/exp2/err.txt
cat gpurun_out/exp2/stamps.txt
