set -x
mkdir -p gpurun_out/exp1
for ov in 0 1; do
  SKYJO_OVERLAP=$ov python bench.py --num-envs 32768 --steps 8000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b32768_ov$ov.json 2>gpurun_out/exp1/err.log
  SKYJO_OVERLAP=$ov python bench.py --num-envs 4096 --num-players 2 --steps 8000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b4096n2_ov$ov.json 2>>gpurun_out/exp1/err.log
  SKYJO_OVERLAP=$ov python bench.py --num-envs 65536 --steps 8000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b65536_ov$ov.json 2>>gpurun_out/exp1/err.log
  SKYJO_OVERLAP=$ov python bench.py --num-envs 16384 --steps 8000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b16384_ov$ov.json 2>>gpurun_out/exp1/err.log
done
python bench.py --num-envs 131072 --rng philox --steps 4000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b131072_philox.json 2>>gpurun_out/exp1/err.log
python bench.py --num-envs 65536 --rng philox --steps 4000 --warmup 800 --no-cpu-baseline > gpurun_out/exp1/b65536_philox.json 2>>gpurun_out/exp1/err.log
tail -n 3 gpurun_out/exp1/*.json
