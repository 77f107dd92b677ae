mkdir -p gpurun_out/exp10
python tools/ab.py build_exp/trust.so build_exp/trust.so@SKYJO_LDS_PAD=5000 build_exp/trust.so@SKYJO_LDS_PAD=12000 2>&1 | tee gpurun_out/exp10/ab.txt
SKYJO_DEAL_INTERVAL=96 SKYJO_BENCH_CHUNK=96 python bench.py --steps 2000 --warmup 10 --no-cpu-baseline > gpurun_out/exp10/long96.json 2> gpurun_out/exp10/err.txt; python -c "
import json; d=json.load(open('gpurun_out/exp10/long96.json')); print('interval 96 long run', d['value'], 'waits', d['waits'], 'episodes', d['episodes'])" 2>&1 | tail -2; tail -3 gpurun_out/exp10/err.txt
