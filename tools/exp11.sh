mkdir -p gpurun_out/exp11
for iv in 88 92; do
SKYJO_DEAL_INTERVAL=$iv SKYJO_BENCH_CHUNK=$iv python bench.py --steps 4000 --warmup 10 --no-cpu-baseline > gpurun_out/exp11/long$iv.json 2> gpurun_out/exp11/err$iv.txt; tail -2 gpurun_out/exp11/err$iv.txt; python -c "
import json; d=json.load(open('gpurun_out/exp11/long$iv.json')); print('interval $iv long run', d['value'], 'waits', d['waits'], 'episodes', d['episodes'])" 2>&1 | tail -1
done
