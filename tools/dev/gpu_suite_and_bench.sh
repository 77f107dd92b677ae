#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/suite; mkdir -p "$out"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?"
tail -n 8 "$out/pytest.log"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_path']['frac_wall'], d['roofline_path']['traffic_over_algorithmic'])
print({k:(round(v.get('value',0)/1e9,2), v.get('dealing')) for k,v in d['other_configs'].items() if isinstance(v,dict)})
"
