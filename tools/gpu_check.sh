#!/bin/bash
# usage: bash tools/gpu_check.sh <tag>   - GPU parity suite + a short bench, outputs under gpurun_out/<tag>/
tag=${1:-check}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/$tag/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rc=$?
cat gpurun_out/$tag/bench.json
tail -n 3 gpurun_out/$tag/bench.err
exit $rc
