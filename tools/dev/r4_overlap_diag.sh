#!/bin/bash
# Round 4, VERDICT item 1a: where does the time go when k_deal runs beside k_step on a full chip?
# usage (GPU box, repo root): bash tools/dev/r4_overlap_diag.sh    -> gpurun_out/r4diag/
set -u
root=$PWD
out=$root/gpurun_out/r4diag
mkdir -p "$out"
export TMPDIR=/tmp
# 1. section stamps of a k_step wavefront, in line vs beside k_deal (coarse and fine sections)
for ov in 0 1; do
  SKYJO_OVERLAP=$ov SKYJO_LIB=$root/build_exp/lib_stamps.so timeout -k 10 200 python3 tools/dev/stamps.py 65536 > "$out/stamps_ov$ov.txt" 2>&1; echo "stamps ov=$ov rc=$?"
  FINE=1 SKYJO_OVERLAP=$ov SKYJO_LIB=$root/build_exp/lib_stamps_fine.so timeout -k 10 200 python3 tools/dev/stamps.py 65536 > "$out/stamps_fine_ov$ov.txt" 2>&1; echo "stamps fine ov=$ov rc=$?"
done
# 2. headline A/B: in line / beside, MT19937 / Philox
ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3" timeout -k 10 400 python3 tools/dev/env_ab.py "" "SKYJO_OVERLAP=1" > "$out/ab_mt.txt" 2>&1; echo "ab mt rc=$?"
ROUNDS=2 BENCH_ARGS="--steps 40 --warmup 10 --blocks 3 --rng philox" timeout -k 10 400 python3 tools/dev/env_ab.py "" "SKYJO_OVERLAP=1" > "$out/ab_philox.txt" 2>&1; echo "ab philox rc=$?"
# 3. does a PMC pass keep the two kernels side by side at all?  (kernel trace of the overlapped pair with and without --pmc)
cd /tmp
SKYJO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_ov" -- python3 "$root/bench.py" --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/trace_ov.json" 2> "$out/trace_ov.err"; echo "trace ov rc=$?"
SKYJO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_ov_sq" -- python3 "$root/bench.py" --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/pmc_ov_sq.json" 2> "$out/pmc_ov_sq.err"; echo "pmc ov sq rc=$?"
SKYJO_OVERLAP=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d "$out/pmc_ov_l2" -- python3 "$root/bench.py" --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/pmc_ov_l2.json" 2> "$out/pmc_ov_l2.err"; echo "pmc ov l2 rc=$?"
cd "$root"
find "$out" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" ! -name "*.txt" -delete
du -sh "$out"
