#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/refresh_profiles.sh
# One pass of everything profiles/ is digested from (tools/collect_profiles.py reads gpurun_out/final/ afterwards):
#   bench.json            python bench.py --steps 20 --warmup 5 (the line as the driver asks for it, every BASELINE config in other_configs)
#   trace/                rocprofv3 --kernel-trace --stats      (short bench run)
#   pmc_fetch|write|sq|sq2|ea|l2/   rocprofv3 --kernel-trace --pmc ...    (separate passes, nothing else traced)
set -u
root=$PWD
out=$root/gpurun_out/final
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --steps 100 --warmup 10 --blocks 2 --no-cpu-baseline --no-other-configs > "$out/trace_bench.json" 2> "$out/trace.err"; echo "trace rc=$?"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "l2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_MISS_sum" "issue SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  set -- $pass; tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$tag" -- python3 "$root/bench.py" --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/pmc_$tag.json" 2> "$out/pmc_$tag.err"; echo "pmc $tag rc=$?"
done
# the same launch with counter-based deals (no generator state at all): FETCH_SIZE / WRITE_SIZE / fabric requests - what is left is
# records + tiles + bank; the difference to the passes above is what the numpy-exact generator moves (EXPERIMENTS.md round 5)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  set -- $pass; tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/philox_pmc_$tag" -- python3 "$root/bench.py" --rng philox --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/philox_pmc_$tag.json" 2> "$out/philox_pmc_$tag.err"; echo "philox pmc $tag rc=$?"
done
# ... and without records (the step wavefronts store nothing per iteration): tiles + bank + generator
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/norec_pmc_$tag" -- python3 "$root/bench.py" --no-records --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/norec_pmc_$tag.json" 2> "$out/norec_pmc_$tag.err"; echo "norec pmc $tag rc=$?"
done
# round 6, VERDICT r5 item 4: the same launch with the dealing walk's completion handling hoisted out of the per-draw path
# (-DSK_EXP_DECK_FAST build, parity green, SLOWER: EXPERIMENTS.md round 6) - its instruction counters beside the shipped kernel's
if [ -f "$root/build_exp/libskyjo_vec_deckfast.so" ]; then
  SKYJO_LIB="$root/build_exp/libskyjo_vec_deckfast.so" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/deckfast_pmc_sq" -- python3 "$root/bench.py" --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/deckfast_pmc_sq.json" 2> "$out/deckfast_pmc_sq.err"; echo "deckfast pmc sq rc=$?"
fi
# round 6: counter passes on the policy net's kernels and the SQ passes of the counter-based-deals launch shape (tools/r6_pmc.sh -> gpurun_out/r6pmc/)
cd "$root"
bash tools/r6_pmc.sh cfg5 philox
# config 1 (one game driven from Python through the reference's own loops)
python3 tools/bench_cfg1.py > "$out/cfg1.json" 2> "$out/cfg1.err"; echo "cfg1 rc=$?"
# config 5 (65 536 x 4 players, the action-mask model on the matrix cores picks every action): bench line + kernel stats
cd "$root"
python3 tools/bench_cfg5.py 65536 64 8 > "$out/cfg5.json" 2> "$out/cfg5.err"; echo "cfg5 rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_cfg5" -- python3 "$root/tools/bench_cfg5.py" 65536 64 4 > "$out/cfg5_trace.json" 2> "$out/cfg5_trace.err"; echo "cfg5 trace rc=$?"
cd "$root"
# keep the merge-back small: only the csv summaries
find "$out" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh "$out"
