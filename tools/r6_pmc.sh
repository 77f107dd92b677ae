#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/r6_pmc.sh [cfg5] [philox]
# Round-6 counter passes: rocprofv3 --kernel-trace --pmc only, the program itself after `--`.
#   cfg5:   the two policy-net kernels of config 5 (tools/bench_cfg5.py, both precisions) -> gpurun_out/r6pmc/cfg5_<pass>/
#   philox: the headline launch shape with counter-based deals (bench.py --rng philox)    -> gpurun_out/r6pmc/philox_<pass>/
# tools/r6_pmc_digest.py turns the csv files into profiles/r6_cfg5_pmc.json / profiles/r6_pmc_philox.json.
set -u
root=$PWD
out=$root/gpurun_out/r6pmc
mkdir -p "$out"
export TMPDIR=/tmp
what=${*:-cfg5 philox}
cd /tmp
for w in $what; do
  if [ "$w" = cfg5 ]; then
    for pass in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
                "b SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
                "c SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
                "d SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
      set -- $pass; tag=$1; shift
      rm -rf "$out/cfg5_$tag"
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/cfg5_$tag" -- python3 "$root/tools/bench_cfg5.py" 65536 64 2 > "$out/cfg5_$tag.json" 2> "$out/cfg5_$tag.err"; echo "cfg5 pmc $tag rc=$?"
    done
  fi
  if [ "$w" = philox ]; then
    for pass in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
                "sq2 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
                "issue SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
      set -- $pass; tag=$1; shift
      rm -rf "$out/philox_$tag"
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/philox_$tag" -- python3 "$root/bench.py" --rng philox --steps 30 --warmup 10 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/philox_$tag.json" 2> "$out/philox_$tag.err"; echo "philox pmc $tag rc=$?"
    done
  fi
done
cd "$root"
find "$out" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
du -sh "$out"
