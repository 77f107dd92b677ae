#!/bin/bash
set -u
root=$PWD; out=$root/gpurun_out/r4c6; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_LEVEL[A-Z_]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u | tr '\n' ' ' > "$out/counters.txt"
cat "$out/counters.txt"; echo
for mode in "inline:SKYJO_OVERLAP=0" "merged:SKYJO_MERGED=1"; do
  tag=${mode%%:*}; envs=${mode#*:}
  for set in "ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "w SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    set -- $set; st=$1; shift
    env $envs timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_${tag}_$st" -- python3 "$root/bench.py" --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-other-configs > "$out/pmc_${tag}_$st.json" 2> "$out/pmc_${tag}_$st.err"; echo "pmc $tag $st rc=$?"
    f=$(find "$out/pmc_${tag}_$st" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:36]
    if not (k.startswith("void k_step") or k.startswith("void k_deal") or k.startswith("void k_cycle")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    print(k, "dispatches", len(n[k]), {c: round(v / len(n[k]) / 1e6, 3) for c, v in sorted(agg[k].items())}, "(millions per dispatch)")
PY
  done
done
cd "$root"; find "$out" -type f ! -name "*.json" ! -name "*.err" ! -name "*.txt" -delete
