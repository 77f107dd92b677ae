out=$PWD/gpurun_out/exp59; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --num-envs 32768 > $out/bench.json 2> $out/err.txt
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
sel=rows[len(rows)//2:len(rows)//2+16]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%-12s start %8.1f end %8.1f dur %6.1f stream/queue %s" % (r["Kernel_Name"][:12].replace("void ",""), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id","?")))
PY
