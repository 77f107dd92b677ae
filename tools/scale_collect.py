"""Digest one scaling point written by tools/scale_run.sh (gpurun_out/scale/N<N>/rank*/ - one rocprofv3 kernel trace per rank) into
gpurun_out/scale/N<N>/scale_point.json, or - `--table` - all points found into profiles/<round>_scale.json:

    python3 tools/scale_collect.py N            per rank: calls / average us of the dominant kernel, achieved GB/s = SURVEY 8(d)
                                                algorithmic bytes per launch (bench.py's formula, taken from rank 0's line) / that
                                                average, its fraction of the 8 TB/s peak; the node: bench.py's value and RCCL fields
    python3 tools/scale_collect.py --table [rK] value and per-rank GB/s at 1 / 2 / 4 / 8 GPUs, efficiency left to the reader (the
                                                driver computes it from its own runs)
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "gpurun_out", "scale")
PEAK_GBS = 8000.0


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def rank_stats(rank_dir, timed):
    """(kernel, dispatches in the trace, average us over the LAST `timed` of them) of the dominant kernel (k_cycle or k_step) from the
    rank's *kernel_trace.csv - the settle and warm-up launches (other dealing intervals, other iteration counts) stay out of it, as in
    bench.py's own roofline figure; the *kernel_stats.csv average over all calls is the fall-back."""
    hits = sorted(glob.glob(os.path.join(rank_dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if hits:
        per = {}
        for r in csv.DictReader(open(hits[-1])):
            k = short(r["Kernel_Name"])
            if k.startswith("k_cycle") or k.startswith("k_step"):
                per.setdefault(k, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        if per:
            k = max(per, key=lambda n: sum(d for _, d in per[n]))
            d = [x[1] for x in sorted(per[k])]
            last = d[-max(1, int(timed)):]
            return k, len(d), sum(last) / len(last)
    hits = sorted(glob.glob(os.path.join(rank_dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not hits:
        return None
    best = None
    for r in csv.DictReader(open(hits[-1])):
        k = short(r["Name"])
        if k.startswith("k_cycle") or k.startswith("k_step"):
            tot = float(r["TotalDurationNs"])
            if best is None or tot > best[3]:
                best = (k, int(r["Calls"]), float(r["AverageNs"]) / 1e3, tot)
    return best[:3] if best else None


def point(n):
    d = os.path.join(BASE, "N%d" % n)
    line = json.loads(open(os.path.join(d, "rank0.json")).read().strip().splitlines()[-1])
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    ranks = []
    for r in range(n):
        st = rank_stats(os.path.join(d, "rank%d" % r), line["roofline"].get("launches_timed", 32))
        if st is None:
            ranks.append({"rank": r, "error": "no kernel_stats.csv"})
            continue
        k, calls, avg_us = st
        gbs = alg / (avg_us * 1e-6) / 1e9
        ranks.append({"rank": r, "kernel": k, "calls": calls, "avg_us": avg_us, "achieved_GBs": gbs, "frac_of_peak": gbs / PEAK_GBS})
    out = {"n_gpus": n, "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "scaling": line["scaling"],
           "collective": line["config"]["collective"], "algorithmic_bytes_per_launch": alg, "peak_GBs": PEAK_GBS, "ranks": ranks,
           "bench_achieved_GBs": line["roofline"].get("achieved"), "launches_timed": line["roofline"].get("launches_timed", 32),
           "note": "per rank: rocprofv3 --kernel-trace --stats of that rank's own process (tools/scale_run.sh), average over the last "
                   "`launches_timed` dispatches of the dominant kernel; value / bench_achieved_GBs: bench.py's own line (rank 0, HIP events)"}
    json.dump(out, open(os.path.join(d, "scale_point.json"), "w"), indent=1)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--table":
        tag = sys.argv[2] if len(sys.argv) > 2 else "r6"
        pts = []
        for d in sorted(glob.glob(os.path.join(BASE, "N*")), key=lambda p: int(os.path.basename(p)[1:])):
            f = os.path.join(d, "scale_point.json")
            if os.path.exists(f):
                pts.append(json.load(open(f)))
        json.dump(pts, open(os.path.join(ROOT, "profiles", tag + "_scale.json"), "w"), indent=1)
        for p in pts:
            gbs = [r.get("achieved_GBs", 0.0) for r in p["ranks"]]
            print("N=%d value %.3e %s  per-rank GB/s min %.0f max %.0f (%.2f of peak)" % (p["n_gpus"], p["value"], p["unit"], min(gbs), max(gbs), min(gbs) / PEAK_GBS))
    else:
        p = point(int(sys.argv[1]))
        print(json.dumps(p, indent=1))
