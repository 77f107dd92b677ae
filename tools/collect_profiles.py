"""Digest gpurun_out/final/ (written by tools/refresh_profiles.sh on the GPU box) into profiles/<round>_*.  Run from the repo root:
    python tools/collect_profiles.py [r4]"""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.getcwd())
src, dst = "gpurun_out/final", "profiles"
R = sys.argv[1] if len(sys.argv) > 1 else "r6"
def find(pat):  # newest match: gpurun merges every refresh into the same local directory
    r = sorted(glob.glob(os.path.join(src, pat), recursive=True), key=os.path.getmtime)
    return r[-1] if r else None
def short(name):
    return name.split("(")[0].replace("void ", "").strip()
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, R + "_bench.json"), "w"), indent=1)
shutil.copy(find("trace/**/*kernel_stats.csv"), os.path.join(dst, R + "_kernel_stats.csv"))
tb = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])
json.dump(tb, open(os.path.join(dst, R + "_bench_under_rocprof.json"), "w"), indent=1)
# kernel trace digest: every dispatch's duration; the last 32 full k_step launches are bench.py's roofline leg
rows = list(csv.DictReader(open(find("trace/**/*kernel_trace.csv"))))
per = collections.defaultdict(list)
for r in rows:
    per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
digest = []
for k, v in per.items():
    if not (k.startswith("k_step") or k.startswith("k_deal") or k.startswith("k_cycle") or k in ("k_scan", "k_publish")):
        continue
    v.sort()
    d = [x[1] for x in v]
    digest.append({"kernel": k, "calls": len(d), "avg_us": sum(d) / len(d), "last32_avg_us": sum(d[-32:]) / len(d[-32:]),
                   "min_us": min(d), "max_us": max(d)})
# where the long launches are: the first launches after seeding, and the launches behind a pause of the launch stream
for k, v in per.items():
    if not k.startswith("k_cycle"):
        continue
    v.sort()
    d = [x[1] for x in v]
    full = sorted(x for x in d if x > 0.5 * max(d))
    med = full[len(full) // 2]
    slow = [(i, x) for i, x in enumerate(d) if x > 1.05 * med]
    allk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    ends = {}
    for i in range(1, len(allk)):
        ends[allk[i][0]] = (allk[i][0] - allk[i - 1][1]) / 1e3   # idle time in front of every dispatch, us
    pauses = [i for i, x in enumerate(v) if ends.get(x[0], 0.0) > 40.0]
    digest.append({"kernel": k, "launch_time_tail": {
        "median_us": med, "launches_over_105pct_of_median": len(slow), "of": len(d),
        "in_the_first_16_after_seeding": sum(1 for i, _ in slow if i < 16), "max_us_in_the_first_16": max([x for i, x in slow if i < 16], default=None),
        "within_12_launches_behind_a_pause_over_40us": sum(1 for i, _ in slow if i >= 16 and any(0 <= i - p0 <= 12 for p0 in pauses)),
        "max_us_behind_a_pause": max([x for i, x in slow if i >= 16 and any(0 <= i - p0 <= 12 for p0 in pauses)], default=None),
        "elsewhere": sum(1 for i, _ in slow if i >= 16 and not any(0 <= i - p0 <= 12 for p0 in pauses)),
        "pauses_at_launch": pauses,
        "note": "launch index counts this kernel's dispatches from the seeding on (100 set-up launches first); a pause = more than 40 us "
                "without any kernel in front of the dispatch (the host reading counters / gathering statistics between two timed blocks)"}})
digest.append({"note": "same run, bench.py's own HIP-event figure for the last 32 k_step launches",
               "avg_launch_ms": tb["roofline"]["avg_launch_ms"], "deal_kernel_avg_ms": tb["roofline"]["deal_kernel_avg_ms"]})
json.dump(digest, open(os.path.join(dst, R + "_kernel_trace_digest.json"), "w"), indent=1)
# PMC passes
pmc = {}
for tag in ("fetch", "write", "sq", "sq2", "ea", "l2", "issue"):
    f = find("pmc_%s/**/*counter_collection.csv" % tag)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    pmc[tag] = {k: dict({"dispatches": len(n[k])}, **{c: v / len(n[k]) for c, v in agg[k].items()}) for k in agg}
f = find("deckfast_pmc_sq/**/*counter_collection.csv")  # round 6, experiment 11: the -DSK_EXP_DECK_FAST build's counters beside the shipped kernel's
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    pmc["deckfast_sq"] = {k: dict({"dispatches": len(n[k])}, **{c: v / len(n[k]) for c, v in agg[k].items()}) for k in agg}
    try:
        pmc["deckfast_sq"]["bench_value"] = json.loads(open(os.path.join(src, "deckfast_pmc_sq.json")).read().strip().splitlines()[-1])["value"]
        pmc["sq"]["bench_value"] = json.loads(open(os.path.join(src, "pmc_sq.json")).read().strip().splitlines()[-1])["value"]
    except Exception as e:
        print("deckfast:", e)
json.dump(pmc, open(os.path.join(dst, R + "_pmc_per_dispatch.json"), "w"), indent=1)
merged = any(k.startswith("k_cycle") for k in pmc["fetch"])
ks = [k for k in pmc["fetch"] if k.startswith("k_cycle" if merged else "k_step")][0]
fk, wk = pmc["fetch"][ks]["FETCH_SIZE"], pmc["write"][ks]["WRITE_SIZE"]
import bench as bench_py  # (module level only defines functions and constants)
traffic = {
    "kernel_source_sha256": bench_py.kernel_source_sha256(),  # bench.py reports this traffic only for these very sources ...
    "kernel_sources": list(bench_py.KERNEL_SOURCES),
    "shape": tb["roofline"]["launch_shape"],                  # ... and this very launch shape
    "kernel": ks,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/refresh_profiles.sh), bench.py --steps 30 "
              "--warmup 10, per dispatch of the dominant kernel (one dealing cycle: k_cycle holds the step AND the dealing wavefronts)",
    "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
    "note": "gfx950: FETCH_SIZE counts half of a wide coalesced 16 B/lane stream (MI355X_MICROARCH.md, HBM) -> doubled; "
            "WRITE_SIZE is exact for 16 B/lane stores",
    "k_step_bytes_per_launch": int(round((2 * fk + wk) * 1024)),
    "uncorrected_bytes_per_launch": int(round((fk + wk) * 1024)),
}
if "ea" in pmc and ks in pmc["ea"]:  # the same kernel at the fabric: read requests by size (nearly all 128-byte lines), write requests (64-byte and smaller)
    e = pmc["ea"][ks]
    rd128, rd, wr64, wr = e["TCC_EA0_RDREQ_128B_sum"], e["TCC_EA0_RDREQ_sum"], e["TCC_EA0_WRREQ_64B_sum"], e["TCC_EA0_WRREQ_sum"]
    traffic["fabric"] = {
        "read_requests": rd, "read_requests_128B": rd128, "write_requests": wr, "write_requests_64B": wr64,
        "read_bytes": int(rd128 * 128 + (rd - rd128) * 64), "write_bytes": int(wr64 * 64 + (wr - wr64) * 32),
        "note": "per dispatch: TCC_EA0_RDREQ[_128B] / TCC_EA0_WRREQ[_64B]; reads that are not 128-byte requests counted as 64 bytes, "
                "writes that are not 64-byte requests as 32 bytes"}
    if not merged:
        kd = [k for k in pmc["ea"] if k.startswith("k_deal")][0]
        e = pmc["ea"][kd]
        traffic["k_deal_fabric"] = {"read_bytes": int(e["TCC_EA0_RDREQ_128B_sum"] * 128 + (e["TCC_EA0_RDREQ_sum"] - e["TCC_EA0_RDREQ_128B_sum"]) * 64),
                                    "write_bytes": int(e["TCC_EA0_WRREQ_64B_sum"] * 64 + (e["TCC_EA0_WRREQ_sum"] - e["TCC_EA0_WRREQ_64B_sum"]) * 32)}
json.dump(traffic, open(os.path.join(dst, R + "_hbm_traffic.json"), "w"), indent=1)
# attribution passes (tools/refresh_profiles.sh): the same launch with counter-based deals / without records
def side_pmc(prefix):
    out = {}
    for tag in ("fetch", "write", "ea"):
        f = find("%s_pmc_%s/**/*counter_collection.csv" % (prefix, tag))
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_cycle") or k.startswith("k_step"):
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in agg:
            out.setdefault(k, {}).update({c: v / len(n[k]) for c, v in agg[k].items()})
            out[k]["dispatches_" + tag] = len(n[k])
    res = {}
    for k, e in out.items():
        d = dict(e)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            d["bytes_per_launch"] = int(round((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024))
            d["read_bytes_per_launch"], d["write_bytes_per_launch"] = int(round(2 * e["FETCH_SIZE"] * 1024)), int(round(e["WRITE_SIZE"] * 1024))
        if "TCC_EA0_RDREQ_sum" in e:
            rd128, rd, wr64, wr = e["TCC_EA0_RDREQ_128B_sum"], e["TCC_EA0_RDREQ_sum"], e["TCC_EA0_WRREQ_64B_sum"], e["TCC_EA0_WRREQ_sum"]
            d["fabric_read_bytes"], d["fabric_write_bytes"] = int(rd128 * 128 + (rd - rd128) * 64), int(wr64 * 64 + (wr - wr64) * 32)
        res[k] = d
    return res
attr = {"note": "per dispatch of the dominant kernel, bench.py's launch shape (tools/refresh_profiles.sh); FETCH_SIZE doubled as in " + R + "_hbm_traffic.json",
        "mt19937_records": {"bytes_per_launch": traffic["k_step_bytes_per_launch"], "read_bytes_per_launch": int(round(2 * fk * 1024)),
                            "write_bytes_per_launch": int(round(wk * 1024)), "fabric": traffic.get("fabric")},
        "philox_records": side_pmc("philox"), "mt19937_no_records": side_pmc("norec")}
json.dump(attr, open(os.path.join(dst, R + "_hbm_traffic_attribution.json"), "w"), indent=1)
# config 5
c1 = os.path.join(src, "cfg1.json")
if os.path.exists(c1):
    try:
        json.dump(json.loads(open(c1).read().strip().splitlines()[-1]), open(os.path.join(dst, R + "_cfg1.json"), "w"), indent=1)
    except Exception as e:
        print("cfg1:", e)
import subprocess
subprocess.run([sys.executable, os.path.join("tools", "r6_pmc_digest.py")], stdout=subprocess.DEVNULL)  # -> profiles/r6_cfg5_pmc.json, r6_pmc_philox.json
c5 = os.path.join(src, "cfg5.json")
if os.path.exists(c5):
    shutil.copy(c5, os.path.join(dst, R + "_cfg5_bench.json"))
    f = find("trace_cfg5/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, R + "_cfg5_kernel_stats.csv"))
print(json.dumps({"bench_value": bench["value"], "roofline": bench["roofline"], "traffic": traffic["k_step_bytes_per_launch"],
                  "digest": digest}, indent=1))
