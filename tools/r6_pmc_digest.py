"""Digest gpurun_out/r6pmc/ (tools/r6_pmc.sh) into profiles/r6_cfg5_pmc.json and profiles/r6_pmc_philox.json: per kernel and
pass, the average per dispatch of every counter.   python tools/r6_pmc_digest.py"""
import collections, csv, glob, json, os, sys
src = "gpurun_out/r6pmc"
def short(name):
    return name.split("(")[0].replace("void ", "").strip()
def digest(prefix, keep):
    res = {}
    for d in sorted(glob.glob(os.path.join(src, prefix + "_*"))):
        if not os.path.isdir(d):
            continue
        tag = os.path.basename(d)[len(prefix) + 1:]
        fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(fs[-1])):
            k = short(r["Kernel_Name"])
            if not keep(k):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        res[tag] = {k: dict({"dispatches": len(n[k])}, **{c: v / len(n[k]) for c, v in agg[k].items()}) for k in agg}
    return res
if __name__ == "__main__":
    c5 = digest("cfg5", lambda k: k.startswith("k_mlp") or k.startswith("k_step") or k.startswith("k_net"))
    if c5:
        json.dump(c5, open("profiles/r6_cfg5_pmc.json", "w"), indent=1)
    ph = digest("philox", lambda k: k.startswith("k_cycle"))
    if ph:
        json.dump(ph, open("profiles/r6_pmc_philox.json", "w"), indent=1)
    print(json.dumps({"cfg5": c5, "philox": ph}, indent=1))
