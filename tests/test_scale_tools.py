"""tools/scale_collect.py (the digest behind `bash tools/scale_run.sh N`, staged for an 8-GPU node: SURVEY 8e asks for rocprof
achieved-GB/s at 1 / 2 / 4 / 8 GPUs) on a synthetic scaling point: two ranks' kernel-stats CSVs in rocprofv3's format and rank 0's
bench line - per rank the dominant kernel's calls / average time / achieved GB/s against the 8 TB/s peak, the node's value from the line."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("scale_collect", os.path.join(ROOT, "tools", "scale_collect.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scale_point_digest(tmp_path, monkeypatch):
    sc = _load()
    monkeypatch.setattr(sc, "BASE", str(tmp_path))
    d = tmp_path / "N2"
    alg = 4057726976
    for r, avg_ns in ((0, 1400000.0), (1, 1450000.0)):
        rd = d / f"rank{r}" / "host" / "123"
        rd.mkdir(parents=True)
        with open(rd / "123_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            w.writerow(["void k_cycle<true, 3, true>(SkParams, unsigned char*, int*, int, unsigned long, unsigned long, unsigned int, unsigned int, unsigned int, int)",
                        342, int(342 * avg_ns), avg_ns, 99.7, 1, 2, 3])
            w.writerow(["void k_deal<3>(SkParams, int, int)", 4, 400000, 100000.0, 0.1, 1, 2, 3])
    line = {"value": 9.4e10, "unit": "env-steps/s", "ms_per_step": 1.46, "scaling": "weak",
            "config": {"collective": {"backend": "nccl", "ranks_gathered": 2, "world_size": 2}},
            "roofline": {"algorithmic_bytes_per_launch": alg}}
    (d / "rank0.json").write_text("noise\n" + json.dumps(line) + "\n")
    p = sc.point(2)
    assert p["n_gpus"] == 2 and p["value"] == 9.4e10 and p["collective"]["ranks_gathered"] == 2
    assert [r["kernel"] for r in p["ranks"]] == ["k_cycle<true, 3, true>"] * 2 and [r["calls"] for r in p["ranks"]] == [342, 342]
    assert abs(p["ranks"][0]["achieved_GBs"] - alg / 1.4e-3 / 1e9) < 1e-6 and abs(p["ranks"][1]["frac_of_peak"] - alg / 1.45e-3 / 1e9 / 8000.0) < 1e-9
    assert json.load(open(d / "scale_point.json"))["ranks"][1]["avg_us"] == 1450.0
