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


def test_scale_point_uses_the_timed_dispatches_of_the_kernel_trace(tmp_path, monkeypatch):
    """With a kernel trace at hand the per-rank figure is the average over the LAST `launches_timed` dispatches of the dominant kernel
    (the settle / warm-up launches - other dealing intervals - stay out, like in bench.py's own roofline leg: ADVICE r5)."""
    sc = _load()
    monkeypatch.setattr(sc, "BASE", str(tmp_path))
    d = tmp_path / "N1"
    rd = d / "rank0" / "host" / "7"
    rd.mkdir(parents=True)
    name = "void k_cycle<true, 3, true>(SkParams, unsigned char*, int*, int, unsigned long, unsigned long, unsigned int, unsigned int, unsigned int, int)"
    with open(rd / "7_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"])
        t = 1000
        for i in range(100):  # 60 settle launches of 90 us, then 40 of 1 400 us
            dur = 90_000 if i < 60 else 1_400_000
            w.writerow(["KERNEL_DISPATCH", 1, 1, 5, name, i, t, t + dur])
            t += dur + 5_000
        w.writerow(["KERNEL_DISPATCH", 1, 1, 6, "void k_seed(SkParams, unsigned long const*, unsigned long, int, int)", 101, t, t + 50_000])
    line = {"value": 4.7e10, "unit": "env-steps/s", "ms_per_step": 1.4, "scaling": "weak",
            "config": {"collective": {"backend": "nccl", "ranks_gathered": 1, "world_size": 1}},
            "roofline": {"algorithmic_bytes_per_launch": 4057726976, "launches_timed": 32, "achieved": 2898.4}}
    (d / "rank0.json").write_text(json.dumps(line) + "\n")
    p = sc.point(1)
    r = p["ranks"][0]
    assert r["calls"] == 100 and r["avg_us"] == 1400.0 and abs(r["achieved_GBs"] - 4057726976 / 1.4e-3 / 1e9) < 1e-6
    assert p["bench_achieved_GBs"] == 2898.4 and p["launches_timed"] == 32
