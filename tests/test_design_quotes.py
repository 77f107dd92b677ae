"""DESIGN.md section 6 quotes its figures from the committed profiles/ files (it once quoted a best box instead).
This test re-reads both: the kernel's average launch time, `value`, the roofline fraction and the traffic figure in DESIGN.md must be
within 2 % of profiles/r6_kernel_stats.csv / r6_bench.json / r6_hbm_traffic.json, and the section must be what
tools/design_section6.py generates from them."""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = "r6"


def _num(text):
    return float(text.replace(" ", "").replace(" ", ""))


def _section6():
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    return s[s.index("<!-- section6:begin -->"):s.index("<!-- section6:end -->")]


def test_quoted_kernel_time_value_fraction_and_traffic_match_the_committed_profiles():
    sec = _section6()
    bench = json.load(open(os.path.join(ROOT, "profiles", R + "_bench.json")))
    traffic = json.load(open(os.path.join(ROOT, "profiles", R + "_hbm_traffic.json")))
    avg_us = None
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", R + "_kernel_stats.csv"))):
        if "k_cycle" in r["Name"]:
            avg_us, calls = float(r["AverageNs"]) / 1e3, int(r["Calls"])
            break
    assert avg_us is not None
    m = re.search(r"`rocprofv3 --stats`: \*\*([\d ]+\.\d) µs\*\* average over (\d+) launches", sec)
    assert m, "DESIGN.md section 6 no longer quotes the rocprofv3 --stats average"
    assert abs(_num(m.group(1)) - avg_us) <= 0.02 * avg_us and int(m.group(2)) == calls, (m.groups(), avg_us, calls)
    m = re.search(r"/ ([\d ]+\.\d) µs \(HIP events", sec)
    assert m and abs(_num(m.group(1)) - bench["roofline"]["avg_launch_ms"] * 1e3) <= 0.02 * bench["roofline"]["avg_launch_ms"] * 1e3
    m = re.search(r"\*\*(\d\.\d\d) × 10¹⁰ env-steps/s\*\*", sec)
    assert m and abs(float(m.group(1)) * 1e10 - bench["value"]) <= 0.02 * bench["value"], (m and m.group(1), bench["value"])
    m = re.search(r"\*\*(0\.\d+) of 8 TB/s\*\*", sec)
    assert m and abs(float(m.group(1)) - bench["roofline"]["frac"]) <= 0.02 * bench["roofline"]["frac"]
    # the HIP-event figure of the un-profiled run and the profiler's average of the profiled one agree (the same kernel, two runs)
    assert abs(bench["roofline"]["avg_launch_ms"] * 1e3 - avg_us) <= 0.03 * avg_us
    m = re.search(r"`roofline.traffic` \| \*\*(\d+\.\d\d) GB\*\*", sec)
    assert m and abs(float(m.group(1)) * 1e9 - traffic["k_step_bytes_per_launch"]) <= 0.02 * traffic["k_step_bytes_per_launch"]
    # the line's own traffic field is the digest's (bench.py only reports it for the kernel sources it was measured on)
    # (or the digest of the refresh before: the line of a refresh run carries the digest committed when it ran, and two refreshes of the
    # same kernel differ in the fifth digit)
    tr = bench["roofline"]["traffic"]
    assert tr is None or abs(tr - traffic["k_step_bytes_per_launch"]) <= 1e-3 * traffic["k_step_bytes_per_launch"]


def test_section_is_what_the_generator_writes():
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); sys.argv = ['x', %r]; import design_section6 as d; print(d.text())" % (os.path.join(ROOT, "tools"), R)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() in _section6(), "DESIGN.md section 6 is stale: run `python tools/design_section6.py r6`"
