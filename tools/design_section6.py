"""Regenerate DESIGN.md section 6 (between the `<!-- section6:begin -->` / `<!-- section6:end -->` markers) from the committed
profiles/<round>_* files, so that every figure quoted there is one a reader finds in profiles/ (tests/test_design_quotes.py checks
exactly that).  python tools/design_section6.py [r6]"""
import csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = lambda name: os.path.join(ROOT, "profiles", R + "_" + name)


def sci(x, digits=2):
    e = len(str(int(x))) - 1
    sup = str(e).translate(str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹"))
    return f"{x / 10 ** e:.{digits}f} × 10{sup}"


def th(x, d=1):
    """1234.5 -> '1 234.5' (thin thousands separator as in the rest of DESIGN.md)"""
    return f"{x:,.{d}f}".replace(",", " ")


def figures():
    b = json.load(open(P("bench.json")))
    t = json.load(open(P("hbm_traffic.json")))
    a = json.load(open(P("hbm_traffic_attribution.json")))
    dg = json.load(open(P("kernel_trace_digest.json")))
    stats = None
    for r in csv.DictReader(open(P("kernel_stats.csv"))):
        if "k_cycle" in r["Name"]:
            stats = r
            break
    kc = [d for d in dg if d.get("kernel", "").startswith("k_cycle")][0]
    under = json.load(open(P("bench_under_rocprof.json")))
    return b, t, a, stats, kc, under


def text():
    b, t, a, stats, kc, under = figures()
    rf, cb, oc = b["roofline"], b["cpu_baseline"], b["other_configs"]
    alg = rf["algorithmic_bytes_per_launch"]
    iters = b["config"]["iterations_per_step"]
    avg_us = float(stats["AverageNs"]) / 1e3
    calls = int(stats["Calls"])
    traffic = t["k_step_bytes_per_launch"]
    m = a["mt19937_records"]
    ph = list(a["philox_records"].values())[0]
    nr = list(a["mt19937_no_records"].values())[0]
    rec_w = m["write_bytes_per_launch"] - nr["write_bytes_per_launch"]
    gen_r, gen_w = m["read_bytes_per_launch"] - ph["read_bytes_per_launch"], m["write_bytes_per_launch"] - ph["write_bytes_per_launch"]
    rest_r, rest_w = ph["read_bytes_per_launch"], ph["write_bytes_per_launch"] - rec_w
    deals = b["roofline_path"]["k_deal"]["deals_per_step"]
    outs = deals * b["roofline_path"]["k_deal"]["rng_outputs_per_deal"]
    g = lambda x: f"{x / 1e9:.2f}"
    o = lambda k: oc[k]["value"]
    lines = []
    w = lines.append
    w(f"## 6. Measurement (SURVEY §8 row d) — one MI355X, `profiles/{R}_*`, `python bench.py --steps 20 --warmup 5`")
    w("")
    w(f"(Generated from the committed profile files by `tools/design_section6.py {R}`; `tests/test_design_quotes.py` fails when a figure here")
    w("drifts from them.)  One bench step = ONE launch of `k_cycle`: " + th(iters, 0) + " lockstep iterations of all 65 536 games = sixteen dealing cycles")
    w("of 64 (the ABI's maximum per launch), the dealing runs inside (the first one planned by the launch before), records in the tile-planar")
    w("layout; 5 timed blocks after 100 set-up + W warm-up launches (blocks 2 .. 5 behind min(W, 8) untimed launches: below), `value` = the median block; asserts `episodes > 0`, `resets > 0`,")
    w("`waits == 0`; inputs resident in HBM (the `*_host` conveniences add PCIe and are for single-game views only: §4).")
    w("")
    w("| Quantity | Value |")
    w("|---|---|")
    w(f"| `value` (config 3: 65 536 × 3, MT19937, records) | **{sci(b['value'])} env-steps/s** (`{R}_bench.json`; blocks {sci(b['blocks']['min'])} – {sci(b['blocks']['max'])}; "
      f"{b['ms_per_iteration'] * 1e3:.3f} µs per lockstep iteration; round 5: 4.77 × 10¹⁰ here, 4.59 on the driver's box; from box to box ± 3 %: 4.68 – 4.96 in the runs of this round - `r6_bench_second_box.json` 4.68, `r6_scale_n1.json` 4.92, the A/B of EXPERIMENTS round 6 #11 4.955; the same-box gains of the round are + 1 - 2 % from the scheduler strategy and the ~ 2 % a 20-launch block lost behind a pause); target was 10⁷ |")
    w(f"| `roofline` (`k_cycle<indirect, 3, planar>`, the only kernel of the path) | {th(alg / 1e6)} MB algorithmic (65 536 × (2·238 + {iters}·60)) / "
      f"{th(rf['avg_launch_ms'] * 1e3)} µs (HIP events, last 32 launches) = {rf['achieved'] / 1e3:.2f} TB/s = **{rf['frac']:.3f} of 8 TB/s** — the dealing is INSIDE this time; "
      f"`rocprofv3 --stats`: **{th(avg_us)} µs** average over {calls} launches of a separate, profiled run, {th(kc['last32_avg_us'])} µs"
      f" over its last 32 (`{R}_kernel_stats.csv`, `{R}_kernel_trace_digest.json`; that run's own line: `{R}_bench_under_rocprof.json`, {th(under['roofline']['avg_launch_ms'] * 1e3)} µs) |")
    w(f"| `roofline.traffic` | **{g(traffic)} GB** per launch from the PMC passes (FETCH_SIZE doubled per the guide + WRITE_SIZE; at the fabric: "
      f"{m['fabric']['read_requests'] / 1e6:.1f} M 128-byte line reads, {m['fabric']['write_requests'] / 1e6:.1f} M write requests, {m['fabric']['write_requests_64B'] / 1e6:.1f} M of them full 64-byte ones) "
      f"= **{traffic / alg:.2f} ×** the algorithmic bytes (round 4: 2.48 ×, round 5: 2.14 ×).  `{R}_hbm_traffic.json` carries the launch shape and the sha256 of the kernel sources it was measured on; "
      "bench.py reports the figure only when both match what runs, `null` with the reason otherwise |")
    w(f"| where the bytes go (`{R}_hbm_traffic_attribution.json`: the same launch with counter-based deals, and without records) | records **{g(rec_w)} GB** written "
      f"(= 65 536 × {th(iters, 0)} × 64 B) · generator state **{g(gen_r)} GB** read + **{g(gen_w)} GB** written = {gen_r / outs / (b['steps'] and 1):.1f} + {gen_w / outs:.1f} B per MT19937 output "
      f"({outs / 1e6:.0f} M outputs per launch: {deals / 1e3:.0f} k deals × 415; the algorithm needs 8 + 4; round 4: 15.3 read) · bank + tiles {g(rest_r)} GB read + {g(rest_w)} GB written "
      f"(round 5's game-major bank: a reset reads 3 lines instead of 18 — this row was 1.0 + 0.43 GB) |")
    w(f"| `roofline_path` | = the kernel (one launch per step): `frac_wall` {b['roofline_path']['frac_wall']:.3f} ({b['ms_per_step'] * 1e3:.0f} µs wall per step); in bytes moved "
      f"{g(traffic)} GB / {rf['avg_launch_ms'] * 1e3:.0f} µs = {traffic / (rf['avg_launch_ms'] * 1e-3) / 1e12:.1f} TB/s at the fabric (the generator state lives in the 256 MB memory-side cache) |")
    w(f"| `cpu_baseline` (`kind: port`, oracle with OpenMP) | {sci(cb['value'], 1)} steps/s on {cb['cores']} host threads, {sci(cb['value_1_thread'], 1)} on one ⇒ GPU ÷ port ≈ "
      f"{th(b['value'] / cb['value'], 0)} / {th(b['value'] / cb['value_1_thread'], 0)}; GPU ÷ reference Python (8.3 k / 52.8 k steps/s, BASELINE.md §2) = {sci(b['value'] / 8.3e3, 1)} / {sci(b['value'] / 52.8e3, 1)} |")
    w(f"| `other_configs` (same run; k_cycle at sixteen cycles per launch, planar records unless said; in brackets: round 5) | cfg2 4 096 × 2: **{sci(o('cfg2_4096x2'))}** (3.37; `roofline_frac` {oc['cfg2_4096x2']['roofline_frac']:.3f}: 64 wavefronts "
      f"on 1 024 SIMDs, each bound by its own ~ 1.2 µs per lockstep iteration - the batch cannot fill the chip, the kernel does not waste bytes: §8.3) · cfg4 shard 32 768 × 3 "
      f"(`game_id0 = 3·32 768`): **{sci(o('cfg4_shard_32768x3'))}** (2.66; × 8 GPUs = {sci(8 * o('cfg4_shard_32768x3'), 1)} is a PROJECTION from one shard, not a measurement: §7) · Philox: "
      f"**{sci(o('philox_65536x3'))}** (5.23; {sci(o('philox_131072x3'))} at 131 072 games, two rounds of workgroups) · row-major records (the ABI's default layout; the tile-planar one is an opt-in whose "
      f"consumers read it in place since round 6, a caller who wants the reference's dense arrays pays `skyjo_vec_unpack_tiles` on top): **{sci(o('row_major_records_65536x3'))}** (4.53) · "
      f"direct observation: **{sci(o('direct_obs_65536x3'))}** (4.19) · cfg5 65 536 × 4 with policy + value net: **{sci(o('cfg5_65536x4_model_bf16'))}** (bf16; 1.45), **{sci(o('cfg5_65536x4_model_fp32'))}** "
      "(float32-grade; 0.85), two launches per lockstep iteration: next row |")
    c5 = json.load(open(P("cfg5_bench.json")))
    pm = json.load(open(P("cfg5_pmc.json")))
    def pipes(kname):
        m = {}
        for t in pm:
            m.update(pm[t].get(kname, {}))
        busy = m["SQ_BUSY_CYCLES"] / 32
        return (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / busy, m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / busy, m["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024 / busy, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"])
    pb, ps = pipes("k_net_bf16"), pipes("k_net_split")
    w(f"| config 5 (`tools/bench_cfg5.py`: `{R}_cfg5_bench.json`, `{R}_cfg5_kernel_stats.csv`; counters: `{R}_cfg5_pmc.json`, the round-2 kernels': `{R}_cfg5_pmc_round2_kernels.json`) | bf16: "
      f"{c5['bf16']['ms_per_iteration'] * 1e3:.1f} µs per lockstep iteration = net `k_net_bf16` {c5['bf16']['dominant_kernel_ms'] * 1e3:.1f} + step {c5['bf16']['step_kernel_ms'] * 1e3:.1f} + launch gaps "
      f"(round 5: 46.6 = 31.5 + 11.9 + gaps); float32-grade: {c5['fp32']['ms_per_iteration'] * 1e3:.1f} = `k_net_split` {c5['fp32']['dominant_kernel_ms'] * 1e3:.1f} + {c5['fp32']['step_kernel_ms'] * 1e3:.1f} + gaps (77.4 = 60.1 + 11.9 + gaps).  "
      f"Which pipe bounds which mode (per SIMD, share of the kernel's cycles): bf16 - matrix pipe busy {pb[0]:.2f}, vector ALU {pb[1]:.2f}, both at once {pb[2]:.2f}, wavefronts parked {pb[3]:.2f} of their life "
      f"(round-2 kernel: 0.26 / 0.50 / 0.15 / 0.43): the **vector ALU** - 1 043 transcendentals of 3 466 vector instructions per wavefront; float32-grade - {ps[0]:.2f} / {ps[1]:.2f} / {ps[2]:.2f} / {ps[3]:.2f} "
      f"(0.43 / 0.38 / 0.07 / 0.39): the **matrix pipe**, with layer 1's activations (vector work only, a fifth of a wavefront's life) in front of it.  In model flops: bf16 {oc['cfg5_65536x4_model_bf16']['roofline_frac']:.2f} of the "
      f"2.5 PFLOP/s bf16 peak ({oc['cfg5_65536x4_model_bf16']['mfma_issue_frac']:.2f} in MFMAs issued), float32-grade {oc['cfg5_65536x4_model_fp32']['roofline_frac']:.2f} ({oc['cfg5_65536x4_model_fp32']['mfma_issue_frac']:.2f}) - at the "
      "2.0 / 1.8 GHz the chip holds under these kernels (stamped), not the 2.4 the peak assumes |")
    dg = json.load(open(P("kernel_trace_digest.json")))
    tail = [d for d in dg if "launch_time_tail" in d][0]["launch_time_tail"]
    w(f"| `k_cycle`'s long launches (`{R}_kernel_trace_digest.json`: `launch_time_tail`) | {tail['launches_over_105pct_of_median']} of {tail['of']} launches of the profiled run are more than 5 % over the median "
      f"({th(tail['median_us'])} µs): {tail['in_the_first_16_after_seeding']} among the first 16 after seeding (up to {th(tail['max_us_in_the_first_16'])} µs: every game ends its first episodes within a few iterations of "
      f"the others - bursts of resets and deals), {tail['within_12_launches_behind_a_pause_over_40us']} within twelve launches behind a pause of the launch stream (the host reading counters between two timed blocks; up to "
      f"{th(tail['max_us_behind_a_pause'] or 0)} µs), {tail['elsewhere']} elsewhere.  Round 5's 1.92 ms maximum was the first; bench.py's timed blocks now start behind untimed launches because of the second |")
    sc = json.load(open(P("scale_n1.json")))
    w(f"| scaling point N = 1 on real profiler output (`{R}_scale_n1.json`, `tools/scale_run.sh 1`, `tests/test_gpu_scale_point.py`) | {sci(sc['value'])} steps/s; rocprofv3, last {sc['launches_timed']} dispatches: "
      f"{th(sc['ranks'][0]['achieved_GBs'], 0)} GB/s = {sc['ranks'][0]['frac_of_peak']:.3f} of peak against the same run's HIP-event figure {th(sc['bench_achieved_GBs'], 0)} GB/s; statistics record through RCCL "
      f"(`backend: {sc['collective']['backend']}`, `ranks_gathered: {sc['collective']['ranks_gathered']}`) |")
    c1 = json.load(open(P("cfg1.json")))
    w(f"| config 1 (`tools/bench_cfg1.py`, `profiles/{R}_cfg1.json`) | ONE game from Python through the reference's own loops: `env(**DEFAULT_CONFIG)` "
      f"{c1['env_steps_per_s'] / 1e3:.1f} k steps/s ({c1['env_us_per_step']:.1f} µs per step), `SkyjoGame` core loop {c1['core_steps_per_s'] / 1e3:.1f} k ({c1['core_us_per_step']:.1f} µs); "
      f"the reference's Python: 6.9 k / 8.3 k.  Floor: one native host-style call is {c1['native_call_floor_us']:.1f} µs, the rest is Python |")
    w("")
    return "\n".join(lines)


if __name__ == "__main__":
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    body = "<!-- section6:begin -->\n" + text() + "<!-- section6:end -->\n"
    if "<!-- section6:begin -->" in s:
        s = re.sub(r"<!-- section6:begin -->.*?<!-- section6:end -->\n", lambda _: body, s, flags=re.S)
    else:  # first time: replace the old section 6 up to section 7's heading
        i, j = s.index("## 6. Measurement"), s.index("## 7. Multi-GPU")
        s = s[:i] + body + "\n" + s[j:]
    open(path, "w").write(s)
    print(text())
