"""GPU: bench.py end to end, as the driver starts it - the one-rank line with every BASELINE configuration in
`other_configs`, and the multi-rank launch path (`--gpus 2`: bench.py spawns its own torchrun child; on this one-GPU box
the two ranks share the card, which bench.py only does when told so - otherwise it refuses loudly)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT)


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_line_carries_every_baseline_config():
    d = _line(_bench("--steps", "4", "--warmup", "2", "--blocks", "3", "--no-cpu-baseline"))
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["unit"] == "env-steps/s" and d["value"] > 1e9
    assert d["blocks"]["n"] == 3 and d["blocks"]["min"] <= d["value"] <= d["blocks"]["max"]
    assert abs(d["value"] - d["blocks"]["median"]) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["waits"] == 0
    # the PMC traffic is reported only for the kernel sources it was measured on (sha256 stored with the digest): a figure, or
    # null with the reason
    assert d["roofline"]["traffic_source"]
    assert (d["roofline"]["traffic"] is None) == (d["roofline_path"]["traffic"] is None)
    assert d["scaling"] == "weak" and d["config"]["collective"]["ranks_gathered"] == 1 and d["config"]["collective"]["world_size"] == 1
    oc = d["other_configs"]
    for k in ("cfg2_4096x2", "cfg4_shard_32768x3", "cfg5_65536x4_model_fp32", "cfg5_65536x4_model_bf16", "philox_65536x3",
              "direct_obs_65536x3"):
        assert "error" not in oc[k], (k, oc[k])
        assert oc[k]["value"] > 1e8 and oc[k]["dominant_kernel_ms"] > 0 and 0 < oc[k]["roofline_frac"] < 1, (k, oc[k])
    assert oc["cfg5_65536x4_model_fp32"]["illegal"] == 0
    assert d["speedup_vs_reference_constant"]["vs_1_core"] > 1e5


def test_bench_two_ranks_on_one_card_rehearsal_and_refusal():
    """`bench.py --gpus 2` starts its two ranks itself.  With one GPU visible it must refuse (exit code 3, a message) unless
    SKYJO_BENCH_SHARED_GPU=1 asks for the rehearsal, in which the ranks share the card and gloo carries the record."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible: the refusal / rehearsal paths are for one-GPU boxes")
    out = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "1", "--no-cpu-baseline")
    assert out.returncode != 0 and "refusing to let ranks share a card" in out.stderr
    d = _line(_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline", env={"SKYJO_BENCH_SHARED_GPU": "1"}))
    assert d["n_gpus"] == 2 and d["config"]["shared_gpu_rehearsal"] is True and d["episode_stats"]["ranks"] == 2
    assert d["config"]["collective"] == dict(d["config"]["collective"], backend="gloo", ranks_gathered=2, world_size=2) and d["scaling"] == "weak"
    assert d["config"]["games_per_gpu"] == 65536 and d["value"] > 1e9 and "other_configs" not in d
    # the two shards together applied steps for 2 x 65 536 games: more than one shard could in the same iterations
    assert d["episodes"] > 2 * 65536
    # --config 4: 32 768 games per rank (BASELINE configs[3])
    d4 = _line(_bench("--gpus", "2", "--config", "4", "--steps", "3", "--warmup", "1", "--blocks", "1", "--no-cpu-baseline",
                      env={"SKYJO_BENCH_SHARED_GPU": "1"}))
    assert d4["config"]["games_per_gpu"] == 32768 and d4["config"]["baseline_config"] == 4 and "262 144" in d4["config"]["parallelism"]
    # strong scaling, stated as such: 65 536 games in total, half of them on each rank
    ds = _line(_bench("--gpus", "2", "--total-games", "65536", "--steps", "3", "--warmup", "1", "--blocks", "1", "--no-cpu-baseline",
                      env={"SKYJO_BENCH_SHARED_GPU": "1"}))
    assert ds["scaling"] == "strong" and ds["config"]["games_per_gpu"] == 32768 and ds["config"]["games_total"] == 65536
    assert ds["config"]["scaling_mode"].startswith("strong")


def test_bench_flags_for_the_other_layout_and_observation():
    """`--record-layout row-major` (the ABI's default layout) and `--direct-obs` (wider records, tile-planar by default) run the same
    timed loop; the line says which layout it measured, and the PMC traffic figure is only reported for the shape it was taken on."""
    common = ("--steps", "2", "--warmup", "1", "--blocks", "1", "--no-cpu-baseline", "--no-other-configs")
    d = _line(_bench("--record-layout", "row-major", *common))
    assert d["config"]["record_layout"] == "row-major" and d["roofline"]["launch_shape"]["record_layout"] == "row-major"
    assert d["roofline"]["traffic"] is None and "launch shape" in d["roofline"]["traffic_source"] and d["value"] > 1e9
    d = _line(_bench("--direct-obs", *common))
    assert d["config"]["record_layout"] == "tile-planar" and d["roofline"]["launch_shape"]["direct_obs"] is True and d["value"] > 1e9
    assert "direct" in d["roofline"]["kernel"]
