"""the scaling tooling on REAL profiler output, N = 1 (what a one-GPU lease can run): `bash tools/scale_run.sh 1` starts
the rank of `bench.py --gpus 1` under its own `rocprofv3 --kernel-trace --stats` - the program itself behind `--`, a process group
of ONE rank over RCCL ("nccl") because the script sets the rendezvous up -, tools/scale_collect.py digests the trace.  The point's
per-rank GB/s (rocprofv3, last `launches_timed` dispatches) must agree with the same run's own roofline figure (HIP events) within 5 %,
and the statistics record must have gone through the RCCL all-gather.  On the day an 8-GPU node is at hand the only new thing is N."""
import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("rocprofv3") is None, reason="rocprofv3 not installed")
def test_scale_run_one_rank_under_rocprofv3():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(["bash", "tools/scale_run.sh", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    p = json.load(open(os.path.join(ROOT, "gpurun_out", "scale", "N1", "scale_point.json")))
    assert p["n_gpus"] == 1 and len(p["ranks"]) == 1
    c = p["collective"]
    assert c["backend"] == "nccl" and c["ranks_gathered"] == 1 and c["world_size"] == 1
    rk = p["ranks"][0]
    assert rk["kernel"].startswith("k_cycle") and rk["calls"] >= p["launches_timed"]
    assert abs(rk["achieved_GBs"] / p["bench_achieved_GBs"] - 1.0) < 0.05, (rk["achieved_GBs"], p["bench_achieved_GBs"])
    assert p["value"] > 1e10 and 0.2 < rk["frac_of_peak"] < 1.0
