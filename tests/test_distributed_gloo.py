"""N > 1 path on CPU: two gloo ranks shard a batch by game id (oracle engine), all-gather their
statistics, and must reproduce the single-process result exactly."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, ITERS, N = 96, 260, 3
CFG = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from skyjo_rl_amd.distributed import gather_stats, make_sharded_env, shard_range
    from tests.oracle_engine import OracleEngine

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = make_sharded_env(TOTAL, rank, world, engine_factory=OracleEngine, **CFG)
    first, count = shard_range(TOTAL, world, rank)
    assert eng.num_envs == count and eng.game_id0 == first
    eng.seed(None, 7)
    eng.rollout_host(ITERS, policy_seed=5)
    obs = eng.observe_host().observations
    allr, totals = gather_stats(eng.counters(), N)
    q.put((rank, first, obs, allr, {k: np.asarray(v) for k, v in totals.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_process():
    sys.path.insert(0, ROOT)
    from skyjo_rl_amd.distributed import shard_range
    from tests.oracle_engine import OracleEngine

    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = OracleEngine(TOTAL, **CFG)
    ref.seed(None, 7)
    ref.rollout_host(ITERS, policy_seed=5)
    full = ref.observe_host().observations
    c = ref.counters()
    for rank, first, obs, allr, totals in res:
        np.testing.assert_array_equal(obs, full[first:first + len(obs)])  # shard == slice of the full batch
        assert allr.shape[0] == 2
        for k in ("steps", "episodes", "resets", "sum_len"):
            assert totals[k] == c[k], (k, totals[k], c[k])
    assert c["episodes"] > 0


def test_the_gather_path_follows_the_groups_backend_not_its_name():
    """which device the statistics record must live on is decided from the process group's backend string per device type
    ("nccl" = RCCL -> device tensors, "gloo" -> host tensors, a mixed "cpu:gloo,cuda:nccl" group -> either, mpi / ucc -> either),
    not from `backend == "gloo"`."""
    from skyjo_rl_amd.distributed import _backend_devices

    assert _backend_devices("nccl") == {"cuda"} and _backend_devices("gloo") == {"cpu"}
    assert _backend_devices("cpu:gloo,cuda:nccl") == {"cpu", "cuda"}
    assert _backend_devices("mpi") == {"cpu", "cuda"} and _backend_devices("ucc") == {"cpu", "cuda"}
