"""GPU: multi-GPU shard semantics on one card (BASELINE config 4: 8 x 32 768 games, shard r owns global game ids
r*32768 ...), the product's statistics gather in two processes, whole-engine snapshot / restore, and the per-handle
device guard of the C ABI."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(num_players=3, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001,
           rng_mode=0, auto_reset=True)


def _engine(*a, **k):
    from skyjo_rl_amd import SkyjoVecEnv
    return SkyjoVecEnv(*a, **k)


def test_config4_shard_32768_with_game_id0():
    """Rank 3 of BASELINE config 4: 32 768 games with game_id0 = 3 * 32 768 (the dealing kernel runs beside the step
    kernel at this size).  A 256-game window of the shard is re-simulated by the oracle from global ids alone."""
    import torch
    from oracle import skyjo_oracle as so
    from skyjo_rl_amd.distributed import make_sharded_env, shard_range

    first, count = shard_range(262144, 8, 3)
    assert (first, count) == (3 * 32768, 32768)
    eng = make_sharded_env(262144, 3, 8, **CFG)
    assert eng.num_envs == count and eng.game_id0 == first
    eng.seed(None, 0)
    lo, n = 20000, 256
    ora = so.OracleVec(num_envs=n, game_id0=first + lo, **CFG)
    ora.seed(None, 0)
    K = 80
    for r in range(5):
        rec = eng.new_records(K)
        eng.rollout(K, policy_seed=1, records=rec)
        oact = ora.rollout(K, 1, record_actions=True)
        v = eng.split(rec)
        np.testing.assert_array_equal(v.action[:, lo:lo + n].cpu().numpy(), oact.astype(np.int8), err_msg=f"round {r}")
        obs, mask, agent, phase = ora.observe()
        np.testing.assert_array_equal(v.observations[K - 1, lo:lo + n].cpu().numpy(), obs)
        np.testing.assert_array_equal(v.action_mask[K - 1, lo:lo + n].cpu().numpy(), mask)
    c = eng.counters()
    assert c["steps"] + c["resets"] == 5 * K * count and c["waits"] == 0 and c["episodes"] > 2 * count
    # per-seat statistics of the RCCL record: mean reward over seats == mean_reward + refunded bonus
    ep = c["episodes"]
    assert abs(c["sum_reward"].sum() / ep - (3 * 1.0 + 0.001 * c["sum_refunded"].sum() / ep)) < 1e-6
    assert np.all(c["sum_reward_sq"] > 0) and np.all(c["sum_score"] / ep > 10)
    eng.close()


_WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from skyjo_rl_amd.distributed import make_sharded_env, gather_stats
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")  # two ranks share the one GPU of the test box: RCCL refuses that, gloo carries the record
cfg = dict(num_players=3, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001, rng_mode=0, auto_reset=True)
eng = make_sharded_env(4096, rank, world, device=0, **cfg)
eng.seed(None, 5)
eng.rollout(400, policy_seed=2)
c = eng.counters()
per_rank, tot = gather_stats(c, 3, device=torch.device("cuda", 0))
if rank == 0:
    print("RESULT " + json.dumps(dict(steps=float(tot["steps"]), episodes=float(tot["episodes"]), sum_len=float(tot["sum_len"]),
          sum_reward=[float(x) for x in tot["sum_reward"]], sum_score=[float(x) for x in tot["sum_score"]], ranks=per_rank.shape[0])))
eng.close()
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gather_of_product_engines_equals_single_process():
    """Two processes, each with the PRODUCT engine for its shard (same GPU), all-gather their statistics record
    (skyjo_rl_amd.distributed.gather_stats); the totals equal those of one process running all 4096 games."""
    import json

    script = _WORKER.format(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sock:  # a free port: suites may run side by side
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), _write_tmp(script)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    got = json.loads(line[7:])
    eng = _engine(4096, **CFG)
    eng.seed(None, 5)
    eng.rollout(400, policy_seed=2)
    c = eng.counters()
    assert got["ranks"] == 2
    assert got["steps"] == c["steps"] and got["episodes"] == c["episodes"] and got["sum_len"] == c["sum_len"]
    np.testing.assert_allclose(got["sum_score"], c["sum_score"], rtol=0, atol=1e-6)  # (integer-valued: exact up to the order of float adds)
    np.testing.assert_allclose(got["sum_reward"], c["sum_reward"], rtol=1e-12)
    eng.close()


def _write_tmp(text):
    import tempfile

    f = tempfile.NamedTemporaryFile("w", suffix="_skyjo_worker.py", delete=False)
    f.write(text)
    f.close()
    return f.name


def test_snapshot_restore_round_trip():
    """Snapshot at t, run 300 iterations, restore, run them again: identical records, actions and counters - the
    snapshot holds the RNG streams with their positions, the banks of pre-dealt episodes and the policy counter."""
    import torch

    B, K = 4096, 100
    eng = _engine(B, **CFG)
    eng.seed(None, 11)
    eng.rollout(137, policy_seed=4)  # somewhere in the middle of things (banks partly used, dealing cycle under way)
    snap = eng.snapshot()
    assert snap.nbytes > B * (288 * 4 + 2496)
    c0 = eng.counters()

    def run():
        recs = []
        for r in range(3):
            rec = eng.new_records(K)
            eng.rollout(K, policy_seed=4, records=rec)
            recs.append(rec)
        return torch.stack(recs), eng.counters()

    a, ca = run()
    eng.restore(snap)
    cr = eng.counters()
    for k in ("steps", "episodes", "resets", "sum_len", "iters"):
        assert cr[k] == c0[k], k
    b, cb = run()
    assert torch.equal(a, b)
    for k in ("steps", "episodes", "resets", "sum_len", "waits"):
        assert ca[k] == cb[k], k
    np.testing.assert_array_equal(ca["sum_reward"], cb["sum_reward"])
    assert ca["episodes"] > c0["episodes"] + B  # the window held resets, i.e. bank hand-overs and deals
    # a single-game view after a restore sees the restored game
    s1 = eng.get_state(5)
    eng.rollout(10, policy_seed=4)
    eng.restore(snap)
    eng.rollout(3 * K, policy_seed=4)
    s2 = eng.get_state(5)
    for k in ("cards", "masked", "draw", "disc"):
        np.testing.assert_array_equal(s1[k], s2[k])
    snap.close()
    eng.close()


def test_handle_runs_on_its_device_from_any_thread():
    """every entry point switches to the handle's device and restores the caller's.  On this one-GPU box the
    observable part is: calls from a fresh thread (whose current device was never set) work and leave torch's current
    device alone; a handle for a device that does not exist is refused."""
    import torch
    from skyjo_rl_amd import SkyjoNativeError

    eng = _engine(64, **CFG)
    eng.seed(None, 1)
    out = {}

    def worker():
        try:
            o = eng.observe_host()
            o2 = eng.step_host(np.full(64, 24, dtype=np.int32))
            out["ok"] = bool((o.phase == 0).all() and (o2.phase == 1).all())
            out["state"] = eng.get_state(3)["phase"]
            out["counters"] = eng.counters()["steps"]
        except Exception as e:  # pragma: no cover
            out["err"] = repr(e)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert out.get("ok") and out["state"] == 1 and out["counters"] == 64, out
    assert torch.cuda.current_device() == 0
    with pytest.raises(SkyjoNativeError):
        _engine(64, device=torch.cuda.device_count(), **CFG)
    eng.close()


@pytest.mark.parametrize("players,form", [(1, 2), (2, 3)], ids=["two_streams", "one_kernel"])
def test_step_kernel_gives_up_loudly_when_the_dealing_kernel_never_arrives(players, form):
    """with the dealing kernel on its own stream a step kernel may have to wait for the one deal in flight for a
    game (bank empty, slot busy).  If that dealing launch does not make progress the wait is bounded and must NOT fall through
    silently: the kernel raises a sticky device error and skyjo_vec_get_counters fails until the engine is re-seeded.  Forced
    here by fault injection: every dealing wavefront sleeps ~0.2 s, the spin limit is 2^6 polls, one-player games (25 steps
    per episode) empty their banks of three inside one long dealing cycle."""
    import torch
    from skyjo_rl_amd import SkyjoNativeError

    cfg = dict(CFG, num_players=players)
    eng = _engine(128, **cfg)
    eng.set_overlap(form)   # 2: the two-stream form (one-player games use the generic kernels); 3: k_cycle - there the step wavefronts
    #                         wait for the dealing wavefronts of their own workgroup on their way out of every launch
    assert eng.dealing_form() == {2: "two streams", 3: "one kernel"}[form]
    eng.set_deal_interval(120 if players == 1 else 30)
    eng.seed(None, 2)
    assert eng.counters()["steps"] == 0          # healthy so far
    eng.set_debug_option(3, 6)                   # give up after 64 polls
    eng.set_debug_option(4, 60000)               # each dealing wavefront sleeps 60000 x 8128 cycles ~ 0.2 s
    eng.rollout(360, policy_seed=1)              # dealing cycles whose wavefronts sleep: banks run dry / the step wavefronts' wait runs out
    with pytest.raises(SkyjoNativeError, match="gave up waiting for the dealing kernel"):
        eng.counters()
    # not only the counters - every synchronising call reports the voided run, and a snapshot of it is refused
    for call in (eng.check_error, eng.observe_host, lambda: eng.step_host(np.full(128, 24, dtype=np.int32)), eng.rewards_host,
                 lambda: eng.get_state(0), eng.snapshot):
        with pytest.raises(SkyjoNativeError, match="gave up waiting for the dealing kernel"):
            call()
    eng.set_debug_option(4, 0)
    eng.set_debug_option(3, 22)
    eng.seed(None, 2)                            # a new seeding clears the error
    eng.rollout(100, policy_seed=1)
    c = eng.counters()
    assert c["steps"] > 0 and c["episodes"] > 0
    eng.check_error()
    eng.close()


def test_the_very_host_call_whose_kernel_timed_out_fails():
    """the sticky device error reaches the host from EVERY wavefront of the host-style kernels, not only from tile 0
    (which ends long before a tile that spins through its timeout): a `step_host` that returns normally must leave nothing for
    `check_error` to find.  Two tiles of one-player games, the two-stream form, dealing wavefronts that sleep, a short spin
    limit: some `step_host` call has to fail - and none may succeed while the device word is already raised."""
    from skyjo_rl_amd import SkyjoNativeError

    cfg = dict(CFG, num_players=1)
    eng = _engine(128, **cfg)
    eng.set_overlap(2)
    eng.set_deal_interval(40)
    eng.seed(None, 4)
    eng.set_debug_option(3, 6)
    eng.set_debug_option(4, 60000)
    failed_at = None
    for t in range(400):
        try:
            o = eng.observe_host()
            acts = np.argmax(o.action_mask, axis=1).astype(np.int32)
            eng.step_host(acts)
        except SkyjoNativeError as e:
            assert "gave up waiting for the dealing kernel" in str(e)
            failed_at = t
            break
        eng.check_error()  # a host call that came back clean leaves no raised error behind
    assert failed_at is not None and failed_at > 10, failed_at
    eng.set_debug_option(4, 0)
    eng.set_debug_option(3, 22)
    eng.seed(None, 4)
    eng.step_host(np.full(128, 24, dtype=np.int32))
    eng.check_error()
    eng.close()


def test_snapshot_of_a_destroyed_handle_is_refused_by_its_successor():
    """a snapshot remembers the handle's generation and its array table, not just its address."""
    from skyjo_rl_amd import SkyjoNativeError

    a = _engine(256, **CFG)
    a.seed(None, 1)
    snap = a.snapshot()
    a.restore(snap)  # its own handle: fine
    a.close()
    for n in (256, 320):  # (the allocator likes to hand the same address to the next handle)
        b = _engine(n, **dict(CFG, num_players=3 if n == 256 else 4))
        b.seed(None, 1)
        with pytest.raises(SkyjoNativeError, match="another handle|does not fit"):
            b.restore(snap)
        b.close()
    snap.close()


def test_config4_all_eight_shards_equal_one_262144_game_engine():
    """BASELINE config 4 at its stated size on ONE card (VERDICT r3 "missing" #2): the eight shards of 262 144 three-player
    games - what ranks 0..7 of `bench.py --gpus 8 --config 4` run, 32 768 games each with game_id0 = r * 32 768 - one after
    another, against ONE engine holding all 262 144 games.  Every record of every iteration of shard r equals the slice
    [r * 32 768, (r + 1) * 32 768) of the big engine's records; the shards' statistics records, combined the way every rank
    combines the all-gathered matrix (distributed.combine_stats - the all-gather itself is exercised with two ranks above and
    with gloo in tests/test_distributed_gloo.py), equal the big engine's counters and per-seat sums."""
    import torch
    from skyjo_rl_amd.distributed import combine_stats, make_sharded_env, shard_range, stats_record

    TOTAL, W, K, LAUNCHES = 262144, 8, 64, 4
    big = _engine(TOTAL, **CFG)
    big.seed(None, 0)
    recs = []
    for _ in range(LAUNCHES):
        r = big.new_records(K)
        big.rollout(K, policy_seed=1, records=r)
        recs.append(r)
    cb = big.counters()
    assert cb["steps"] + cb["resets"] == LAUNCHES * K * TOTAL and cb["waits"] == 0 and cb["episodes"] > TOTAL
    big.close()
    rows = []
    for rank in range(W):
        first, count = shard_range(TOTAL, W, rank)
        eng = make_sharded_env(TOTAL, rank, W, **CFG)
        assert (eng.game_id0, eng.num_envs) == (first, count) == (rank * 32768, 32768)
        eng.seed(None, 0)
        rec = eng.new_records(K)
        for l in range(LAUNCHES):
            eng.rollout(K, policy_seed=1, records=rec)
            assert torch.equal(rec, recs[l][:, first:first + count]), f"shard {rank}, launch {l}: records differ from the 262 144-game engine's slice"
        c = eng.counters()
        assert c["waits"] == 0
        rows.append(stats_record(c, 3))
        eng.close()
    tot = combine_stats(np.asarray(rows), 3)
    for k in ("steps", "episodes", "illegal", "resets", "sum_len", "reshuffles", "waits"):
        assert tot[k] == cb[k], (k, tot[k], cb[k])
    np.testing.assert_allclose(tot["sum_score"], cb["sum_score"], rtol=0, atol=1e-6)     # (integer-valued sums)
    np.testing.assert_array_equal(tot["sum_refunded"], cb["sum_refunded"])
    np.testing.assert_allclose(tot["sum_reward"], cb["sum_reward"], rtol=1e-12)
    np.testing.assert_allclose(tot["sum_reward_sq"], cb["sum_reward_sq"], rtol=1e-12)
    assert abs(tot["mean_episode_len"] - cb["sum_len"] / cb["episodes"]) < 1e-12


def test_engines_of_different_shapes_side_by_side_in_one_process():
    """The one-kernel form sizes its workgroups by the batch (S = 1 .. 4 step + dealing wavefronts, up to 159 KB of dynamic LDS):
    engines of different sizes alive at the same time, launched alternately, each equal to its oracle."""
    from oracle import skyjo_oracle as so

    shapes = [(65536, 3), (4096, 2), (20000, 3), (40000, 4)]
    engs, oras = [], []
    for B, N in shapes:
        cfg = dict(CFG, num_players=N)
        e = _engine(B, **cfg)
        assert e.dealing_form() == "one kernel"
        e.seed(None, 9)
        o = so.OracleVec(num_envs=B, **cfg)
        o.seed(None, 9)
        engs.append(e), oras.append(o)
    import torch
    for r in range(3):
        for (B, N), e, o in zip(shapes, engs, oras):
            K = 2 * e.deal_interval() if r == 1 else 48
            act = torch.empty((K, B), dtype=torch.int32, device="cuda")
            e.rollout(K, policy_seed=5, actions=act)
            np.testing.assert_array_equal(act.cpu().numpy(), o.rollout(K, 5, threads=16, record_actions=True), err_msg=f"{B} x {N}, round {r}")
    for e in engs:
        e.close()
