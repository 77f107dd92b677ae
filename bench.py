#!/usr/bin/env python3
"""bench.py - env-steps/sec of the vectorised SkyJo hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one lockstep pass of the hot path over the whole batch: every live game applies one
action chosen by the on-device random admissible policy (state transition + observation / action
mask build, record and action written to HBM), finished games take their next deal.  The workload
is BASELINE.json configs[2]: 65 536 parallel 3-player games per GPU, DEFAULT_CONFIG (indirect
observation, D = 31), game g seeded base + g, numpy-legacy MT19937 deals (bit-identical to the
reference).  `value` = env-steps (applied actions, counted on device) of all ranks / max-over-ranks
wall time, inputs resident in HBM.  One JSON line on stdout (rank 0).

Extra objects on the same line:
  roofline      dominant kernel (k_step, fused rollout) timed with HIP events on its launch stream
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's algorithm) timed on host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec
CHUNK = int(os.environ.get("SKYJO_BENCH_CHUNK", "80"))  # lockstep iterations per kernel launch (<= kMaxRolloutChunk in skyjo_capi.hip;
                                                         # 80 = the engine's dealing interval for 3 and more players)


def algorithmic_bytes_per_launch(B, N, D, iters, actions=True):
    """packed state S(N) = 24N + 150 + 16 read once and written once per launch,
    plus per env-step the record the path must emit: D obs + 26 mask + 2 (agent, done) (+4 action)."""
    S = 24 * N + 150 + 16
    per_step = D + 26 + 2 + (4 if actions else 0)
    return B * (2 * S + iters * per_step)


def cpu_baseline(num_players, seconds=12.0):
    """Oracle (kind 'port') on the host cores: same rollout (same policy restatement), bounded sample."""
    from oracle import skyjo_oracle as so

    threads = os.cpu_count() or 1
    B = 2048 * threads
    ora = so.OracleVec(num_envs=B, num_players=num_players, score_penalty=2.0, observe_other_player_indirect=True,
                       mean_reward=1.0, reward_refunded=0.001, rng_mode=so.RNG_MT19937, auto_reset=True)
    ora.seed(None, 0)
    ora.rollout(8, 1, threads=threads)  # warm-up
    s0 = ora.counters()["steps"]
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < seconds:
        ora.rollout(32, 1, threads=threads)
        iters += 32
    dt = time.perf_counter() - t0
    steps = ora.counters()["steps"] - s0
    # the same port on ONE host thread (SURVEY 8d asks for both), a few seconds
    one = so.OracleVec(num_envs=4096, num_players=num_players, score_penalty=2.0, observe_other_player_indirect=True,
                       mean_reward=1.0, reward_refunded=0.001, rng_mode=so.RNG_MT19937, auto_reset=True)
    one.seed(None, 0)
    one.rollout(8, 1, threads=1)
    s1, t1 = one.counters()["steps"], time.perf_counter()
    while time.perf_counter() - t1 < 3.0:
        one.rollout(16, 1, threads=1)
    single = (one.counters()["steps"] - s1) / (time.perf_counter() - t1)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port", "value_1_thread": single,
            "sample": f"{B} games x {iters} lockstep iterations ({steps} env-steps, {dt:.1f} s), "
                      f"oracle/skyjo_oracle.c with OpenMP over games, same on-device-policy restatement"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)   # 57 ms of GPU time at 2.8 us per lockstep iteration: short enough to
    ap.add_argument("--warmup", type=int, default=2000)   # finish at once, long enough that the first launches after the barrier do not show
    ap.add_argument("--num-envs", type=int, default=65536, help="games per GPU")
    ap.add_argument("--num-players", type=int, default=3)
    ap.add_argument("--rng", choices=["mt19937", "philox"], default="mt19937")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-records", action="store_true", help="do not write records/actions (not the headline)")
    ap.add_argument("--direct-obs", action="store_true", help="observe_other_player_indirect=False: D = 19 + 12 N (not the headline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from skyjo_rl_amd import RNG_MT19937, RNG_PHILOX, SkyjoVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B, N = args.num_envs, args.num_players
    global CHUNK
    if N < 3:
        CHUNK = min(CHUNK, 64)  # the engine deals every 64 iterations below three players (shorter episodes)
    eng = SkyjoVecEnv(B, num_players=N, score_penalty=2.0, observe_other_player_indirect=not args.direct_obs, mean_reward=1.0,
                      reward_refunded=0.001, device=local_rank,
                      rng_mode=RNG_MT19937 if args.rng == "mt19937" else RNG_PHILOX, auto_reset=True,
                      game_id0=rank * B)  # shards by global game id: results do not depend on the GPU count
    eng.seed(None, 0)
    D = eng.obs_dim
    record = not args.no_records
    rec = eng.new_records(CHUNK) if record else None           # [CHUNK, B, 64] ring reused by every launch
    act = torch.empty((CHUNK, B), dtype=torch.int32, device=dev) if record else None

    def run(iters):
        done = 0
        while done < iters:
            n = min(CHUNK, iters - done)
            eng.rollout(n, policy_seed=1, records=rec[:n] if record else None, actions=act[:n] if record else None)
            done += n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    barrier()
    c0 = eng.counters()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()

    # the one collective of the path: per-rank episode statistics, all-gathered over RCCL
    stats = torch.tensor([c1["steps"] - c0["steps"], c1["episodes"] - c0["episodes"], c1["sum_len"] - c0["sum_len"],
                          c1["resets"] - c0["resets"], c1["waits"] - c0["waits"], c1["illegal"] - c0["illegal"], dt],
                         dtype=torch.float64, device=dev)
    if world > 1:
        gathered = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(gathered, stats)
        allstats = torch.stack(gathered).cpu()
    else:
        allstats = stats.cpu().unsqueeze(0)
    steps_total = float(allstats[:, 0].sum())
    t_max = float(allstats[:, 6].max())

    # roofline leg: the same launches again, each k_step launch carrying a HIP event pair that receives the
    # kernel's begin / end timestamps on its launch stream (comparable with rocprofv3's kernel trace, profiles/)
    eng.profile(1)
    run(32 * CHUNK)
    prof = eng.profile(0)
    full = prof["step_launches"]
    avg_ms = prof["step_ms"] / max(full, 1)
    alg = algorithmic_bytes_per_launch(B, N, D, CHUNK, actions=record)
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
    if os.path.exists(tpath) and B == 65536 and N == 3 and record and CHUNK == 80:  # (the PMC passes ran this very launch shape)
        traffic = json.load(open(tpath)).get("k_step_bytes_per_launch")

    if rank == 0:
        episodes = float(allstats[:, 1].sum())
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 parallel 3-player games",
            "value": steps_total / t_max,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": f"{B} parallel {N}-player games per GPU, random admissible policy on device, "
                                   f"{'direct' if args.direct_obs else 'DEFAULT_CONFIG (indirect'} obs D={D}{'' if args.direct_obs else ')'}, auto-reset, records+actions written every step"
                                   if record else f"{B} x {N}-player games per GPU, no records",
                       "games_per_gpu": B, "num_players": N, "rng_mode": args.rng,
                       "iterations_per_launch": CHUNK, "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "mean_episode_len": float(allstats[:, 2].sum()) / max(episodes, 1.0),
            "episodes": episodes,
            "waits": float(allstats[:, 4].sum()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step<indirect,policy>", "avg_launch_ms": avg_ms, "launches_timed": full,
                         "algorithmic_bytes_per_launch": alg,
                         "deal_kernel_avg_ms": prof["deal_ms"] / max(prof["deal_launches"], 1)},
        }
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: a reported baseline, not part of the scaling runs)
            out["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
