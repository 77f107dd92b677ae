"""Multi-GPU sharding: games are independent, so a batch shards by global game id with no
data-path collective; the only exchange is an all-gather of per-rank episode statistics
(RCCL when the tensors live on the GPU: torch.distributed backend "nccl" is RCCL on ROCm).

One process per GPU.  Game g is seeded ``base_seed + g`` and its policy stream is keyed by g, so
trajectories do not depend on how many ranks the batch is split over.
"""
import numpy as np

STAT_FIELDS = ("steps", "episodes", "illegal", "resets", "sum_len", "reshuffles", "waits")
SEAT_FIELDS = ("sum_score", "sum_reward", "sum_reward_sq", "sum_refunded")  # float64 [num_players] each


def shard_range(total_envs, world_size, rank):
    """Contiguous block of games owned by `rank`: (first global game id, number of games)."""
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def make_sharded_env(total_envs, rank, world_size, engine_factory=None, **config):
    """Local engine for this rank's shard of a `total_envs`-game batch."""
    first, count = shard_range(total_envs, world_size, rank)
    if engine_factory is None:
        from .vec_env import SkyjoVecEnv as engine_factory
    return engine_factory(count, game_id0=first, **config)


def stats_record(counters, num_players):
    """Fixed-size float64 record [len(STAT_FIELDS) + len(SEAT_FIELDS) * num_players] of one rank:
    {steps, episodes, illegal, resets, sum of episode lengths, ...} + per seat {score, reward, reward^2, refunded} sums."""
    rec = [float(counters.get(k, 0)) for k in STAT_FIELDS]
    for key in SEAT_FIELDS:
        v = counters.get(key)
        rec += [float(x) for x in (v if v is not None else np.zeros(num_players))][:num_players]
    return rec


def combine_stats(per_rank, num_players):
    """Totals of the gathered records [W, F] (one row per rank, stats_record's layout): what every rank holds after the
    all-gather.  Sums are taken in rank order, so the result does not depend on which rank computes it."""
    allr = np.asarray(per_rank, dtype=np.float64)
    tot = allr.sum(0)
    totals = {k: tot[i] for i, k in enumerate(STAT_FIELDS)}
    n = len(STAT_FIELDS)
    for j, key in enumerate(SEAT_FIELDS):
        totals[key] = tot[n + j * num_players:n + (j + 1) * num_players]
    totals["mean_episode_len"] = totals["sum_len"] / max(totals["episodes"], 1.0)
    ep = max(totals["episodes"] + totals["illegal"], 1.0)
    totals["mean_reward"] = totals["sum_reward"] / ep                       # per seat, over finished + illegal episodes
    totals["std_reward"] = np.sqrt(np.maximum(totals["sum_reward_sq"] / ep - totals["mean_reward"] ** 2, 0.0))
    totals["mean_score"] = totals["sum_score"] / max(totals["episodes"], 1.0)
    return totals


def gather_stats(counters, num_players, device=None):
    """All-gather the per-rank statistics record; returns (per_rank [W, F] ndarray, totals dict).

    Whenever a process group is initialised the record goes through ``dist.all_gather`` - also with ONE rank (a single-rank RCCL
    communicator is a real communicator: tests/test_gpu_rccl.py runs exactly that on a one-GPU box, the only place where the
    "nccl" leg of this function can execute before an 8-GPU node is at hand).  Without a process group the record stays local."""
    import torch
    import torch.distributed as dist

    rec = torch.tensor(stats_record(counters, num_players), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size()
        on = _backend_devices(dist.get_backend())
        kind = "cuda" if rec.is_cuda else "cpu"
        if kind not in on:
            # the group has no backend for where the record lives: take it to where the group works, or - one rank, nothing to
            # exchange - leave it where it is (a single-rank "nccl" group with a host record used to assert)
            if world == 1 and kind == "cpu":
                allr = rec.numpy()[None]
                return allr, combine_stats(allr, num_players)
            if "cpu" in on:
                rec = rec.cpu()
            elif torch.cuda.is_available():
                rec = rec.to(torch.device("cuda", torch.cuda.current_device()))
            else:
                raise ValueError(f"gather_stats: the process group's backend ({dist.get_backend()}) needs a device tensor and no GPU is visible")
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)
        allr = torch.stack(out).cpu().numpy()
    else:
        allr = rec.cpu().numpy()[None]
    return allr, combine_stats(allr, num_players)


def _backend_devices(backend):
    """Device types ("cpu", "cuda") the default group's backend string serves: "nccl" (= RCCL) -> cuda, "gloo" -> cpu (this path
    gathers host tensors through it), a mixed group "cpu:gloo,cuda:nccl" -> both, anything else (mpi, ucc) -> both."""
    b = str(backend).lower()
    if ":" in b:
        return {part.split(":", 1)[0] for part in b.split(",")}
    if b == "nccl":
        return {"cuda"}
    if b == "gloo":
        return {"cpu"}
    return {"cpu", "cuda"}
