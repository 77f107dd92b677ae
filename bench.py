#!/usr/bin/env python3
"""bench.py - env-steps/sec of the vectorised SkyJo hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|4] [--blocks R] [--no-other-configs]

A bench "step" is ONE FUSED LAUNCH of the hot path over the whole batch: `iterations_per_step` lockstep
iterations (1 024 at the headline size: sixteen dealing cycles of 64 in one launch of k_cycle) - in each of them
every live game applies one action chosen by the on-device random admissible policy (state transition +
observation / action-mask build, the 64-byte record with the applied action written to HBM) and finished games
take their next deal - with the dealing runs of those cycles inside the same launch (its dealing wavefronts).
So `--steps 20 --warmup 5` are 20 480 timed lockstep iterations after 5 120 untimed ones (about 190 episodes per
game inside a timed block).  Before the warm-up the freshly seeded games are run for 100 launches (set-up:
seeded together they end their first episodes together, EXPERIMENTS.md round 2).  Records are written in the
tile-planar layout wherever that kernel exists (`--record-layout`; the row-major form of the same launch is
`other_configs.row_major_records_65536x3`).

Workload (`--config 3`, the default): BASELINE.json configs[2], 65 536 parallel 3-player games per GPU
(weak scaling under --gpus N), DEFAULT_CONFIG (indirect observation, D = 31), game g seeded base + g,
numpy-legacy MT19937 deals (bit-identical to the reference).  `--config 4`: BASELINE.json configs[3],
32 768 games per GPU - 262 144 in total at --gpus 8 (the shard a rank runs at N < 8 is the same size).

Timing: W untimed steps, then `--blocks` (5) timed blocks of EXACTLY K steps each, every block bracketed
by a barrier + torch.cuda.synchronize() on both sides and taken as the MAX over ranks.  `value` /
`ms_per_step` are the MEDIAN block (env-steps counted on device over all ranks / that block's time);
`blocks` lists every block with min / max.  Inputs are resident in HBM.  One JSON line on stdout (rank 0).

`--gpus N` (N > 1) without a torchrun environment starts the N ranks itself: a fresh
`python -m torch.distributed.run` child, started before this process touches the GPU.  With more ranks
than GPUs the run is refused unless SKYJO_BENCH_SHARED_GPU=1 (rehearsal: ranks share cards, gloo carries
the statistics record).

Extra objects on the same line:
  roofline       dominant kernel (k_cycle: step AND dealing wavefronts of sixteen dealing cycles) timed with HIP events
                 on its launch stream
  roofline_path  the whole path (= that one kernel in the one-kernel form; k_step + k_deal per dealing cycle in the older
                 forms): kernel-time sum from the same events, and the wall time of the timed region
  episode_stats  what the ranks all-gather (RCCL): per-seat reward / score statistics
  other_configs  (N = 1) short blocks of the other BASELINE.json configurations in the same run: cfg2 (4 096 x 2),
                 the cfg4 shard (32 768 x 3, game_id0 = 3 * 32 768), cfg5 (65 536 x 4, the action-mask model's policy +
                 value net in the loop, float32-grade and bf16), the counter-based RNG mode (also at 131 072 games: two
                 rounds of workgroups), the direct observation
  cpu_baseline   the CPU oracle (oracle/, a port of the reference's algorithm) timed on host cores, with
                 speedup_vs_cpu_port and speedup_vs_reference_constant beside it
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (the guide's figure without sparsity)
CHUNK = int(os.environ.get("SKYJO_BENCH_CHUNK", "0"))  # lockstep iterations per kernel launch; 0 = the engine's dealing interval
CYCLES_PER_LAUNCH = int(os.environ.get("SKYJO_BENCH_CYCLES", "16"))  # dealing cycles per launch of the one-kernel form (k_cycle): the ABI's maximum
SETTLE = int(os.environ.get("SKYJO_BENCH_SETTLE", "100"))  # launches between seeding and the warm-up (see main)
# the reference's own Python loop (core loop + policy_ra, N = 3, indirect observation), measured in the
# build container (8-core Xeon 2.1 GHz); the reference cannot travel to the GPU box, so these are constants
REFERENCE_STEPS_PER_S_1_CORE = 8.3e3
REFERENCE_STEPS_PER_S_8_CORES = 52.8e3
TRAFFIC_PROFILE = os.path.join("profiles", "r6_hbm_traffic.json")  # rocprofv3 PMC passes of this very launch shape (tools/refresh_profiles.sh)
# (the environment's translation unit: skyjo_device.h and its parts; the policy net's unit - skyjo_policy.* - has nothing to do with k_cycle's traffic)
KERNEL_SOURCES = ("skyjo_rl_amd/csrc/skyjo_device.h", "skyjo_rl_amd/csrc/skyjo_rng.h", "skyjo_rl_amd/csrc/skyjo_transition.h",
                  "skyjo_rl_amd/csrc/skyjo_record.h", "skyjo_rl_amd/csrc/skyjo_step.h", "skyjo_rl_amd/csrc/skyjo_deal.h",
                  "skyjo_rl_amd/csrc/skyjo_cycle.h", "skyjo_rl_amd/csrc/skyjo_callers.h", "skyjo_rl_amd/csrc/skyjo_draw.h",
                  "skyjo_rl_amd/csrc/skyjo_capi.hip", "skyjo_rl_amd/csrc/skyjo_layout.h")


def _code_only(text):
    """C++ source without comments and with runs of white space collapsed (string / character literals are kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":  # a literal: copy through to its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_source_sha256():
    """sha256 over the kernel sources' CODE (comments and white space do not count): tools/collect_profiles.py stores it with the
    PMC traffic it digests, and the traffic figure is only reported for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "r", errors="replace") as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()


def committed_traffic(shape):
    """(dominant kernel's bytes per launch, dealing kernel's fabric bytes per run or None, source note) from the committed PMC digest
    - or (None, None, reason) when the digest is missing, was taken for another launch shape (`shape`: games, players, iterations per
    launch, rng, dealing form, records), or for other kernel sources than the ones built."""
    tpath = os.path.join(ROOT, TRAFFIC_PROFILE)
    if not os.path.exists(tpath):
        return None, None, f"{TRAFFIC_PROFILE} not found"
    t = json.load(open(tpath))
    if t.get("shape") != shape:
        return None, None, f"no PMC digest for this launch shape (the committed one: {t.get('shape')})"
    if t.get("kernel_source_sha256") != kernel_source_sha256():
        return None, None, (f"{TRAFFIC_PROFILE} was measured on other kernel sources (sha256 {str(t.get('kernel_source_sha256'))[:16]} != "
                            f"{kernel_source_sha256()[:16]}): re-run tools/refresh_profiles.sh + tools/collect_profiles.py")
    fab = t.get("k_deal_fabric") or {}
    deal = fab.get("read_bytes", 0) + fab.get("write_bytes", 0) if fab else None
    return t.get("k_step_bytes_per_launch"), deal, (f"{TRAFFIC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_* passes of this launch shape on these "
                                                     f"kernel sources, tools/refresh_profiles.sh; not measured in this run)")


def launch_shape(B, N, chunk, rng, form, records, direct, layout="row-major"):
    return {"games": int(B), "players": int(N), "iterations_per_launch": int(chunk), "rng": rng, "dealing": form, "records": bool(records), "direct_obs": bool(direct),
            "record_layout": layout}


def pick_layout(eng, asked, indirect):
    """'tile-planar' wherever the kernel exists (the one-kernel form of the fused rollout) unless row-major is asked for."""
    if asked == "auto":
        return "tile-planar" if eng.dealing_form() == "one kernel" else "row-major"
    return asked


def algorithmic_bytes_per_launch(B, N, D, iters, records=True):
    """packed state S(N) = 24N + 150 + 16 read once and written once per launch, plus per env-step the
    record the path must emit: D obs + 26 mask + 2 (agent, done) + 1 (the action as int8, byte D of the record)."""
    S = 24 * N + 150 + 16
    per_step = (D + 26 + 2 + 1) if records else 0
    return B * (2 * S + iters * per_step)


def rng_outputs_per_deal(N):
    """Expected MT19937 outputs one deal consumes: shuffle(150), shuffle(150 - 12 N), N x permutation(12), each
    step i of a legacy shuffle drawing until (u32 & mask) <= i (mask = next power of two - 1): (mask + 1) / (i + 1) draws."""
    def shuffle(n):
        return sum((1 << i.bit_length()) / (i + 1) for i in range(1, n))
    return shuffle(150) + shuffle(150 - 12 * N) + N * shuffle(12)


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


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a fresh child (this process has not touched the
    GPU and never will), pass its one JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


ENV_CFG = dict(score_penalty=2.0, mean_reward=1.0, reward_refunded=0.001, auto_reset=True)


def time_rollout(eng, chunk, launches, rec, sync):
    """`launches` fused launches of `chunk` lockstep iterations; returns (seconds, env-steps applied, counters)."""
    sync()
    eng.reset_counters()
    sync()
    t0 = time.perf_counter()
    for _ in range(launches):
        eng.rollout(chunk, policy_seed=1, records=rec)
    sync()
    dt = time.perf_counter() - t0
    return dt, eng.counters()


def side_rollout_config(name, B, N, steps, warmup, device, rng_mode, indirect=True, game_id0=0, settle=100, layout="auto"):
    """One of the other BASELINE configurations as a short block in the same process: value, time per lockstep iteration, the
    dominant kernel's launch time (HIP events) and its roofline fraction (SURVEY 8d bytes of that shape)."""
    import torch
    from skyjo_rl_amd import SkyjoVecEnv

    eng = SkyjoVecEnv(B, num_players=N, observe_other_player_indirect=indirect, device=device, rng_mode=rng_mode, game_id0=game_id0,
                      **ENV_CFG)
    eng.seed(None, 0)
    mult = CYCLES_PER_LAUNCH if eng.dealing_form() == "one kernel" else 1  # (k_cycle: several dealing cycles per launch)
    chunk = eng.deal_interval() * mult
    layout = pick_layout(eng, layout, indirect)
    if layout == "tile-planar":
        eng.set_record_layout("tile-planar")
    new_rec = eng.new_planar_records if layout == "tile-planar" else eng.new_records
    rec = new_rec(chunk)
    sync = torch.cuda.synchronize
    for _ in range(settle + warmup):
        eng.rollout(chunk, policy_seed=1, records=rec)
    chunk2 = eng.deal_interval() * mult  # (the interval adapts itself while the banks settle)
    if chunk2 != chunk:
        chunk = chunk2
        rec = new_rec(chunk)
    runs = sorted((time_rollout(eng, chunk, steps, rec, sync) for _ in range(3)), key=lambda x: x[0])
    dt, c = runs[1]  # the median of three blocks
    eng.profile(1)
    for _ in range(16):
        eng.rollout(chunk, policy_seed=1, records=rec)
    prof = eng.profile(0)
    k_ms = prof["step_ms"] / max(prof["step_launches"], 1)
    alg = algorithmic_bytes_per_launch(B, N, eng.obs_dim, chunk)
    out = {"workload": f"{B} x {N}-player games, {'indirect' if indirect else 'direct'} observation D={eng.obs_dim}, "
                       f"{'mt19937' if rng_mode == 0 else 'philox'} deals" + (f", game_id0={game_id0}" if game_id0 else ""),
           "value": c["steps"] / dt, "unit": "env-steps/s", "ms_per_iteration": 1e3 * dt / (steps * chunk),
           "iterations_per_launch": chunk, "timed_launches": steps,
           "dominant_kernel": (f"k_cycle<{'indirect' if indirect else 'direct'},{N}> ({mult} dealing cycles per launch, the dealing runs inside)"
                               if eng.dealing_form() == "one kernel" else "k_step"),
           "dominant_kernel_ms": k_ms,
           "deal_kernel_ms": None if eng.dealing_form() == "one kernel" else prof["deal_ms"] / max(prof["deal_launches"], 1),
           "roofline_frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
           "dealing": eng.dealing_form(), "record_layout": layout, "waits": int(c["waits"]),
           "mean_episode_len": c["sum_len"] / max(c["episodes"], 1)}
    eng.close()
    return out


def side_model_config(precision, B, N, T, rounds, device):
    """BASELINE configs[4]: 65 536 four-player games, every action drawn by the action-mask model (policy + value branch in one
    launch, random weights), collected by skyjo_vec_model_rollout - two launches per lockstep iteration."""
    import torch
    from skyjo_rl_amd import SkyjoVecEnv
    from skyjo_rl_amd.action_mask_model import ActionMaskModel, FusedNet
    from skyjo_rl_amd.rollout import RolloutBuffer, collect

    torch.manual_seed(0)
    env = SkyjoVecEnv(B, num_players=N, device=device, **ENV_CFG)
    layout = os.environ.get("SKYJO_BENCH_CFG5_LAYOUT", "row-major")  # ("tile-planar-all": the step kernel writes / the nets read planar blocks)
    if layout != "row-major":
        env.set_overlap(3)
        env.set_record_layout(layout)
    env.seed(None, 3)
    model = ActionMaskModel(obs_dim=env.obs_dim).cuda()
    pol, val = FusedNet(model.policy, device=device, precision=precision), FusedNet(model.value, device=device, precision=precision)
    buf = RolloutBuffer(env, T)
    collect(env, pol, val, buf, seed=9, first_ticket=0)  # warm-up (incl. the first episodes' common end)
    collect(env, pol, val, buf, seed=9, first_ticket=T, first_records=buf.records[T].clone())
    per_round = []  # (seconds, env-steps) of every timed round: the median one is reported (a round is ~3 ms: one hiccup of the
    for r in range(rounds):  # host - an allocation, a page fault - would otherwise be the result)
        first = buf.records[T].clone()
        torch.cuda.synchronize()
        env.reset_counters()
        t0 = time.perf_counter()
        collect(env, pol, val, buf, seed=9, first_ticket=(2 + r) * T, first_records=first)
        torch.cuda.synchronize()
        per_round.append((time.perf_counter() - t0, env.counters()))
    per_round.sort(key=lambda x: x[0])
    dt, c = per_round[len(per_round) // 2]
    dt *= rounds  # (the fields below are written for `rounds` rounds of the median round's duration)
    c = dict(c, steps=c["steps"] * rounds)
    env.profile(1)
    collect(env, pol, val, buf, seed=9, first_ticket=(2 + rounds) * T, first_records=buf.records[T].clone())
    prof = env.profile(0)
    mlp_ms = prof["k_mlp_ms"] / max(prof["k_mlp_launches"], 1)
    flops = 2.0 * B * 2 * (32 * 256 + 256 * 256 + 256 * 32)  # policy + value net, the model's own (float32) multiply-adds
    out = {"workload": f"{B} x {N}-player games, action-mask model (policy + value net, 256-256 tanh, random weights) picks every action; "
                       f"skyjo_vec_model_rollout: 2 launches per lockstep iteration, rollout columns written",
           "precision": precision, "record_layout": layout, "value": c["steps"] / dt, "unit": "env-steps/s", "ms_per_iteration": 1e3 * dt / (rounds * T),
           "timed_iterations": rounds * T, "dominant_kernel": "k_net_split" if precision == "fp32" else "k_net_bf16",
           "dominant_kernel_ms": mlp_ms, "step_kernel_ms": prof["step_ms"] / max(prof["step_launches"], 1),
           "roofline_bound": "mfma", "roofline_achieved_tflops": flops / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else None,
           "roofline_frac": flops / (mlp_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if mlp_ms > 0 else None,
           "mfma_issue_frac": (3.0 if precision == "fp32" else 1.0) * flops / (mlp_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if mlp_ms > 0 else None,
           "illegal": int(c["illegal"]), "mean_episode_len": c["sum_len"] / max(c["episodes"], 1),
           "round_ms": [1e3 * x[0] for x in per_round]}
    pol.close(), val.close(), env.close()
    return out


def other_configs(device, steps, warmup):
    out = {}
    t0 = time.perf_counter()
    from skyjo_rl_amd import RNG_MT19937, RNG_PHILOX
    for name, fn in (
            ("cfg2_4096x2", lambda: side_rollout_config("cfg2", 4096, 2, 4 * steps, warmup, device, RNG_MT19937)),
            ("cfg4_shard_32768x3", lambda: side_rollout_config("cfg4", 32768, 3, 2 * steps, warmup, device, RNG_MT19937, game_id0=3 * 32768)),
            ("cfg5_65536x4_model_fp32", lambda: side_model_config("fp32", 65536, 4, 64, 7, device)),
            ("cfg5_65536x4_model_bf16", lambda: side_model_config("bf16", 65536, 4, 64, 7, device)),
            ("philox_65536x3", lambda: side_rollout_config("philox", 65536, 3, steps, warmup, device, RNG_PHILOX)),
            # the headline batch with the default (row-major) record layout: what a caller gets who does not opt in to the tile-planar one
            ("row_major_records_65536x3", lambda: side_rollout_config("rowmajor", 65536, 3, steps, warmup, device, RNG_MT19937, layout="row-major")),
            ("direct_obs_65536x3", lambda: side_rollout_config("direct", 65536, 3, steps, warmup, device, RNG_MT19937, indirect=False)),
            # twice the metric's batch: two rounds of k_cycle workgroups (a workgroup's LDS fills its CU)
            ("philox_131072x3", lambda: side_rollout_config("philox131k", 131072, 3, steps, warmup, device, RNG_PHILOX))):
        try:
            out[name] = fn()
        except Exception as e:  # a side configuration must not take the headline line down with it
            out[name] = {"error": repr(e)}
    out["seconds"] = time.perf_counter() - t0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="fused launches (of `iterations_per_step` lockstep iterations each) per timed block")
    ap.add_argument("--warmup", type=int, default=25, help="untimed launches before the first block")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps launches each (value = the median block)")
    ap.add_argument("--config", type=int, choices=[3, 4], default=3,
                    help="3: 65 536 games per GPU (BASELINE configs[2], weak scaling); 4: 32 768 per GPU (configs[3]: 262 144 at --gpus 8)")
    ap.add_argument("--num-envs", type=int, default=None, help="games per GPU (overrides --config)")
    ap.add_argument("--total-games", type=int, default=None,
                    help="STRONG scaling: this many games in total, G / N per GPU (default: weak scaling, --config's games on every GPU)")
    ap.add_argument("--num-players", type=int, default=3)
    ap.add_argument("--rng", choices=["mt19937", "philox"], default="mt19937")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short blocks of the other BASELINE configurations")
    ap.add_argument("--no-records", action="store_true", default=bool(os.environ.get("SKYJO_BENCH_NO_RECORDS")), help="do not write records (not the headline)")
    ap.add_argument("--actions-array", action="store_true", help="also write the int32 action array (the action is byte D of every record anyway)")
    ap.add_argument("--record-layout", choices=["auto", "row-major", "tile-planar"], default=os.environ.get("SKYJO_BENCH_RECORD_LAYOUT", "auto"),
                    help="how the fused rollout lays out its records (include/skyjo_vec.h: SKYJO_OPT_RECORD_LAYOUT); auto = tile-planar "
                         "wherever the kernel exists (the one-kernel form, either observation), row-major otherwise")
    ap.add_argument("--direct-obs", action="store_true", help="observe_other_player_indirect=False: D = 19 + 12 N (not the headline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import torch
    import torch.distributed as dist

    from skyjo_rl_amd import RNG_MT19937, RNG_PHILOX
    from skyjo_rl_amd.distributed import gather_stats, make_sharded_env

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs a GPU"
    shared_gpu = world > ndev
    if shared_gpu and os.environ.get("SKYJO_BENCH_SHARED_GPU") != "1":
        # one rank per GPU over RCCL is the measured configuration; silently sharing cards would report a scaling number that is none
        sys.stderr.write(f"bench.py: --gpus {world} but only {ndev} GPU(s) visible - refusing to let ranks share a card.  "
                         f"(SKYJO_BENCH_SHARED_GPU=1 allows it as a REHEARSAL of the launch path: gloo carries the statistics record.)\n")
        sys.exit(3)
    device = local_rank % ndev
    # (a process group also for ONE rank when a launcher set the rendezvous up - tools/scale_run.sh 1: the N = 1 point of the scaling
    # table goes through the same RCCL all-gather as the others)
    if world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ):
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))  # RCCL
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)

    strong = args.total_games is not None
    if strong:
        assert args.num_envs is None and args.total_games % world == 0, "--total-games must be a multiple of --gpus (and excludes --num-envs)"
        B = args.total_games // world
    else:
        B = args.num_envs if args.num_envs else (65536 if args.config == 3 else 32768)
    N = args.num_players
    # shards by global game id (rank r owns games r*B .. (r+1)*B - 1): results do not depend on the GPU count
    eng = make_sharded_env(world * B, rank, world, num_players=N, observe_other_player_indirect=not args.direct_obs, device=device,
                           rng_mode=RNG_MT19937 if args.rng == "mt19937" else RNG_PHILOX, **ENV_CFG)
    assert eng.num_envs == B and eng.game_id0 == rank * B
    eng.seed(None, 0)
    global CHUNK
    CHUNK = min(CHUNK, eng.deal_interval()) if CHUNK > 0 else eng.deal_interval()  # one launch per dealing cycle ...
    if eng.dealing_form() == "one kernel" and CYCLES_PER_LAUNCH > 1:
        CHUNK = eng.deal_interval() * CYCLES_PER_LAUNCH  # ... or ONE launch of k_cycle over several (the tiles stay in LDS between them)
    D = eng.obs_dim
    record = not args.no_records
    planar = pick_layout(eng, args.record_layout, not args.direct_obs) == "tile-planar"
    if planar:
        eng.set_record_layout("tile-planar")
    rec = (eng.new_planar_records(CHUNK) if planar else eng.new_records(CHUNK)) if record else None  # [CHUNK, B, 64] ring reused by every launch
    act = torch.empty((CHUNK, B), dtype=torch.int32, device=dev) if args.actions_array else None

    def run(launches):
        for _ in range(launches):
            eng.rollout(CHUNK, policy_seed=1, records=rec, actions=act)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not part of W or K: freshly seeded, all games start their first episode in the same iteration and end it within
    # a few iterations of each other - bursts of resets and deals instead of the steady 0.95 % per iteration.  It takes some
    # 50 episodes per game until the ends are spread evenly (measured: 20 timed launches after 5 / 25 / 50 / 100 / 200 launches:
    # 2.91 / 2.92 / 3.01 / 3.05 / 3.04 x 10^10 steps/s), so the games are run for SETTLE launches before the warm-up starts.
    run(SETTLE)
    run(args.warmup)
    blocks = []
    for b in range(max(args.blocks, 1)):
        # Between two blocks the host reads counters and gathers statistics - 50 .. 200 us without a launch, after which the next
        # ~ 8 launches run up to 10 % slower (the kernel trace of profiles/r6_kernel_trace_digest.json: the clock comes back over
        # ~ 10 ms).  A block therefore starts like the first one does, behind untimed launches: W of them again, at most 8.
        if b > 0:
            run(min(args.warmup, 8))
        eng.reset_counters()
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        c1 = eng.counters()
        # every rank must have been in steady state: games ended and were re-dealt inside the timed region, nobody dealt in place
        assert c1["episodes"] > 0 and c1["resets"] > 0, "no episode ended inside the timed region: --steps too small to mean anything"
        assert c1["waits"] == 0 or os.environ.get("SKYJO_BENCH_ALLOW_WAITS"), f"{c1['waits']} deals were made on the in-kernel slow path (bank ran dry)"
        assert c1["iters"] == args.steps * CHUNK
        # the one collective of the path: per-rank episode statistics (counts + per-seat reward / score sums), all-gathered
        # over RCCL (device tensors; gloo when ranks share a card)
        c1["wall"] = dt
        per_rank, tot = gather_stats(c1, N, device=dev)
        # a silent single-rank fallback must be impossible: the gathered matrix has one row per rank, every row a full shard
        assert per_rank.shape[0] == world, f"gathered {per_rank.shape[0]} statistics records for {world} ranks"
        assert all(per_rank[r, 0] + per_rank[r, 3] == args.steps * CHUNK * B for r in range(world)), "a rank's steps + resets are not its shard's"
        if dist.is_initialized():
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_max = float(t.item())
        else:
            t_max = dt
        blocks.append({"t_max": t_max, "steps": float(tot["steps"]), "value": float(tot["steps"]) / t_max, "tot": tot, "per_rank": per_rank,
                       "c1": c1})
    order = sorted(range(len(blocks)), key=lambda i: blocks[i]["value"])
    med = blocks[order[len(order) // 2]]
    t_max, tot, per_rank, c1 = med["t_max"], med["tot"], med["per_rank"], med["c1"]
    steps_total = med["steps"]
    episodes = float(tot["episodes"])

    # roofline leg: the same launches again, every kernel carrying a HIP event pair that receives its begin / end
    # timestamps on its launch stream (comparable with rocprofv3's kernel trace, profiles/)
    eng.profile(1)
    run(32)
    prof = eng.profile(0)
    full = prof["step_launches"]
    avg_ms = prof["step_ms"] / max(full, 1)
    alg = algorithmic_bytes_per_launch(B, N, D, CHUNK, records=record) + (4 * B * CHUNK if act is not None else 0)
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # NOT measured in this run: PMC counters need rocprofv3 around the process.  The figure is the committed digest of this very
    # launch shape, and only if it was taken on the kernel sources that are built here (sha256 stored with it).
    shape = launch_shape(B, N, CHUNK, args.rng, eng.dealing_form(), record and act is None, args.direct_obs, "tile-planar" if planar else "row-major")
    traffic, deal_traffic, traffic_source = committed_traffic(shape)
    path_traffic = traffic if (traffic is not None and eng.dealing_form() == "one kernel") else (
        traffic + deal_traffic if traffic is not None and deal_traffic is not None else None)  # (k_cycle's counters already hold both roles)
    kernel_ms = {k: prof[k + "_ms"] / 32.0 for k in ("k_step", "k_scan", "k_deal", "k_publish")}  # per bench step = per LAUNCH (k_cycle: CYCLES_PER_LAUNCH dealing cycles, slot "k_step")
    path_ms = sum(kernel_ms.values())
    wall_ms = 1e3 * t_max / args.steps
    # the dealing kernel's own algorithmic traffic (MT19937 mode): per RNG output one state word read as the old element, one
    # as element i + 397, one written (12 B), plus the 24N + 150 + 16 B record, for every deal consumed in the timed region
    deals_per_step = float(c1["resets"]) / args.steps
    deal_alg = deals_per_step * ((24 * N + 150 + 16) + (12.0 * rng_outputs_per_deal(N) if args.rng == "mt19937" else 0.0))

    if rank == 0:
        vals = [b["value"] for b in blocks]
        if strong:
            scaling_note = f"strong: {args.total_games} games in total, {B} per GPU"
        elif args.num_envs is None and args.config == 4:
            scaling_note = f"BASELINE configs[3]: 32 768 games per GPU ({world * B} in total; 262 144 at --gpus 8)"
        else:
            scaling_note = f"weak: {B} games per GPU ({world * B} in total)"
        backend = dist.get_backend() if dist.is_initialized() else "none (single rank)"
        try:
            nccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())  # RCCL's version on ROCm
        except Exception as e:  # (reported, not needed by a single rank)
            nccl_version = f"unavailable: {e!r}"
        out = {
            "metric": "env-steps/sec (whole node) at 65 536 parallel 3-player games",
            "value": med["value"],
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_ms,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": f"{B} parallel {N}-player games per GPU, random admissible policy on device, "
                                   f"{'direct' if args.direct_obs else 'DEFAULT_CONFIG (indirect'} obs D={D}{'' if args.direct_obs else ')'}, auto-reset, "
                                   f"record (obs + mask + applied action) written every step ({'tile-planar' if planar else 'row-major'} layout); one bench step = one fused launch of {CHUNK} lockstep iterations, its dealing runs inside"
                                   if record else f"{B} x {N}-player games per GPU, no records",
                       "baseline_config": args.config if args.num_envs is None else None,
                       "games_per_gpu": B, "games_total": world * B, "scaling_mode": "strong (--total-games)" if strong else "weak (per-GPU batch fixed)",
                       "num_players": N, "rng_mode": args.rng, "record_layout": "tile-planar" if planar else "row-major",
                       "collective": {"backend": backend, "rccl_version": nccl_version, "ranks_gathered": int(per_rank.shape[0]),
                                      "world_size": world, "op": "all_gather of one float64 statistics record per rank and timed block"},
                       "iterations_per_step": CHUNK, "timed_iterations_per_block": args.steps * CHUNK, "warmup_iterations": args.warmup * CHUNK,
                       "settle_launches_before_warmup": SETTLE,
                       "dealing": {"one kernel": "one kernel per dealing cycle (k_cycle): one step + one dealing wavefront per SIMD, hand-over inside the CU",
                                   "two streams": "beside k_step (own stream; planned and published by k_step itself)",
                                   "in line": "in line (k_deal scans the banks itself)"}[eng.dealing_form()],
                       "shared_gpu_rehearsal": shared_gpu,
                       "parallelism": f"{scaling_note}; games sharded over {world} GPU(s) by global game id, no data-path collective; "
                                      f"one all-gather of the statistics record ({'gloo: ranks share a card' if shared_gpu else 'RCCL' if world > 1 else 'single rank'})"},
            "blocks": {"n": len(blocks), "steps_per_block": args.steps, "values": vals, "median": med["value"], "min": min(vals), "max": max(vals),
                       "rel_spread": (max(vals) - min(vals)) / med["value"], "ms_per_step": [1e3 * b["t_max"] / args.steps for b in blocks]},
            "ms_per_iteration": wall_ms / CHUNK,
            "mean_episode_len": float(tot["mean_episode_len"]),
            "episodes": episodes,
            "waits": float(tot["waits"]),
            "episode_stats": {"record_doubles_per_rank": int(per_rank.shape[1]), "ranks": int(per_rank.shape[0]),
                              "mean_reward_per_seat": [float(x) for x in tot["mean_reward"]],
                              "std_reward_per_seat": [float(x) for x in tot["std_reward"]],
                              "mean_score_per_seat": [float(x) for x in tot["mean_score"]],
                              "refunded_per_episode": float(tot["sum_refunded"].sum() / max(episodes, 1.0))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": (f"k_cycle<{'direct' if args.direct_obs else 'indirect'},{N},{'planar' if planar else 'row-major'}> (step + dealing wavefronts of "
                                    f"{CHUNK // max(eng.deal_interval(), 1)} dealing cycles)" if eng.dealing_form() == "one kernel" else
                                    f"k_step<{'indirect' if not args.direct_obs else 'direct'},policy,{N if N in (2, 3, 4) else 0}>"), "avg_launch_ms": avg_ms,
                         "launches_timed": full, "algorithmic_bytes_per_launch": alg, "launch_shape": shape,
                         "deal_kernel_avg_ms": prof["deal_ms"] / max(prof["deal_launches"], 1)},
            "roofline_path": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                              "kernel_ms_per_step": kernel_ms, "kernel_ms_sum": path_ms,
                              "achieved_kernel_time": alg / (path_ms * 1e-3) / 1e9 if path_ms > 0 else 0.0,
                              "frac_kernel_time": alg / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if path_ms > 0 else 0.0,
                              "wall_ms_per_step": wall_ms,
                              "traffic": path_traffic, "traffic_k_deal_fabric": deal_traffic, "traffic_source": traffic_source,
                              "traffic_over_algorithmic": path_traffic / alg if path_traffic is not None else None,
                              "achieved_wall": alg / (wall_ms * 1e-3) / 1e9,
                              "frac_wall": alg / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "k_deal": {"deals_per_step": deals_per_step, "rng_outputs_per_deal": rng_outputs_per_deal(N) if args.rng == "mt19937" else None,
                                         "algorithmic_bytes_per_step": deal_alg,
                                         "achieved": deal_alg / (kernel_ms["k_deal"] * 1e-3) / 1e9 if kernel_ms["k_deal"] > 0 else 0.0,
                                         "frac": deal_alg / (kernel_ms["k_deal"] * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms["k_deal"] > 0 else 0.0},
                              "achieved_wall_incl_dealing_bytes": (alg + deal_alg) / (wall_ms * 1e-3) / 1e9,
                              "frac_wall_incl_dealing_bytes": (alg + deal_alg) / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "achieved_* / frac_*: algorithmic bytes of one k_step launch (SURVEY 8d: state once + records) over the time of one "
                                      "whole dealing cycle; *_incl_dealing_bytes adds the generator-state and record bytes the dealing kernel itself "
                                      "has to move (not part of SURVEY 8d's per-step figure); with the dealing kernel on its own stream the kernel times "
                                      "overlap and only the wall figures are path times"},
        }
    eng.close()
    if rank == 0:
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(device, max(args.steps, 10), max(args.warmup, 5))
        out["speedup_vs_reference_constant"] = {
            "vs_1_core": med["value"] / REFERENCE_STEPS_PER_S_1_CORE, "vs_8_cores": med["value"] / REFERENCE_STEPS_PER_S_8_CORES,
            "reference_steps_per_s": {"1_core": REFERENCE_STEPS_PER_S_1_CORE, "8_cores": REFERENCE_STEPS_PER_S_8_CORES},
            "source": "BASELINE.md section 2: the reference's Python core loop + policy_ra (N = 3, indirect observation), measured in the build "
                      "container (8-core Xeon 2.1 GHz) - the reference cannot run on the GPU box"}
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only: a reported baseline, not part of the scaling runs)
            out["cpu_baseline"] = cpu_baseline(N)
            out["speedup_vs_cpu_port"] = {"vs_all_host_threads": med["value"] / out["cpu_baseline"]["value"],
                                          "vs_1_thread": med["value"] / out["cpu_baseline"]["value_1_thread"]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
