"""GPU parity: the HIP engine (through the C ABI) vs golden vectors from the reference and vs the
pinned CPU oracle on identical seeded inputs.  Integer / byte work: every comparison is bit-exact;
float64 scores and rewards are compared with == as well (same operation order, no FMA)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")) + glob.glob(os.path.join(GOLDEN, "dense_*.npz")))
REWARD_CFGS = [(1.0, 0.001), (0.0, 0.0), (-1.0, 0.01), (1.0, 0.0)]


def _engine(*a, **k):
    from skyjo_rl_amd import SkyjoVecEnv
    return SkyjoVecEnv(*a, **k)


def _cmp_state(s, d, prefix, e):
    np.testing.assert_array_equal(s["cards"], d[prefix + "cards"][e])
    np.testing.assert_array_equal(s["masked"], d[prefix + "masked"][e])
    assert s["n_draw"] == d[prefix + "n_draw"][e] and s["n_disc"] == d[prefix + "n_disc"][e]
    np.testing.assert_array_equal(s["draw"], d[prefix + "draw"][e][: s["n_draw"]])
    np.testing.assert_array_equal(s["disc"], d[prefix + "disc"][e][: s["n_disc"]])
    assert (s["hand"], s["player"], s["phase"]) == (d[prefix + "hand"][e], d[prefix + "player"][e], d[prefix + "phase"][e])


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[:-4] for p in TRAJ])
def test_golden_trajectory(path):
    """Reference trajectories replayed on the GPU: 3 copies of the game, every step bit-equal."""
    d = np.load(path)
    N, seed, ind = int(d["num_players"]), int(d["seed"]), bool(d["indirect"])
    dense = str(d["kind"]) == "dense"
    B = 3
    cfg_k = (len(os.path.basename(path)) + seed) % len(REWARD_CFGS)  # spread reward configs over the fixtures
    mr, rr = REWARD_CFGS[cfg_k]
    eng = _engine(B, num_players=N, score_penalty=float(d["score_penalty"]), observe_other_player_indirect=ind,
                  mean_reward=mr, reward_refunded=rr, auto_reset=False)
    eng.seed(np.full(B, seed, dtype=np.uint64))
    E = len(d["ep_start"]) - 1
    for e in range(E):
        if e > 0:
            eng.reset_host()
        if dense:
            nd, ns = int(d["deal_n_draw"][e]), int(d["deal_n_disc"][e])
            for i in range(B):
                eng.set_state(i, d["deal_cards"][e], d["deal_masked"][e], d["deal_draw"][e][:nd],
                              d["deal_disc"][e][:ns], int(d["deal_hand"][e]), int(d["deal_player"][e]),
                              int(d["deal_phase"][e]))
        _cmp_state(eng.get_state(B - 1), d, "deal_", e)
        o = eng.observe_host()
        for t in range(int(d["ep_start"][e]), int(d["ep_start"][e + 1])):
            pid = int(d["player"][t])
            for i in range(B):
                assert (o.agent[i], o.phase[i], o.done[i]) == (pid, d["phase"][t], 0), (t, i)
                np.testing.assert_array_equal(o.observations[i], d["obs"][t], err_msg=f"obs t={t}")
                np.testing.assert_array_equal(o.action_mask[i], d["mask"][t], err_msg=f"mask t={t}")
            if t % 7 == 0:  # observe() of a player who is not on turn (skyjo_env.py:199-214)
                oo = eng.observe_host(np.full(B, (pid + 1) % N, dtype=np.int32))
                np.testing.assert_array_equal(oo.observations[0], d["obs_other"][t])
                np.testing.assert_array_equal(oo.action_mask[0], d["mask_other"][t])
            o = eng.step_host(np.full(B, int(d["action"][t]), dtype=np.int32))
            assert np.all(o.status == 0)
            assert np.all(o.done == d["game_over"][t]), t
        s = eng.get_state(1)
        assert s["is_terminated"] and s["done"]
        _cmp_state(s, d, "end_", e)
        np.testing.assert_array_equal(s["final_score"], d["final_score"][e])
        np.testing.assert_array_equal(s["num_refunded"], d["num_refunded"][e])
        np.testing.assert_array_equal(s["num_placed"], d["num_placed"][e])
        np.testing.assert_array_equal(s["rewards"], d["rewards"][e][cfg_k])
        rew, sc, done = eng.rewards_host()
        np.testing.assert_array_equal(rew[0], d["rewards"][e][cfg_k])
        assert np.all(done == 1)
        # trap 18: stepping a finished game without auto-reset is a no-op (skyjo.py:316-321)
        o2 = eng.step_host(np.full(B, 24, dtype=np.int32))
        assert np.all(o2.status == 2) and np.all(o2.done == 1)
        _cmp_state(eng.get_state(1), d, "end_", e)
    eng.close()


def test_golden_scenarios():
    d = np.load(os.path.join(GOLDEN, "scenarios.npz"))
    for name in d["names"]:
        p = str(name) + "/"
        N, ind, np_seed = (int(x) for x in d[p + "cfg"])
        eng = _engine(2, num_players=N, score_penalty=float(d[p + "penalty"]), observe_other_player_indirect=bool(ind),
                      auto_reset=False)
        eng.seed(None, 5)
        nd, ns = int(d[p + "init_n_draw"]), int(d[p + "init_n_disc"])
        for i in range(2):
            eng.set_state(i, d[p + "init_cards"], d[p + "init_masked"], d[p + "init_draw"][:nd],
                          d[p + "init_disc"][:ns], int(d[p + "init_hand"]), int(d[p + "init_player"]),
                          int(d[p + "init_phase"]))
            if np_seed >= 0:
                eng.seed_raw(i, np_seed)
        o = eng.observe_host()
        for t, a in enumerate(d[p + "actions"]):
            np.testing.assert_array_equal(o.observations[1], d[p + "step_obs"][t], err_msg=f"{name} obs t={t}")
            np.testing.assert_array_equal(o.action_mask[1], d[p + "step_mask"][t], err_msg=f"{name} mask t={t}")
            o = eng.step_host(np.full(2, int(a), dtype=np.int32))
            assert np.all(o.done == d[p + "step_over"][t]), (name, t)
            s = eng.get_state(1)
            np.testing.assert_array_equal(s["cards"], d[p + "step_cards"][t], err_msg=f"{name} t={t}")
            np.testing.assert_array_equal(s["masked"], d[p + "step_masked"][t], err_msg=f"{name} t={t}")
            assert s["n_draw"] == d[p + "step_n_draw"][t] and s["n_disc"] == d[p + "step_n_disc"][t], (name, t)
            np.testing.assert_array_equal(s["draw"], d[p + "step_draw"][t][: s["n_draw"]], err_msg=f"{name} t={t}")
            np.testing.assert_array_equal(s["disc"], d[p + "step_disc"][t][: s["n_disc"]], err_msg=f"{name} t={t}")
            assert (s["hand"], s["player"], s["phase"]) == (
                d[p + "step_hand"][t], d[p + "step_player"][t], d[p + "step_phase"][t]), (name, t)
        np.testing.assert_array_equal(o.observations[0], d[p + "final_obs"], err_msg=str(name))
        np.testing.assert_array_equal(o.action_mask[0], d[p + "final_mask"], err_msg=str(name))
        s = eng.get_state(0)
        assert s["is_terminated"] == bool(d[p + "terminated"])
        np.testing.assert_array_equal(s["num_refunded"], d[p + "num_refunded"])
        np.testing.assert_array_equal(s["num_placed"], d[p + "num_placed"])
        if s["is_terminated"]:
            np.testing.assert_array_equal(s["final_score"], d[p + "final_score"])
            np.testing.assert_array_equal(s["rewards"], d[p + "rewards"][0])  # default reward cfg (1.0, 0.001)
        eng.close()


def _oracle_vec(**k):
    from oracle import skyjo_oracle as so
    return so.OracleVec(**k)


@pytest.mark.parametrize("N,ind,rng_mode,B", [(2, True, 0, 4096), (3, False, 0, 1000), (12, True, 0, 300),
                                              (4, True, 1, 1500), (1, False, 1, 200), (8, False, 1, 130)])
def test_step_vs_oracle_random_actions(N, ind, rng_mode, B):
    """Lockstep stepping with caller-provided actions (legal, some illegal), auto-reset on."""
    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=ind, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=rng_mode, auto_reset=True)
    eng = _engine(B, **cfg)
    ora = _oracle_vec(num_envs=B, **cfg)
    eng.seed(None, 1234)
    ora.seed(None, 1234)
    rng = np.random.default_rng(7)
    steps = 700 if N <= 4 else 1500
    for t in range(steps):
        obs, mask, agent, phase = ora.observe()
        o = eng.observe_host()
        np.testing.assert_array_equal(o.observations, obs, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(o.action_mask, mask, err_msg=f"mask t={t}")
        np.testing.assert_array_equal(o.agent, agent)
        np.testing.assert_array_equal(o.phase, phase)
        # uniformly random legal action, 0.2 % illegal ones
        u = rng.random((B, 26)) * mask
        acts = np.argmax(u, axis=1).astype(np.int32)
        bad = rng.random(B) < 0.002
        acts[bad] = rng.integers(-2, 29, size=int(bad.sum()))
        ora.step(acts)
        o = eng.step_host(acts)
        np.testing.assert_array_equal(o.status, ora.status, err_msg=f"status t={t}")
        np.testing.assert_array_equal(o.done, ora.dones, err_msg=f"done t={t}")
        rew, sc, done = eng.rewards_host()
        dn = ora.dones.astype(bool)
        np.testing.assert_array_equal(rew[dn], ora.rewards[dn], err_msg=f"rewards t={t}")
    c, oc = eng.counters(), ora.counters()
    for k in ("steps", "episodes", "illegal", "resets", "sum_len"):
        assert c[k] == oc[k], (k, c[k], oc[k])
    assert c["episodes"] > 0 and c["illegal"] > 0
    eng.close()


@pytest.mark.parametrize("N,ind,rng_mode,B", [(3, True, 0, 4096), (2, True, 0, 4096), (4, True, 1, 2048),
                                              (3, False, 1, 777), (12, True, 0, 128),
                                              # k_cycle with two / three step wavefronts per workgroup and a last workgroup that is not full
                                              # (258 tiles -> S = 2, 129 workgroups; 626 tiles -> S = 3, 209 workgroups, the last with two tiles)
                                              (3, True, 0, 16500), (2, True, 0, 40010)])
def test_rollout_vs_oracle(N, ind, rng_mode, B):
    """Fused K-step rollout kernel with the on-device policy vs the oracle's restatement of it."""
    import torch

    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=ind, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=rng_mode, auto_reset=True)
    eng = _engine(B, game_id0=1000, **cfg)
    ora = _oracle_vec(num_envs=B, game_id0=1000, **cfg)
    eng.seed(None, 99)
    ora.seed(None, 99)
    K, rounds = 40, (12 if B <= 4096 else 8) if N <= 4 else 30
    for r in range(rounds):
        rec = eng.new_records(K)
        act = torch.empty((K, B), dtype=torch.int32, device="cuda")
        eng.rollout(K, policy_seed=4242, records=rec, actions=act)
        oact = ora.rollout(K, 4242, record_actions=True, threads=8 if B > 4096 else 1)
        np.testing.assert_array_equal(act.cpu().numpy(), oact, err_msg=f"actions round {r}")
        last = eng.split(rec[K - 1])
        obs, mask, agent, phase = ora.observe()
        np.testing.assert_array_equal(last.observations.cpu().numpy(), obs)
        np.testing.assert_array_equal(last.action_mask.cpu().numpy(), mask)
        np.testing.assert_array_equal(last.agent.cpu().numpy(), agent)
        np.testing.assert_array_equal(last.done.cpu().numpy(), ora.dones)
        np.testing.assert_array_equal(last.status.cpu().numpy(), ora.status)
    c, oc = eng.counters(), ora.counters()
    for k in ("steps", "episodes", "illegal", "resets", "sum_len", "iters"):
        assert c[k] == oc[k] if k != "iters" else c[k] == oc["iter"], (k, c, oc)
    # "waits" counts deals made on the in-kernel slow path (pre-dealt episode invalidated by a reshuffle)
    assert c["illegal"] == 0 and c["episodes"] > B and (N > 4 or c["waits"] == 0)
    dn = ora.dones.astype(bool)
    rew, sc, done = eng.rewards_host()
    np.testing.assert_array_equal(rew[dn], ora.rewards[dn])
    eng.close()


@pytest.mark.parametrize("interval,overlap", [(1, False), (16, False), (1000, False), (8, True), (64, True), (8, 2), (64, 2), (8, 3), (40, 3)])
@pytest.mark.parametrize("N,rng_mode", [(2, 0), (3, 0), (4, 0), (12, 0), (1, 1), (1, 0), (8, 1)])
def test_deal_cadence_does_not_change_results(N, rng_mode, interval, overlap, options=None):
    """However rarely the dealing kernel runs (bank of pre-dealt episodes full, partly filled or empty - then the
    lane deals in place), whether it runs in line (overlap 0 / False), on its own stream beside the step kernels (2) or inside the
    step kernel's own workgroups (3: k_cycle; True = whichever of the two the engine prefers) - then the rare
    stream users wait for the one deal in flight for their game -, and however often a mid-game reshuffle rolls
    the stream back (N=12: ~17 per episode), every step equals the oracle's."""
    import torch

    if overlap == 3 and N not in (2, 3, 4):
        pytest.skip("the one-kernel form is compiled for two to four players")
    B = 192
    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=rng_mode, auto_reset=True)
    eng = _engine(B, **cfg)
    eng.set_deal_interval(interval)
    eng.set_overlap(overlap)
    for opt, val in (options or {}).items():
        eng.set_option(opt, val)
    assert eng.dealing_form() == {0: "in line", 2: "two streams", 3: "one kernel"}.get(int(overlap), eng.dealing_form())
    ora = _oracle_vec(num_envs=B, **cfg)
    eng.seed(None, 31)
    ora.seed(None, 31)
    K = 48
    for r in range(10):
        rec = eng.new_records(K)
        act = torch.empty((K, B), dtype=torch.int32, device="cuda")
        eng.rollout(K, policy_seed=9, records=rec, actions=act)
        oact = ora.rollout(K, 9, record_actions=True)
        np.testing.assert_array_equal(act.cpu().numpy(), oact, err_msg=f"round {r}")
        last = eng.split(rec[K - 1])
        obs, mask, agent, phase = ora.observe()
        np.testing.assert_array_equal(last.observations.cpu().numpy(), obs)
        np.testing.assert_array_equal(last.action_mask.cpu().numpy(), mask)
    c, oc = eng.counters(), ora.counters()
    for k in ("steps", "episodes", "resets", "sum_len"):
        assert c[k] == oc[k], (k, c[k], oc[k])
    if interval == 1000 and N <= 2:
        assert c["waits"] > 0  # the bank ran empty: episodes were dealt on the in-kernel slow path
    eng.close()


@pytest.mark.parametrize("N,rng_mode,interval", [(3, 0, 16), (2, 0, 1000), (12, 0, 8), (8, 1, 16)])
def test_inline_dealing_with_work_list_form(N, rng_mode, interval):
    """In line the dealing kernel scans the banks itself (lane = game); SKYJO_OPT_INLINE_WORK_LIST selects the k_scan + work list
    form the dealing run beside the step kernel uses, here in line: same results."""
    from skyjo_rl_amd import _lib
    test_deal_cadence_does_not_change_results(N, rng_mode, interval, False, options={_lib.OPT_INLINE_WORK_LIST: 1})


@pytest.mark.parametrize("N,rng_mode,interval", [(3, 0, 8), (2, 0, 64), (12, 0, 8), (8, 1, 64)])
def test_dealing_beside_the_step_kernel_with_scan_and_publish_kernels(N, rng_mode, interval):
    """Beside the step kernel the dealing runs are pipelined by default (the step kernel plans and publishes them itself);
    SKYJO_OPT_UNPIPELINED selects the k_scan / k_publish form whose caller's stream waits for every run: same results."""
    from skyjo_rl_amd import _lib
    test_deal_cadence_does_not_change_results(N, rng_mode, interval, True, options={_lib.OPT_UNPIPELINED: 1})


@pytest.mark.parametrize("seed,N,B", [(1, 3, 1024), (2, 2, 700), (3, 4, 2048)])
def test_mixed_calls_while_dealing_runs_are_in_flight(seed, N, B):
    """Every kind of call in a random order while dealing runs are pipelined beside the step kernel (small batch: on by
    default): fused rollouts of any length, single steps with caller actions, masked resets, re-seeding one game,
    snapshot / run on / restore, switching the dealing between the two streams and changing its interval.  After every
    call the engine shows what the oracle shows."""
    import torch

    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=0, auto_reset=True)
    eng = _engine(B, **cfg)
    ora = _oracle_vec(num_envs=B, **cfg)
    assert eng.overlap()
    eng.seed(None, 77)
    ora.seed(None, 77)
    rng = np.random.default_rng(seed)
    pol = 0

    def same(tag):
        obs, mask, agent, phase = ora.observe()
        o = eng.observe_host()
        np.testing.assert_array_equal(o.observations, obs, err_msg=tag)
        np.testing.assert_array_equal(o.action_mask, mask, err_msg=tag)
        np.testing.assert_array_equal(o.agent, agent, err_msg=tag)
        np.testing.assert_array_equal(o.phase, phase, err_msg=tag)

    for r in range(70):
        op = rng.choice(["rollout", "rollout", "rollout", "step", "reset", "seed_one", "snapshot", "overlap", "interval"])
        if op == "rollout":
            k = int(rng.integers(1, 120))
            pol += 1
            act = torch.empty((k, B), dtype=torch.int32, device="cuda")
            eng.rollout(k, policy_seed=pol, actions=act)
            np.testing.assert_array_equal(act.cpu().numpy(), ora.rollout(k, pol, record_actions=True), err_msg=f"round {r}")
        elif op == "step":
            for _ in range(int(rng.integers(1, 6))):
                obs, mask, agent, phase = ora.observe()
                acts = np.argmax(rng.random((B, 26)) * mask, axis=1).astype(np.int32)
                ora.step(acts)
                o = eng.step_host(acts)
                np.testing.assert_array_equal(o.status, ora.status, err_msg=f"round {r}")
                np.testing.assert_array_equal(o.done, ora.dones, err_msg=f"round {r}")
        elif op == "reset":
            m = (rng.random(B) < 0.05).astype(np.uint8)
            ora.reset(m)
            eng.reset_host(m)
        elif op == "seed_one":
            g, v = int(rng.integers(0, B)), int(rng.integers(0, 1 << 30))
            ora.seed_one(g, v)
            eng.seed_one(g, v)
        elif op == "snapshot":
            snap = eng.snapshot()
            eng.rollout(int(rng.integers(1, 60)), policy_seed=999)
            eng.restore(snap)
        elif op == "overlap":
            eng.set_overlap(int(rng.choice([0, 1, 2, 3])))  # in line / the engine's choice / two streams / one kernel (k_cycle)
        else:
            eng.set_deal_interval(int(rng.integers(4, 100)))
        same(f"round {r} after {op}")
    c, oc = eng.counters(), ora.counters()
    for k in ("steps", "episodes", "resets", "sum_len"):
        assert c[k] == oc[k], (k, c[k], oc[k])
    assert c["episodes"] > 0
    eng.close()


def test_headline_size_properties():
    """BASELINE config 3 (65 536 three-player games): size-independent invariants + an oracle-checked subset."""
    import torch

    B, N = 65536, 3
    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=0, auto_reset=True)
    eng = _engine(B, **cfg)
    eng.seed(None, 0)
    sub0, subn = 40000, 192  # games 40000..40191 are re-simulated by the oracle from their ids alone
    ora = _oracle_vec(num_envs=subn, game_id0=sub0, **cfg)
    ora.seed(None, 0)
    K = 64
    for r in range(6):
        rec = eng.new_records(K)
        act = torch.empty((K, B), dtype=torch.int32, device="cuda")
        eng.rollout(K, policy_seed=1, records=rec, actions=act)
        oact = ora.rollout(K, 1, record_actions=True)
        np.testing.assert_array_equal(act[:, sub0:sub0 + subn].cpu().numpy(), oact)
        v = eng.split(rec)
        # every mask is non-empty for live games and consistent with the phase byte
        m = v.action_mask.to(torch.int32)
        live = v.done == 0
        assert bool(((m.sum(-1) > 0) | ~live).all())
        draw = v.phase == 0
        assert bool((m[..., 24:].sum(-1)[draw & live] == 2).all()) and bool((m[..., :24].sum(-1)[draw & live] == 0).all())
        # observation bounds: Box(-24, 127) (skyjo_env.py:129-133), hand is 15 exactly in the draw phase
        o = v.observations.to(torch.int32)
        assert int(o.min()) >= -24
        assert bool(((o[..., 18] == 15) == draw)[live].all())
        # histogram of the discard pile never exceeds the deck: 10 per value, + 3 zeros per collapse
        assert int(o[..., 2:17].max()) <= 10 + 3 * 4 * N
    c = eng.counters()
    assert c["steps"] + c["resets"] == 6 * K * B and c["waits"] == 0 and c["illegal"] == 0 and c["iters"] == 6 * K
    assert 95 < c["sum_len"] / c["episodes"] < 120  # SURVEY: mean episode length 108 at N=3
    # finished episodes: sum of rewards over seats == N * mean_reward + refunded bonus (skyjo_env.py:307-312)
    # all games stay phase-aligned (draw on even iterations, a reset also takes one iteration), and a game can
    # only end on a draw: stop on an even iteration to catch finished games before their auto-reset
    eng.rollout(1, policy_seed=1)
    rew, sc, done = eng.rewards_host()
    dn = done.astype(bool)
    assert dn.sum() > 100
    resid = rew[dn].sum(1) - N * 1.0
    assert np.all(resid > -1e-9) and np.all(resid < 12 * 0.001 + 1e-9)
    # card conservation on a sample: cards on the table + piles + hand == 150 - 12N + 3 * collapses
    for g in (0, 1, 12345, 65535):
        s = eng.get_state(g)
        collapsed = int((s["masked"] == 0).sum()) // 3
        assert s["n_draw"] + s["n_disc"] + (s["hand"] != 15) == 150 - 12 * N + 3 * collapsed
    eng.close()


def test_dealing_interval_adapts_to_short_episodes():
    """One-player games end after ~25 steps: at the default interval of 64 the banks run dry and finished games deal in
    place (exact, slow).  The engine sees the empty banks in its dealing runs and shortens the interval by itself; the
    trajectories do not depend on it (same oracle comparison as everywhere)."""
    import torch

    B = 512
    cfg = dict(num_players=1, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0,
               reward_refunded=0.001, rng_mode=0, auto_reset=True)
    eng = _engine(B, **cfg)
    assert eng.overlap() and eng.deal_interval() == 56  # a small batch deals beside the step kernel (pipelined form), shorter interval
    eng.set_overlap(False)
    ora = _oracle_vec(num_envs=B, **cfg)
    eng.seed(None, 5)
    ora.seed(None, 5)
    assert eng.deal_interval() == 64
    early = late = 0
    for r in range(60):
        act = torch.empty((32, B), dtype=torch.int32, device="cuda")
        w0 = eng.counters()["waits"]
        eng.rollout(32, policy_seed=3, actions=act)
        oact = ora.rollout(32, 3, record_actions=True)
        np.testing.assert_array_equal(act.cpu().numpy(), oact, err_msg=f"actions round {r}")
        w = eng.counters()["waits"] - w0
        early += w if r < 10 else 0
        late += w if r >= 50 else 0
    # (the host reads the engine's report asynchronously, so how fast the interval comes down varies a little)
    assert eng.deal_interval() < 56
    assert early > 0 and late <= early / 2
    eng.close()


@pytest.mark.parametrize("form", [0, 3])
def test_rollout_without_auto_reset_runs_every_game_to_its_end(form):
    """auto_reset off under the fused rollout (in line and in the one-kernel form, several dealing cycles per launch): every game
    plays its first episode to the end and then stands still (status NOOP_DONE, its last record repeated), whatever the dealing
    wavefronts do beside it; every record equals the oracle's."""
    import torch

    B, N, K = 5000, 3, 256
    cfg = dict(num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001, rng_mode=0,
               auto_reset=False)
    eng = _engine(B, **cfg)
    eng.set_overlap(form)
    ora = _oracle_vec(num_envs=B, **cfg)
    eng.seed(None, 3)
    ora.seed(None, 3)
    rec = eng.new_records(K)
    eng.rollout(K, policy_seed=11, records=rec)
    oact, obs, mask, meta, eplen = ora.rollout(K, 11, threads=8, record_obs=True)
    v = eng.split(rec)
    np.testing.assert_array_equal(v.action.cpu().numpy(), oact.astype(np.int8))
    np.testing.assert_array_equal(v.observations.cpu().numpy(), obs)
    np.testing.assert_array_equal(v.action_mask.cpu().numpy(), mask)
    np.testing.assert_array_equal(v.done.cpu().numpy(), meta[..., 2])
    np.testing.assert_array_equal(v.status.cpu().numpy(), meta[..., 3])
    c = eng.counters()
    assert c["episodes"] == B and c["resets"] == 0 and bool((v.status[K - 1] == 2).all())  # SKYJO_ST_NOOP_DONE everywhere at the end
    rew, sc, done = eng.rewards_host()
    assert done.all()
    np.testing.assert_array_equal(rew, ora.rewards)
    eng.close()
