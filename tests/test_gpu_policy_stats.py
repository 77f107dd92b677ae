"""The ON-DEVICE random admissible policy (skyjo_device.h: policy_pick, fused into k_step) against the statistics of the
REFERENCE's policy_ra loop (tests/golden/policy_stats.npz - written by oracle/gen_golden.py from the imported reference;
rlskyjo/models/random_admissible_policy.py:26-28).  VERDICT r3 "weak" #2: the bit-exact tests compare the device's pick with the
oracle's restatement of the same formula; this one pins the formula's DISTRIBUTION to the reference's.  What is compared and
with which bounds: tests/policy_stats_checks.py.  16 384 games x 8 192 iterations (~ 1.3 million episodes per case), every record of the rollout is digested on the device."""
import numpy as np
import pytest

from tests import policy_stats_checks as psc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,rng_mode", [(2, 0), (3, 0), (4, 0), (3, 1)], ids=["N2", "N3_headline_shape", "N4", "N3_philox_deals"])
def test_on_device_policy_matches_the_reference_statistics(N, rng_mode):
    import torch
    from skyjo_rl_amd import SkyjoVecEnv

    # A run of T iterations only counts the episodes that END inside it, which favours short ones (the episode in progress
    # at the end is the censored one): the mean is low by about var / T.  T = 8 192 makes that 0.02 .. 0.04 steps, a fraction of
    # the fixture's standard error (0.16 .. 0.25), and ~ 100 episodes per game leave no trace of the common start either.
    B, K = 16384, 64
    launches = 128
    eng = SkyjoVecEnv(B, num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0, reward_refunded=0.001,
                      rng_mode=rng_mode, auto_reset=True)
    eng.seed(None, 17)
    c = psc.Candidate(N)
    dev = torch.device("cuda", 0)
    rank_counts = torch.zeros(27 * 26, dtype=torch.int64, device=dev)
    len_hist = torch.zeros(len(c.len_hist), dtype=torch.int64, device=dev)
    prev_mask = eng.split(eng.observe()).action_mask.clone()   # the masks the first iteration's actions are drawn from
    rec = eng.new_records(K)
    ar = torch.arange(26, device=dev)
    for _ in range(launches):
        eng.rollout(K, policy_seed=23, records=rec)
        v = eng.split(rec)
        act = v.action.to(torch.int64)                                           # [K, B]; -1: nothing applied (a re-deal)
        before = torch.cat([prev_mask[None], v.action_mask[:-1]], dim=0) != 0    # [K, B, 26]
        acted = act >= 0
        n_legal = before.sum(dim=2)
        rank = (before & (ar[None, None, :] < act[..., None])).sum(dim=2)
        assert bool(torch.gather(before, 2, act.clamp(min=0)[..., None])[..., 0][acted].all()), "an action outside its mask"
        rank_counts += torch.bincount((n_legal * 26 + rank)[acted], minlength=27 * 26)
        ended = acted & (v.done != 0)
        len_hist += torch.bincount(v.episode_steps[ended].to(torch.int64), minlength=len(c.len_hist))
        prev_mask = v.action_mask[-1].clone()
    cnt = eng.counters()
    c.rank_counts = rank_counts.cpu().numpy().reshape(27, 26)
    c.len_hist = len_hist.cpu().numpy()
    c.episodes, c.sum_len = int(cnt["episodes"]), int(cnt["sum_len"])
    c.sum_score, c.sum_refunded = np.asarray(cnt["sum_score"], dtype=np.float64), np.asarray(cnt["sum_refunded"], dtype=np.float64)
    assert cnt["illegal"] == 0 and c.episodes > 50 * B
    assert int(c.rank_counts.sum()) == int(cnt["steps"])
    report = psc.check_against_reference(c)
    print(N, rng_mode, {k: (tuple(round(x, 4) for x in v) if isinstance(v, tuple) else round(v, 4)) for k, v in report.items()})
    eng.close()
