"""The oracle's restatement of the on-device random admissible policy (oracle/skyjo_oracle.c: sko_policy_action - Philox word,
k = mulhi(word, n_legal), k-th legal action) against the statistics of the REFERENCE's policy_ra loop
(tests/golden/policy_stats.npz, oracle/gen_golden.py: gen_policy_stats; rlskyjo/models/random_admissible_policy.py:26-28).
CPU only; the GPU engine is held to the same fixture in tests/test_gpu_policy_stats.py."""
import os

import numpy as np
import pytest

from tests import policy_stats_checks as psc


def collect_from_oracle(N, B, chunks, K, threads):
    from oracle import skyjo_oracle as so

    ora = so.OracleVec(num_envs=B, num_players=N, score_penalty=2.0, observe_other_player_indirect=True, mean_reward=1.0,
                       reward_refunded=0.001, rng_mode=so.RNG_MT19937, auto_reset=True)
    ora.seed(None, 5)
    c = psc.Candidate(N)
    _, prev_mask, _, _ = ora.observe()
    for _ in range(chunks):
        act, obs, mask, meta, eplen = ora.rollout(K, 3, threads=threads, record_obs=True)
        before = np.concatenate([prev_mask[None], mask[:-1]], axis=0) != 0   # the mask each action was drawn from
        acted = act >= 0
        n_legal = before.sum(axis=2)
        rank = (before & (np.arange(26)[None, None, :] < act[..., None])).sum(axis=2)
        np.add.at(c.rank_counts, (n_legal[acted], rank[acted]), 1)
        ended = acted & (meta[..., 2] != 0)
        c.len_hist += np.bincount(eplen[ended], minlength=len(c.len_hist))
        c.episodes += int(ended.sum())
        c.sum_len += int(eplen[ended].astype(np.int64).sum())
        prev_mask = mask[-1]
    c.sum_score, c.sum_refunded = ora.seat_sums()
    return c


@pytest.mark.parametrize("N", [2, 3, 4])
def test_oracle_policy_matches_the_reference_statistics(N):
    threads = min(8, os.cpu_count() or 1)
    B, K = 2048, 128
    chunks = {2: 40, 3: 56, 4: 72}[N]   # ~ 65 episodes per game: > 130 000 episodes (T ~ 5 000 .. 9 000: censoring bias var / T ~ 0.03)
    c = collect_from_oracle(N, B, chunks, K, threads)
    report = psc.check_against_reference(c)
    assert abs(report["episode length"][2]) < psc.SIGMAS
