"""Shared checks: a random-admissible-policy implementation against tests/golden/policy_stats.npz.

The fixture holds 6 000 games per player count of the REFERENCE's own loop (SkyjoGame + policy_ra,
rlskyjo/models/random_admissible_policy.py:26-28: choice(arange(26), p=mask/sum(mask))), written by
oracle/gen_golden.py.  The on-device policy and the oracle's restatement of it draw from another
random stream by construction, so they are held to the reference's DISTRIBUTIONS:

  * variance of the episode length: the ratio to the fixture's sample variance within 4.5 standard errors of a sample variance,
  * mean episode length, mean num_refunded and mean final score per seat: within 4 standard errors
    (the fixture's sample variance / its 6 000 games + the candidate's own, which is far smaller),
  * the histogram of episode lengths and the histogram of "how many legal actions did the turn have":
    chi-square of the fixture's counts against the candidate's proportions, p > 1e-4,
  * the k-th-legal-action pick is uniform: for every number of legal actions n, chi-square of the
    candidate's rank counts against n equal cells, p > 1e-6 each (a few dozen cells are tested);
    the fixture's own counts pass the same test.

`Candidate` is what a run has to hand in: episodes, sum of lengths, per-seat sums, a length histogram and the
[27][26] table counts[n_legal][rank].
"""
import os

import numpy as np
from scipy import stats

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_stats.npz")
SIGMAS = 4.0


class Candidate:
    def __init__(self, N):
        self.N = N
        self.episodes = 0
        self.sum_len = 0
        self.sum_score = np.zeros(N)
        self.sum_refunded = np.zeros(N)
        self.len_hist = np.zeros(4096, dtype=np.int64)
        self.rank_counts = np.zeros((27, 26), dtype=np.int64)
        # optional second moments (when absent the fixture's variance stands in: same distribution under H0)
        self.sum_len_sq = None


def _chi2_p(observed, proportions):
    """Goodness of fit of `observed` counts to `proportions` (cells with expectation < 5 are pooled into one)."""
    observed = np.asarray(observed, dtype=np.float64)
    expected = np.asarray(proportions, dtype=np.float64) / np.sum(proportions) * observed.sum()
    small = expected < 5.0
    if small.any():
        observed = np.concatenate([observed[~small], [observed[small].sum()]])
        expected = np.concatenate([expected[~small], [expected[small].sum()]])
        if expected[-1] == 0.0:
            assert observed[-1] == 0.0, "the fixture has outcomes the candidate never produced"
            observed, expected = observed[:-1], expected[:-1]
    chi2 = float(((observed - expected) ** 2 / expected).sum())
    return float(stats.chi2.sf(chi2, len(observed) - 1)), chi2, len(observed) - 1


def check_uniform_ranks(rank_counts, min_total=2000, p_min=1e-6, who="candidate"):
    tested = 0
    for n in range(2, 27):
        row = rank_counts[n, :n]
        assert rank_counts[n, n:].sum() == 0, f"{who}: a rank beyond the number of legal actions"
        if row.sum() < min_total:
            continue
        p, chi2, dof = _chi2_p(row, np.ones(n))
        assert p > p_min, f"{who}: picks among {n} legal actions are not uniform (chi2 {chi2:.1f}, dof {dof}, p {p:.2e}): {row}"
        tested += 1
    assert tested >= 8, f"{who}: too few cells with data ({tested})"
    assert rank_counts[0].sum() == 0 and rank_counts[1, 1:].sum() == 0
    return tested


def check_against_reference(c: Candidate):
    N = c.N
    g = np.load(GOLD)
    assert N in g["players"]
    ep_len, score, refunded, ref_ranks = g[f"N{N}_ep_len"], g[f"N{N}_score"], g[f"N{N}_refunded"], g[f"N{N}_rank_counts"]
    n_ref, n_c = len(ep_len), c.episodes
    assert n_c >= 20 * n_ref, "the candidate's sample should dwarf the fixture's"
    report = {}

    def mean_check(name, cand_mean, ref_sample, cand_var=None):
        var_ref = float(np.var(ref_sample, ddof=1))
        se = np.sqrt(var_ref / n_ref + (cand_var if cand_var is not None else var_ref) / n_c)
        d = cand_mean - float(np.mean(ref_sample))
        report[name] = (cand_mean, float(np.mean(ref_sample)), d / se if se > 0 else 0.0)
        assert abs(d) <= SIGMAS * se + 1e-12, f"{name}: candidate {cand_mean:.5f} vs reference {np.mean(ref_sample):.5f} +- {se:.5f} ({d / se:+.2f} sigma)"

    lens = np.arange(len(c.len_hist))
    assert c.len_hist.sum() == n_c and int((c.len_hist * lens).sum()) == c.sum_len
    var_len = float((c.len_hist * lens.astype(np.float64) ** 2).sum() / n_c - (c.sum_len / n_c) ** 2)
    mean_check("episode length", c.sum_len / n_c, ep_len, var_len)
    # (the variance of the lengths themselves, as a ratio: chi-square interval of the fixture's sample variance)
    # Var(s^2) = sigma^4 (2 / (n - 1) + excess kurtosis / n): the lengths are skewed, so the kurtosis term counts
    ratio = float(np.var(ep_len, ddof=1)) / var_len
    kurt = float(stats.kurtosis(ep_len, fisher=True, bias=False))
    se_ratio = np.sqrt(2.0 / (n_ref - 1) + max(kurt, 0.0) / n_ref)
    lo, hi = 1.0 - (SIGMAS + 0.5) * se_ratio, 1.0 + (SIGMAS + 0.5) * se_ratio
    assert lo < ratio < hi, f"variance of the episode length: reference / candidate = {ratio:.3f} outside [{lo:.3f}, {hi:.3f}]"
    report["episode length variance ratio"] = ratio
    mean_check("refunds per episode", float(c.sum_refunded.sum()) / n_c, refunded.sum(axis=1))
    for p in range(N):
        mean_check(f"final score, seat {p}", float(c.sum_score[p]) / n_c, score[:, p])
        mean_check(f"num_refunded, seat {p}", float(c.sum_refunded[p]) / n_c, refunded[:, p])
    # episode-length histogram: the fixture's counts against the candidate's proportions, in bins of ~5 % candidate mass
    cdf = np.cumsum(c.len_hist) / n_c
    edges = np.unique(np.searchsorted(cdf, np.linspace(0.05, 0.95, 19)))
    bins_c = np.add.reduceat(c.len_hist, np.concatenate([[0], edges + 1]))
    ref_hist = np.bincount(ep_len, minlength=len(c.len_hist))[:len(c.len_hist)]
    assert len(np.bincount(ep_len)) <= len(c.len_hist)
    bins_r = np.add.reduceat(ref_hist, np.concatenate([[0], edges + 1]))
    p, chi2, dof = _chi2_p(bins_r, bins_c)
    report["episode length histogram p"] = p
    assert p > 1e-4, f"episode-length histogram differs from the reference's (chi2 {chi2:.1f}, dof {dof}, p {p:.2e})"
    # how many legal actions a turn has (the game's dynamics under the policy)
    p, chi2, dof = _chi2_p(ref_ranks.sum(axis=1), c.rank_counts.sum(axis=1))
    report["legal-action-count histogram p"] = p
    assert p > 1e-4, f"histogram of the number of legal actions differs from the reference's (chi2 {chi2:.1f}, dof {dof}, p {p:.2e})"
    # the pick among them is uniform - candidate and fixture alike
    check_uniform_ranks(c.rank_counts, who="candidate")
    check_uniform_ranks(ref_ranks, min_total=500, who="reference fixture")
    return report
