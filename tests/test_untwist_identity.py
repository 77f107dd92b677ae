"""The identity behind mt_untwist (skyjo_rl_amd/csrc/skyjo_device.h): MT19937's in-place regeneration can be run
backwards, so a pre-dealt episode is taken back without a log of the words it overwrote.  Restated here on numpy's
own legacy state (the stream the reference draws from, rlskyjo/game/skyjo.py:84-94) as a property test of the
arithmetic; the device code itself is covered by tests/test_gpu_parity.py::test_deal_cadence_does_not_change_results."""
import numpy as np
import pytest

N, M, MAG = 624, 397, 0x9908B0DF


def regenerate(mt, i):
    y = (mt[i] & 0x80000000) | (mt[(i + 1) % N] & 0x7FFFFFFF)
    mt[i] = mt[(i + M) % N] ^ (y >> 1) ^ (MAG if y & 1 else 0)


def untwist(mt, first, end):
    i = end
    while i != first:
        i = (i - 1) % N
        t = mt[i] ^ mt[(i + M) % N]
        odd = t >> 31
        if odd:
            t ^= MAG
        y = ((t << 1) & 0xFFFFFFFF) | odd
        mt[(i + 1) % N] = (mt[(i + 1) % N] & 0x80000000) | (y & 0x7FFFFFFF)
        mt[i] = y & 0x80000000


def _check(mt, first, count):
    ref = list(mt)
    i = first
    for _ in range(count):
        regenerate(mt, i)
        i = (i + 1) % N
    untwist(mt, first, i)
    diff = [k for k in range(N) if mt[k] != ref[k]]
    assert diff in ([], [first]) and all((mt[k] ^ ref[k]) < 2 ** 31 for k in diff)  # only dead low bits may differ
    a, b, j = list(mt), list(ref), first
    for _ in range(2000):  # the stream continues identically
        regenerate(a, j), regenerate(b, j)
        assert a[j] == b[j]
        j = (j + 1) % N


@pytest.mark.parametrize("first,count", [(0, 400), (600, 300), (300, 623), (10, 1), (5, 0), (100, 560)])
def test_untwist_restores_the_stream(first, count):
    mt = [int(x) for x in np.random.RandomState(5).get_state()[1]]
    for i in range(N):  # one full pass: every word has been made by the recurrence
        regenerate(mt, i)
    _check(mt, first, count)


def test_untwist_of_the_first_deal_after_seeding():
    _check([int(x) for x in np.random.RandomState(5).get_state()[1]], 0, 400)  # a freshly seeded stream starts at element 0


def test_nested_deals_are_taken_back_newest_first():
    mt = [int(x) for x in np.random.RandomState(9).get_state()[1]]
    for i in range(N):
        regenerate(mt, i)
    ref = list(mt)
    cuts, i = [37], 37
    for n in (410, 395, 402, 388):  # four pre-dealt episodes, more than two turns of the state
        for _ in range(n):
            regenerate(mt, i)
            i = (i + 1) % N
        cuts.append(i)
    for k in range(len(cuts) - 1, 0, -1):
        untwist(mt, cuts[k - 1], cuts[k])
    assert [k for k in range(N) if mt[k] != ref[k]] in ([], [37])
    assert (mt[37] ^ ref[37]) < 2 ** 31
