"""Identity-JIT stand-in for numba (fixture generation only, THIS container only).

Test infrastructure, not product code.  The reference's own "numba is optional"
fallback is broken (rlskyjo/game/skyjo.py:13-16 defines ``njit(fastmath)`` but applies
``@njit()`` at :77/:91), so importing the reference needs a module named ``numba``.
This stand-in reproduces ``numba.config.DISABLE_JIT = True`` semantics, which is the mode
the reference's seeded test pins (tests/environment/test_skyjo_env_jit.py:1-2).
"""


class _Cfg:
    DISABLE_JIT = True


config = _Cfg()


def njit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
