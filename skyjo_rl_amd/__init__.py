"""skyjo_rl_amd - MI355X-native vectorised SkyJo environment (drop-in for rlskyjo's env hot path).

Public surface:
  SkyjoVecEnv            batched engine over libskyjo_vec.so (HIP kernels for gfx950)
  env / SimpleSkyjoEnv   PettingZoo-AEC compatible single-game view (rlskyjo.environment.skyjo_env)
  SkyjoGame              core-API compatible single-game view (rlskyjo.game.skyjo)
  policy_ra              rlskyjo.models.random_admissible_policy.policy_ra
"""
from ._lib import (RNG_MT19937, RNG_PHILOX, ST_ILLEGAL, ST_NOOP_DONE, ST_OK, ST_RESET,  # noqa: F401
                   SkyjoNativeError)
from .vec_env import SkyjoVecEnv  # noqa: F401
from .aec_env import DEFAULT_CONFIG, SimpleSkyjoEnv, env  # noqa: F401
from .game import SkyjoGame  # noqa: F401
from .policy import policy_ra  # noqa: F401

__version__ = "0.1.0"
