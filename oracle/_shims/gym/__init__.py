"""Constructor-only stand-in for gym==0.21.0 spaces (fixture generation only)."""
from . import spaces  # noqa: F401
