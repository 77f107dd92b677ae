from . import wrappers  # noqa: F401
