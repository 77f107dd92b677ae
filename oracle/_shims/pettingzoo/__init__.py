"""Minimal stand-in for pettingzoo==1.14.0 (fixture generation only, THIS container only).

Test infrastructure, not product code.  pettingzoo is a pinned third-party dependency of the
reference (requirements.txt:4) that is absent from /root/reference and from this image.  The
classes below restate its published AEC bookkeeping from memory, so
every fixture produced through them is labelled "wrapper semantics unpinned".  Only
rlskyjo/environment/skyjo_env.py (reference code) running on top of these is authoritative.
"""
from .aec import AECEnv  # noqa: F401
