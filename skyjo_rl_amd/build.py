"""Build libskyjo_vec.so (HIP, gfx950) in-tree.  `python -m skyjo_rl_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "skyjo_capi.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "skyjo_device.h"), os.path.join(HERE, "csrc", "skyjo_layout.h"),
        os.path.join(HERE, "csrc", "skyjo_policy.h"),
        os.path.join(ROOT, "include", "skyjo_vec.h")]
OUT = os.path.join(HERE, "libskyjo_vec.so")

# -ffp-contract=off: rewards are float64 and must round exactly like numpy (no fused multiply-add)
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wall"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
