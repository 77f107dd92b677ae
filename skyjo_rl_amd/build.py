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


def ubsan_runtime():
    """clang's shared UBSan runtime (to LD_PRELOAD next to libskyjo_vec_ubsan.so), or None."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so"))
    return hits[-1] if hits else None


def build_ubsan_host(verbose=False):
    """The same library with the HOST half under UndefinedBehaviorSanitizer (argument validation, state packing, snapshot
    bookkeeping: `-Xarch_host -fsanitize=undefined`; the device code is built as usual - GPU sanitizers are not available on
    this pool).  Diagnostic build for tests/test_sanitizers.py: libskyjo_vec_ubsan.so, loaded through SKYJO_LIB."""
    out = os.path.join(HERE, "libskyjo_vec_ubsan.so")
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS if os.path.exists(d)):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # (-shared-libsan: the runtime comes in as a shared library - UBSAN_RUNTIME below - which the process that loads this library
    # through ctypes has to preload)
    cmd = [hipcc, "-O1", "-g"] + [f for f in FLAGS if f != "-O3"] + ["-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
                                                                    "-fno-sanitize-recover=undefined", "-shared-libsan", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--ubsan-host" in sys.argv:
        print(build_ubsan_host(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
