"""Build libskyjo_vec.so (HIP, gfx950) in-tree.  `python -m skyjo_rl_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HEADERS = sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")) + [os.path.join(ROOT, "include", "skyjo_vec.h")]
# Two translation units, each with the instruction scheduler that suits it (EXPERIMENTS.md round 5 #11, round 6): the
# environment kernels gain 1 - 2 % under max-ilp, the policy net's hand-placed MFMA gaps want the default strategy.
UNITS = [("skyjo_capi", os.path.join(CSRC, "skyjo_capi.hip"), ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         # (the net kernels are straight-line code: their gap loops must unroll completely, whatever the size)
         ("skyjo_policy", os.path.join(CSRC, "skyjo_policy.hip"), ["-mllvm", "-unroll-threshold=100000", "-mllvm", "-pragma-unroll-threshold=1000000"])]
SRC = UNITS[0][1]
DEPS = [u[1] for u in UNITS] + HEADERS
OUT = os.path.join(HERE, "libskyjo_vec.so")
OBJ_DIR = os.path.join(HERE, "_obj")  # (object files of the two units; git-ignored, not sent to the GPU box)

# -ffp-contract=off: rewards are float64 and must round exactly like numpy (no fused multiply-add); the policy unit shares the
# masked draw's float32 arithmetic with the environment unit (skyjo_draw.h) and must round like it
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=(), out=None, jobs=2):
    """Compile the units (in parallel) and link them into `out` (default: libskyjo_vec.so next to this file)."""
    out = out or OUT
    if not force and out == OUT and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = os.path.splitext(os.path.basename(out))[0]
    procs, objs = [], []
    for name, src, unit_flags in UNITS:
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (tag, name))
        cmd = [hipcc] + FLAGS + list(unit_flags) + list(extra) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
        if len(procs) >= jobs:
            c, pr = procs.pop(0)
            if pr.wait():
                raise subprocess.CalledProcessError(pr.returncode, c)
    for c, pr in procs:
        if pr.wait():
            raise subprocess.CalledProcessError(pr.returncode, c)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


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
    # (-shared-libsan: the runtime comes in as a shared library - UBSAN_RUNTIME below - which the process that loads this library
    # through ctypes has to preload)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    for name, src, unit_flags in UNITS:
        obj = os.path.join(OBJ_DIR, "ubsan.%s.o" % name)
        cmd = [hipcc, "-O1", "-g"] + [f for f in FLAGS if f != "-O3"] + ["-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
                                                                        "-fno-sanitize-recover=undefined", "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=undefined", "-shared-libsan", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--ubsan-host" in sys.argv:
        print(build_ubsan_host(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
