"""The C ABI library must load on a machine without a GPU and export every symbol that
include/skyjo_vec.h declares (no compute call is made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "skyjo_vec.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skyjo_(?:vec|dev)_[a-z_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from skyjo_rl_amd import build

    return ctypes.CDLL(build.build())


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/skyjo_vec.h but not exported"


def test_python_signature_table_matches_header():
    from skyjo_rl_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()


def test_struct_sizes_match_header(tmp_path):
    """ctypes mirrors of the ABI structs have the sizes the C compiler gives the header's."""
    import subprocess

    from skyjo_rl_amd import _lib

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "skyjo_vec.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                   "sizeof(skyjo_vec_config),sizeof(skyjo_vec_info),sizeof(skyjo_vec_counters),"
                   "sizeof(skyjo_game_state));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(_lib.Config), ctypes.sizeof(_lib.Info), ctypes.sizeof(_lib.Counters),
                     ctypes.sizeof(_lib.GameState)]


def test_no_gpu_means_loud_failure():
    """Without a usable gfx950 device the product raises - it never falls back to a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from skyjo_rl_amd import SkyjoNativeError, SkyjoVecEnv

    with pytest.raises(SkyjoNativeError):
        SkyjoVecEnv(4)


def test_product_does_not_import_oracle():
    """Nothing under skyjo_rl_amd/ may reference the oracle."""
    pkg = os.path.join(ROOT, "skyjo_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "skyjo_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
