"""The C-ABI library loads on a CPU-only box and exports every symbol include/columba_amd.h declares.
No compute calls here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

import columba_amd as ca

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    return ca.build_library()


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "columba_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cmb_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(ca.EXPORTS), declared ^ set(ca.EXPORTS)
    L = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_means_loud_failure(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from columba_amd import indexbuild as ib
    ix = ib.build_index(b"ACGTACGTTTGACA" * 20)
    with pytest.raises(ca.CmbError) as e:
        ca.Index(ix)
    assert e.value.code == -2  # CMB_ERR_DEVICE: no CPU fallback


def test_strategy_tables_host_side(built_lib):
    # host-only part of the boundary: scheme validation and critical parts (search.h:525-588)
    s = ca.SearchStrategy("multiple_opt")
    assert s.describe(4) == (3, 5, [0, 2, 4])
    assert s.describe(2)[0:2] == (2, 3)
    s6 = s.describe(6)
    assert s6[0] == 4 and s6[1] == 7
    k = ca.SearchStrategy("kuch1", "hamming", "dynamic")
    assert k.describe(2)[0:2] == (1, 3)
    with pytest.raises(ca.CmbError):
        ca.SearchStrategy("no_such_scheme")
    with pytest.raises(ca.CmbError):
        s.describe(3)
    # invalid scheme: connectivity violated
    with pytest.raises(ca.CmbError):
        ca.SearchStrategy.from_tables({"schemes": {2: [[([0, 2, 1], [0, 0, 0], [0, 1, 2])]]}})


def test_header_is_plain_c():
    """include/columba_amd.h is the C-ABI: it must compile as C99 on its own (no C++ or torch types in the signatures),
    and the adapters on top of it as C++17."""
    import shutil
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "columba_amd.h")])
    for h in ("columba_amd.hpp", "columba_amd_io.hpp"):
        src = f'#include "{h}"\nint main() {{ return 0; }}\n'
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", "-I", inc, "-"], input=src,
                       text=True, check=True)
