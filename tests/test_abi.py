"""CPU: the C-ABI library loads and exports every symbol include/xsw.h declares; no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from xsarsea_amd import _build, _lib


def header_functions():
    txt = open(os.path.join(REPO, "include", "xsw.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xsw_[a-z_0-9]+)\s*\(", txt)))


def test_library_is_built_for_gfx950():
    _build.build()
    assert os.path.exists(_build.LIB)
    blob = open(_build.LIB, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded in libxsw.so"


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/xsw.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert lib.xsw_version() == 4


def test_struct_layouts_match_header():
    # xsw_lut: 9 pointers + 3 int32 (padded to 8) ; xsw_invert_args: see header
    assert ctypes.sizeof(_lib.LutStruct) == 9 * 8 + 16
    assert ctypes.sizeof(_lib.InvertArgs) == 2 * 8 + 6 * 4 + 5 * 8 + 2 * 8 + 3 * 8 + 2 * 8 + 2 * 8
    assert ctypes.sizeof(_lib.Stats) == 32


def test_fails_loudly_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.XswError, match="no HIP device"):
        _lib.Context(0)
    import xsarsea_amd
    with pytest.raises(_lib.XswError):
        xsarsea_amd.sigma0_detrend(np.ones((4, 4)), np.full((4, 4), 30.0))
