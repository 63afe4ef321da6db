import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def default_luts():
    """Default-resolution CMOD5.N + S1-v2 dB LUTs built by the oracle (as the goldens were)."""
    from oracle import lut as olut
    return olut.to_lut("gmf_cmod5n"), olut.to_lut("gmf_s1_v2")


@pytest.fixture(scope="session")
def lowres_luts():
    from oracle import lut as olut
    return olut.to_lut("gmf_cmod5n", resolution="low"), olut.to_lut("gmf_s1_v2", resolution="low")


@pytest.fixture(scope="session")
def gpu_ctx():
    from xsarsea_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X box")
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture
def xr_env(monkeypatch):
    """xarray + dask.array for the container tests: the real packages when installed, else the duck-typed stand-ins of
    tests/xr_standin.py bound into the product modules (which hold `xr = None` / `da = None` when the imports failed)."""
    import types
    try:
        import dask.array as da
        import xarray as xr
        standin = False
    except ImportError:
        import xr_standin
        xr, da, standin = xr_standin.make_xarray_module(), xr_standin.make_dask_array_module(), True
    import xsarsea_amd.detrend as pdet
    from xsarsea_amd.windspeed import gmfs, lut, models, windspeed
    for mod in (pdet, gmfs, lut, models, windspeed):
        monkeypatch.setattr(mod, "xr", xr)
    monkeypatch.setattr(windspeed, "da", da)
    return types.SimpleNamespace(xr=xr, da=da, standin=standin)
