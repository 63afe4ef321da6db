"""CPU: the block-parallel host passes of the drop-in API return exactly the bits of the one-shot numpy expressions
of the reference (windspeed.py:107, :126-130, :417, :426-428)."""
import numpy as np
import pytest

from xsarsea_amd import _host
from xsarsea_amd.windspeed import _engine

N = 4 * _host.BLOCK + 12345  # just above the threshold that switches the block path on


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_to_db_blocks_bit_identical(dtype):
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.01, 2.0, N).astype(dtype)
    x[::1000] = np.nan
    x[5], x[6], x[7] = 0.0, np.inf, -1.0
    with np.errstate(all="ignore"):
        ref = 10 * np.log10(x + 1e-15)
    got = _engine._to_db(x)
    assert got.dtype == ref.dtype
    u = np.uint32 if dtype == np.float32 else np.uint64
    assert np.array_equal(ref.view(u), got.view(u))
    x2 = x[: 3000 * 1400].reshape(3000, 1400)
    with np.errstate(all="ignore"):
        assert np.array_equal(_engine._to_db(x2).view(u), (10 * np.log10(x2 + 1e-15)).view(u))


def test_dual_select_and_abs_blocks_bit_identical():
    rng = np.random.default_rng(4)
    a = rng.uniform(0, 10, N) * np.exp(1j * rng.uniform(-3, 3, N))
    b = rng.uniform(0, 10, N) * np.exp(1j * rng.uniform(-3, 3, N))
    a[::7] = np.nan
    b[::11] = complex(np.nan, 0)
    a[5], b[6] = 5.0, 3 + 4j  # |.| exactly 5: not < 5
    with np.errstate(all="ignore"):
        ref = np.where((np.abs(a) < 5) | (np.abs(b) < 5), a, b)
    got = _engine.dual_select(a, b)
    assert got.dtype == ref.dtype and np.array_equal(ref.view(np.uint64), got.view(np.uint64))
    assert np.array_equal(np.abs(a).view(np.uint64), _engine.abs_blocks(a).view(np.uint64))


def test_any_valid_early_exit_semantics():
    a = np.full(N, np.nan + 0j, dtype=np.complex64)
    assert not _engine.any_valid(a) and _engine.all_nan(a)
    a[-1] = 1
    assert _engine.any_valid(a) and not _engine.all_nan(a)
    assert _engine.any_valid(np.array([np.nan, 2.0])) and not _engine.any_valid(np.array([np.nan]))


def test_empty_touched_shape_dtype():
    out = _host.empty_touched((3000, 3000), np.complex128)  # 144 MB: above the touch threshold
    assert out.shape == (3000, 3000) and out.dtype == np.complex128 and out.flags.c_contiguous
    assert _host.empty_touched((4, 5), np.float32).shape == (4, 5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_in_place_piecewise_db_equals_the_one_shot_expression(dtype):
    """The staging callback of `_engine.invert_numpy` converts sigma0 -> dB piece by piece, in place, into a foreign buffer
    at arbitrary offsets (add, log10, multiply with out=): the bits must be those of `10 * np.log10(x + 1e-15)` on the whole
    raster whatever the piece boundaries and alignments (numpy's float32 log10 is a SIMD routine with masked tails)."""
    rng = np.random.default_rng(8)
    n = 300_007
    x = rng.uniform(-0.01, 2.0, n).astype(dtype)
    x[::997] = np.nan
    x[3], x[4], x[5] = 0.0, np.inf, -1.0
    with np.errstate(all="ignore"):
        ref = 10 * np.log10(x + 1e-15)
    u = np.uint32 if dtype == np.float32 else np.uint64
    buf = np.empty(n + 64, dtype=dtype)
    for shift in (0, 1, 3):  # destination alignment differs from the source's
        got = buf[shift:shift + n]
        edges = [0, 1, 17, 4096, 4099, 65536 + 5, 200_001, n]
        for a, b in zip(edges[:-1], edges[1:]):
            out = got[a:b]
            with np.errstate(all="ignore"):
                np.add(x[a:b], 1e-15, out=out)
                np.log10(out, out=out)
                np.multiply(out, 10, out=out)
        assert np.array_equal(ref.view(u), got.view(u)), (dtype, shift)


def test_ensure_luts_never_leaves_a_stale_key():
    """`ctx.lut_key` follows the context's tables install by install -- a co-pol install that succeeds followed by a
    cross-pol install that fails must not leave the OLD co-pol key in place (a later call with the old LUT object would skip
    the upload and search the wrong GMF); both LUTs' axes are validated before the context is touched."""
    from xsarsea_amd.windspeed import _engine

    class FakeCtx:
        def __init__(self):
            self.lut_key = (None, None)
            self.installed = []

        def upload_luts(self, co=None, cr=None):
            if cr is not None and cr.get("boom"):
                raise RuntimeError("cross-pol upload failed")
            self.installed.append("co" if co is not None else "cr")

    class L:  # stands in for a host Lut: _co_dict / _cr_dict are patched below
        def __init__(self, boom=False):
            self.boom = boom

    old_co, new_co, bad_cr = L(), L(), L(boom=True)
    co_dict, cr_dict = _engine._co_dict, _engine._cr_dict
    _engine._co_dict = lambda lut: {"lut": lut}
    _engine._cr_dict = lambda lut: {"lut": lut, "boom": lut.boom}
    try:
        ctx = FakeCtx()
        _engine.ensure_luts(ctx, old_co, None)
        assert ctx.lut_key == (old_co, None)
        with pytest.raises(RuntimeError):
            _engine.ensure_luts(ctx, new_co, bad_cr)
        assert ctx.lut_key == (new_co, None)  # the co-pol table IS the new one; the cross-pol slot holds nothing
        _engine.ensure_luts(ctx, old_co, None)  # the old LUT object is installed again, not skipped
        assert ctx.lut_key == (old_co, None) and ctx.installed == ["co", "co", "co"]
    finally:
        _engine._co_dict, _engine._cr_dict = co_dict, cr_dict


def test_a_one_entry_devices_list_is_honoured():
    """options.devices = [3] means GPU 3, not options.device."""
    import xsarsea_amd
    from xsarsea_amd.windspeed import _engine
    prev = xsarsea_amd.options.devices
    try:
        xsarsea_amd.options.devices = [3]
        assert _engine._device_list() == [3]
        xsarsea_amd.options.devices = None
        assert _engine._device_list() is None
        xsarsea_amd.options.devices = []
        with pytest.raises(ValueError):
            _engine._device_list()
    finally:
        xsarsea_amd.options.devices = prev
