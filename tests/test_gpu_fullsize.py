"""GPU: BASELINE.json's mono-VV rasters at their STATED shapes -- config 2 (10000 x 10000) and the metric's raster
(20000 x 20000) -- float32 in / complex64 out, device-resident, checked through size-independent properties: determinism, tile independence (row tiling == whole raster),
grid membership of every solution, and agreement of the branch-and-bound kernel with the LDS-tiled
exhaustive sweep and with the oracle on crops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = {"config2_10000": 10000, "metric_20000": 20000}


@pytest.fixture(scope="module", params=sorted(SIZES))
def full_scene(request):
    torch = pytest.importorskip("torch")
    N = SIZES[request.param]
    import bench
    from xsarsea_amd import _lib
    dev = torch.device("cuda", 0)
    lut, co = bench.build_product_lut()
    ctx = _lib.Context(0)
    ctx.upload_luts(co=co)
    inc, s_vv, anc = bench.make_scene(N, N, N, 0, 424242 if N == 20000 else 20260320 + 2, dev)
    out = torch.empty((N, N), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()  # the context launches on its own stream: the scene must be complete first
    ctx.invert_raw(N, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                   anc.data_ptr(), out.data_ptr(), None, algo=_lib.ALGO_PRUNED)
    ctx.synchronize()
    yield dict(torch=torch, ctx=ctx, lut=lut, inc=inc, s_vv=s_vv, anc=anc, out=out, N=N)
    ctx.close()


def test_deterministic_and_tile_independent(full_scene):
    f = full_scene
    torch, ctx, N = f["torch"], f["ctx"], f["N"]
    from xsarsea_amd import _lib
    again = torch.empty_like(f["out"])
    ctx.invert_raw(N, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, f["inc"].data_ptr(), f["s_vv"].data_ptr(), None, None,
                   f["anc"].data_ptr(), again.data_ptr(), None, algo=_lib.ALGO_PRUNED)
    ctx.synchronize()
    a, b = torch.view_as_real(f["out"]), torch.view_as_real(again)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "two runs differ"
    # row tiles of unequal heights (as ranks of a multi-GPU job would take them) == the whole raster
    tiled = torch.empty_like(f["out"])
    for l0, l1 in ((0, N // 4 - 1), (N // 4 - 1, N // 2 + 2), (N // 2 + 2, N - 2999), (N - 2999, N)):
        off = l0 * N
        ctx.invert_raw(l1 - l0, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, f["inc"].data_ptr() + off * 4,
                       f["s_vv"].data_ptr() + off * 4, None, None, f["anc"].data_ptr() + off * 8, tiled.data_ptr() + off * 8,
                       None, algo=_lib.ALGO_PRUNED)
    ctx.synchronize()
    assert torch.equal(torch.view_as_real(tiled).view(torch.int32), a.view(torch.int32)), "row tiling changes results"
    # checksum of checksums (documents the run; any change of a single pixel changes it)
    cs = torch.view_as_real(f["out"]).view(torch.int32).to(torch.int64).sum(dim=(1, 2)).cpu().numpy()
    assert cs.shape == (N,) and np.unique(cs).size > N // 2


def test_solutions_lie_on_the_lut_grid(full_scene):
    """|wind| must be one of the 499 grid speeds and its direction one of the 181 grid directions (+-), NaN
    exactly where incidence or sigma0 is NaN."""
    f = full_scene
    torch, N = f["torch"], f["N"]
    out = f["out"]
    nan_in = torch.isnan(f["inc"]) | torch.isnan(f["s_vv"])
    assert torch.equal(torch.isnan(out.real), nan_in)
    ok = ~nan_in
    spd = out.abs()[ok].double()
    k = torch.round((spd - 0.2) / 0.1)
    assert float((spd - (0.2 + 0.1 * k)).abs().max()) < 2e-5 and float(k.min()) >= 0 and float(k.max()) <= 498
    ang = torch.rad2deg(torch.atan2(out.imag[ok], out.real[ok])).double().abs()
    assert float((ang - torch.round(ang)).abs().max()) < 2e-3
    # sign of the direction follows the ancillary wind's azimuth component (windspeed.py:234-242)
    im_o, im_a = out.imag[ok], f["anc"].imag[ok]
    both = (im_o.abs() > 1e-3) & (im_a.abs() > 1e-3)
    assert bool(((im_o[both] > 0) == (im_a[both] > 0)).all())


def test_pruned_equals_exhaustive_and_oracle_on_crops(full_scene, default_luts):
    f = full_scene
    ctx, N = f["ctx"], f["N"]
    from oracle import invert as oinv
    from util import oracle_full
    lco, _ = default_luts
    rng = np.random.default_rng(0)
    for _ in range(3):
        l0, s0 = int(rng.integers(0, N - 256)), int(rng.integers(0, N - 256))
        sl = (slice(l0, l0 + 256), slice(s0, s0 + 256))
        ci, cs, ca = (f[k][sl].contiguous().cpu().numpy() for k in ("inc", "s_vv", "anc"))
        ref_full = f["out"][sl].cpu().numpy()
        pr = ctx.invert_host(ci, sigma0_co=cs, anc=ca, algo="pruned", want_idx=True, out_dtype=np.complex64)
        ex = ctx.invert_host(ci, sigma0_co=cs, anc=ca, algo="exhaustive", want_idx=True, out_dtype=np.complex64)
        assert np.array_equal(pr[2], ex[2]), "pruned and exhaustive kernels disagree"
        assert np.array_equal(pr[0].view(np.int32), ref_full.view(np.int32)), "crop != same pixels of the full run"
        # strict oracle parity with identical dB values
        o = oracle_full(ci, cs, None, None, ca, lco, None)
        st = ctx.invert_host(ci, sigma0_co=oinv.to_db(cs), anc=ca, sigma0_is_db=True, algo="pruned", want_idx=True)
        assert np.array_equal(st[2][..., :2], o[2][..., :2])


def test_pruned_equals_exhaustive_on_the_whole_raster(full_scene):
    """Every pixel of the raster (1e8 for config 2, 4e8 for the metric's): the branch-and-bound kernel (552 candidates scored per pixel) and the LDS-tiled exhaustive sweep
    (all 90 319, an independent code path: no window, no forward differences along a window, float32 screening + float64
    settle) return the same bits."""
    f = full_scene
    torch, ctx, N = f["torch"], f["ctx"], f["N"]
    from xsarsea_amd import _lib
    ex = torch.empty_like(f["out"])
    ctx.invert_raw(N, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, f["inc"].data_ptr(), f["s_vv"].data_ptr(), None, None,
                   f["anc"].data_ptr(), ex.data_ptr(), None, algo=_lib.ALGOS["exhaustive"])
    ctx.synchronize()
    a, b = torch.view_as_real(f["out"]).view(torch.int32), torch.view_as_real(ex).view(torch.int32)
    diff = int((a != b).any(dim=-1).sum().item())
    assert diff == 0, f"{diff} of {N * N} pixels differ between the pruned and the exhaustive kernel"
