"""GPU: BASELINE.json configs 3, 4 and 5 at their STATED shapes (device-resident float32 rasters, complex64 out), checked
through size-independent properties -- the production kernel against an independent kernel on every pixel, row-tile
independence -- and against the C oracle on crops.  Config 2 (10000 x 10000 mono) runs at its stated shape in
test_gpu_fullsize.py (parametrised over 10000 and 20000); config 1 is test_gpu_api.py::test_sigma0_detrend."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(torch, t):
    return torch.view_as_real(t).view(torch.int32)


def _crops(rng, lines, samples, side, n):
    for _ in range(n):
        l0, s0 = int(rng.integers(0, lines - side)), int(rng.integers(0, samples - side))
        yield (slice(l0, l0 + side), slice(s0, s0 + side))


def test_config3_dual_pol_20000x20000(default_luts):
    """Dual-pol (CMOD5.N + S1 VH GMF) at 20000 x 20000: the production kernel (branch-and-bound co-pol search + interval-pruned
    cross-pol search, fused dual select) against XSW_ALGO_EXACT (every candidate of both searches in the reference's operation
    order) on all 4e8 pixels, both outputs, bit for bit; then 128 x 128 crops against the C oracle (indices + NaN masks)."""
    torch = pytest.importorskip("torch")
    import bench
    from oracle import invert as oinv
    from util import oracle_full
    from xsarsea_amd import _lib
    from xsarsea_amd.windspeed import _engine, get_model
    N = 20000
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        ctx.upload_luts(co=_engine._co_dict(get_model("gmf_cmod5n")._lut(units="dB")),
                        cr=_engine._cr_dict(get_model("gmf_s1_v2")._lut(units="dB")))
        inc, s_vv, anc = bench.make_scene(N, N, N, 0, 20260320 + 3, dev)
        s_vh, dsig = bench.make_crosspol(inc, anc, 3003, dev)
        outs = {}
        torch.cuda.synchronize()
        for algo in ("pruned", "exact"):
            co = torch.empty((N, N), dtype=torch.complex64, device=dev)
            cr = torch.empty_like(co)
            ctx.invert_raw(N, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), s_vh.data_ptr(),
                           dsig.data_ptr(), anc.data_ptr(), co.data_ptr(), cr.data_ptr(), algo=_lib.ALGOS[algo], dual_select=True)
            ctx.synchronize()
            outs[algo] = (co, cr)
        for k, name in ((0, "co"), (1, "dual")):
            a, b = _bits(torch, outs["pruned"][k]), _bits(torch, outs["exact"][k])
            diff = int((a != b).any(dim=-1).sum().item())
            assert diff == 0, f"{diff} of {N * N} {name} pixels differ between the production and the exact kernel"
        assert torch.equal(torch.isnan(outs["pruned"][0].real), torch.isnan(inc) | torch.isnan(s_vv))
        # crops vs the C oracle: same dB values on both sides (host numpy conversion), raw cross-pol output
        lco, lcr = default_luts
        for sl in _crops(np.random.default_rng(3), N, N, 128, 3):
            ci, cs, ch, cd, ca = (t[sl].contiguous().cpu().numpy() for t in (inc, s_vv, s_vh, dsig, anc))
            o = oracle_full(ci, cs, ch, cd, ca, lco, lcr)
            g = ctx.invert_host(ci, sigma0_co=oinv.to_db(cs), sigma0_cr=oinv.to_db(ch), dsig_cr=cd, anc=ca, sigma0_is_db=True,
                                algo="pruned", want_idx=True)
            assert np.array_equal(g[2], o[2]), "crop: grid indices differ from the C oracle"
            assert np.array_equal(np.isnan(g[1].real), np.isnan(o[1].real))
    finally:
        ctx.close()


def test_config4_25000x17000_eight_row_tiles():
    """Sentinel-1 IW full-swath shape, 25000 lines x 17000 samples (samples not a multiple of 64): the eight `tile_bounds` row
    tiles an 8-GPU job would invert (3125 lines each), inverted one after the other on this GPU, equal the single launch bit
    for bit; and the single launch equals the LDS-tiled exhaustive sweep on every pixel."""
    torch = pytest.importorskip("torch")
    import bench
    from xsarsea_amd import _lib, multi_gpu
    L, S = 25000, 17000
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        _, co = bench.build_product_lut()
        ctx.upload_luts(co=co)
        # the scene as the 8 ranks of `bench.py --gpus 8 --config 4` generate it: one tile per rank, its own seed
        parts = [bench.make_scene(b1 - b0, S, L, b0, 20260320 + 2 + r, dev)
                 for r, (b0, b1) in enumerate(multi_gpu.tile_bounds(L, 8, q) for q in range(8))]
        inc, s_vv, anc = (torch.cat([p[i] for p in parts]) for i in range(3))
        del parts
        whole = torch.empty((L, S), dtype=torch.complex64, device=dev)
        tiled = torch.empty_like(whole)
        torch.cuda.synchronize()

        def run(l0, l1, dst, algo):
            off = l0 * S
            ctx.invert_raw(l1 - l0, S, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr() + off * 4, s_vv.data_ptr() + off * 4,
                           None, None, anc.data_ptr() + off * 8, dst.data_ptr() + off * 8, None, algo=algo)

        run(0, L, whole, _lib.ALGO_PRUNED)
        for r in range(8):
            b0, b1 = multi_gpu.tile_bounds(L, 8, r)
            assert b1 - b0 == 3125
            for k in range(8):  # and each tile in the 8 chunks bench.py pipelines behind the gather
                c0, c1 = multi_gpu.chunk_bounds(b1 - b0, 8, k)
                run(b0 + c0, b0 + c1, tiled, _lib.ALGO_PRUNED)
        ctx.synchronize()
        assert torch.equal(_bits(torch, tiled), _bits(torch, whole)), "row tiling changes results"
        del tiled
        ex = torch.empty_like(whole)
        run(0, L, ex, _lib.ALGOS["exhaustive"])
        ctx.synchronize()
        diff = int((_bits(torch, whole) != _bits(torch, ex)).any(dim=-1).sum().item())
        assert diff == 0, f"{diff} of {L * S} pixels differ between the pruned and the exhaustive kernel"
        nan_in = torch.isnan(inc) | torch.isnan(s_vv)
        assert torch.equal(torch.isnan(whole.real), nan_in) and int(nan_in.sum()) > 0
    finally:
        ctx.close()


def test_config5_cmod7_shaped_lut_20000x20000(tmp_path):
    """CMOD7-format table (250 x 73 x 51 float32, Fortran order) -> product reader -> 501 x 499 x 181 dB LUT, inverted at
    20000 x 20000: production kernel == exhaustive sweep on every pixel, and 160 x 160 crops == the C oracle fed the same LUT."""
    torch = pytest.importorskip("torch")
    import bench
    from oracle import invert as oinv, lut as olut
    from util import oracle_full
    from xsarsea_amd import _lib
    from xsarsea_amd.windspeed import _engine
    N = 20000
    dev = torch.device("cuda", 0)
    model = bench.cmod7_shaped_model(str(tmp_path))
    lut = model._lut(units="dB")
    assert lut.shape == (501, 499, 181)
    ctx = _lib.Context(0)
    try:
        ctx.upload_luts(co=_engine._co_dict(lut))
        inc, s_vv, anc = bench.make_scene(N, N, N, 0, 20260320 + 5, dev)
        pr = torch.empty((N, N), dtype=torch.complex64, device=dev)
        ex = torch.empty_like(pr)
        torch.cuda.synchronize()
        for dst, algo in ((pr, _lib.ALGO_PRUNED), (ex, _lib.ALGOS["exhaustive"])):
            ctx.invert_raw(N, N, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                           anc.data_ptr(), dst.data_ptr(), None, algo=algo)
        ctx.synchronize()
        diff = int((_bits(torch, pr) != _bits(torch, ex)).any(dim=-1).sum().item())
        assert diff == 0, f"{diff} of {N * N} pixels differ between the pruned and the exhaustive kernel (CMOD7-shaped LUT)"
        lco = olut.Lut(lut.values, lut.incidence, lut.wspd, lut.phi, "dB", "high", "gmf_cmod7", "VV")
        for sl in _crops(np.random.default_rng(5), N, N, 160, 3):
            ci, cs, ca = (t[sl].contiguous().cpu().numpy() for t in (inc, s_vv, anc))
            o = oracle_full(ci, cs, None, None, ca, lco, None)
            g = ctx.invert_host(ci, sigma0_co=oinv.to_db(cs), anc=ca, sigma0_is_db=True, algo="pruned", want_idx=True)
            assert np.array_equal(g[2][..., :2], o[2][..., :2]), "crop: grid indices differ from the C oracle"
    finally:
        ctx.close()


def _run_bench(extra_args, tmp_path):
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(XSW_BENCH_BACKEND="gloo", XSW_BENCH_ONE_DEVICE="1")  # N ranks on this one GPU: everything but RCCL itself
    r = subprocess.run([sys.executable, bench] + extra_args, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, f"bench.py failed ({r.returncode}):\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_contract(tmp_path):
    """The default bench command (at a reduced raster so that it runs in seconds) prints ONE JSON line carrying the driver's
    contract: metric/value/unit/n_gpus/..., `roofline` (hbm, dominant kernel, its own HIP-event time, the valu view),
    `cpu_baseline` (oracle C port on a bounded crop) and the parity block; the two-kernel path reports both kernels."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--lines", "768", "--samples", "2048", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "Mpixels/s" and j["vs_baseline"] is None and j["value"] > 0
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    assert rf["kernel"] == "k_invert_band" and rf["kernel_ms"] > 0
    sk = rf["second_kernel"]  # the rest of the chain: k_invert_band2 (pixels with long runs of band rows) + k_invert_list
    assert sk["kernel"] == "k_invert_band2 + k_invert_blocks + k_invert_list" and sk["k_invert_band2_ms"] > 0 and sk["k_invert_blocks_ms"] >= 0 and sk["k_invert_list_ms"] > 0
    assert abs(sk["k_invert_band2_ms"] + sk["k_invert_blocks_ms"] + sk["k_invert_list_ms"] - sk["kernel_ms"]) < 0.01
    assert rf["chain"]["ms"] >= rf["kernel_ms"] and 0 < rf["chain"]["frac"] < rf["frac"] * 1.01
    assert rf["kernel_ms"] + rf["second_kernel"]["kernel_ms"] <= rf["step_kernels_ms"] * 1.05 + 0.05
    v = rf["valu"]
    assert v["candidates_per_pixel_full_grid"] == 499 * 181 and 16 < v["evaluated_candidates_per_pixel"] < 1000 and 0 < v["frac"] < 1
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "crop" in cb["sample"]
    p = j["parity"]
    assert p["nan_mask_equal"] is True and p["index_match_host_db"] == 1.0 and p["index_match_device_db"] > 0.999
    assert j["detrend"]["roofline"]["bound"] == "hbm" and j["nesz_flatten"]["roofline"]["bound"] == "hbm" and j["lut"]["lut_device_build_ms"] > 0
    # round 3: the line states the error of the path it times, times the bit-parity route end to end, and carries the host path
    for k in ("max_rel_err_uv_c64_device_db", "pixels_outside_1e-4_device_db", "pixels_outside_1e-4_host_db"):
        assert k in p, k
    assert p["pixels_outside_1e-4_host_db"] == 0 and p["frac_outside_1e-4_device_db"] < 1e-3
    e2e = j["parity_config_end_to_end"]
    assert e2e["value"] > 0 and e2e["ms"] > 0 and e2e["bytes_over_pcie"] == 4 * 768 * 2048  # sigma0 alone crosses the link (XSW_MEM_DEVICE_SIGMA0_HOST)
    hp = j["host_path"]
    assert hp["value"] > 0 and hp["roofline"]["bound"] == "pcie" and hp["bytes_over_pcie"] == 20 * 768 * 2048
    assert "numpy_restatement" in cb and cb["numpy_restatement"]["equals_c_port"] is True
    # counter-derived fields are either fresh (stamped with the loaded library's device-code hash) or null with a reason
    assert rf["traffic"] is None or "measured_on" in rf["traffic_provenance"]


@pytest.mark.parametrize("cfg,shape", [("4", (2503, 1700)), ("3", (1001, 1030))])
def test_two_rank_rehearsal_gathers_the_single_launch_raster(cfg, shape, tmp_path):
    """`python bench.py --gpus 2` starts its own two ranks (here both on this GPU, gloo instead of RCCL): strong scaling,
    uneven tiles, chunked gather inside the step; rank 0 then inverts the whole raster in ONE launch and the gathered raster
    must equal it bit for bit (mono config-4 aspect ratio, and dual-pol with both outputs)."""
    j = _run_bench(["--gpus", "2", "--config", cfg, "--lines", str(shape[0]), "--samples", str(shape[1]), "--steps", "2",
                    "--warmup", "1", "--verify-gather", "--no-extras", "--no-cpu-baseline"], tmp_path)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["gather_verified"] is True
    assert j["config"]["lines"] == shape[0] and j["config"]["lines_rank0"] == shape[0] // 2
    # the gather ships 4-byte grid codes (two for dual-pol) and explains itself
    mg = j["multi_gpu"]
    assert mg["gather_bytes"] == (shape[0] - shape[0] // 2) * shape[1] * (4 if cfg == "4" else 8)
    assert mg["gather_only_ms"] > 0 and len(mg["kernel_ms_per_rank"]) == 2 and mg["no_gather"]["value"] > 0
    assert "exposed_gather_ms" in mg
    assert j["roofline"]["valu"]["evaluated_candidates_per_pixel"] > 10


def test_tiled_api_real_inversions():
    """`multi_gpu.invert_from_model_tiled` with the real drop-in call per tile: 3 ranks on this GPU (gloo), uneven tiles, a tile
    whose ancillary wind is all NaN, a 1-D incidence row on a square raster; rank 0's raster == the single-process call, bit
    for bit (mono + dual).  Fresh processes: the parent makes no GPU call."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rehearse_tiled_two_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, script, "3"], env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
