"""world_size-2 `gloo` test of the row-tiling layer (the N>1 path of bench.py, minus the kernel)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xsarsea_amd import multi_gpu


def test_tile_bounds_cover_exactly():
    for lines in (0, 1, 7, 8, 25000, 20001):
        for world in (1, 2, 3, 8):
            spans = [multi_gpu.tile_bounds(lines, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == lines
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
    with pytest.raises(ValueError):
        multi_gpu.tile_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lines, samples, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(lines * samples, dtype=torch.float32).reshape(lines, samples)
        full = torch.complex(full, -full)
        l0, l1 = multi_gpu.tile_bounds(lines, world, rank)
        tile = full[l0:l1] * 2  # stand-in for "invert my tile"
        out = multi_gpu.gather_rows(tile, lines, dst=0)
        if rank == 0:
            ret["ok"] = bool(torch.equal(out, full * 2))
        else:
            assert out is None
        # pipelined variant used by bench.py (equal tiles): two row chunks, waits deferred
        if lines % world == 0:
            per = lines // world
            glob = torch.zeros_like(full) if rank == 0 else None
            reqs = []
            for r0, r1 in ((0, per // 2), (per // 2, per)):
                reqs += multi_gpu.gather_rows_async(tile, lines, r0, r1, dst=0, out=glob)
            for q in reqs:
                q.wait()
            if rank == 0:
                ret["ok"] = ret["ok"] and bool(torch.equal(glob, full * 2))
    finally:
        dist.destroy_process_group()


def _chunk_worker(rank, world, port, lines, samples, n_chunks, ret):
    """bench.py's N > 1 step minus the kernel: uneven row tiles, every tile cut into n_chunks, chunk k gathered while
    chunk k+1 "computes"; waits deferred to the end of the step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(lines * samples, dtype=torch.float32).reshape(lines, samples)
        full = torch.complex(full, -full)
        l0, l1 = multi_gpu.tile_bounds(lines, world, rank)
        tile = torch.zeros((l1 - l0, samples), dtype=full.dtype)
        glob = torch.full_like(full, float("nan")) if rank == 0 else None
        reqs = []
        for k in range(n_chunks):
            c0, c1 = multi_gpu.chunk_bounds(l1 - l0, n_chunks, k)
            tile[c0:c1] = full[l0 + c0:l0 + c1] * 3  # stand-in for "invert chunk k of my tile"
            reqs += multi_gpu.gather_chunk_async(tile, lines, k, n_chunks, dst=0, out=glob)
        for q in reqs:
            q.wait()
        if rank == 0:
            ret["ok"] = bool(torch.equal(glob, full * 3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lines,n_chunks", [(2, 25, 8), (3, 10, 8), (2, 3, 8), (3, 64, 4)])
def test_gather_chunks_uneven_tiles_gloo(world, lines, n_chunks):
    """tile_bounds gives the last rank the remainder and a tile may hold fewer lines than chunks (empty chunks)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_chunk_worker, args=(world, _free_port(), lines, 7, n_chunks, ret), nprocs=world, join=True)
    assert ret.get("ok") is True


def test_chunk_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 9, 3125):
        for c in (1, 3, 8):
            spans = [multi_gpu.chunk_bounds(n, c, k) for k in range(c)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a1 == b0 and a0 <= a1 for (a0, a1), (b0, b1) in zip(spans, spans[1:]))


def test_bench_refuses_a_rank_count_it_cannot_start():
    """`bench.py --gpus N` must run N ranks or fail: never print an n_gpus it did not use (checked before any GPU call)."""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "XSW_BENCH_ONE_DEVICE")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


@pytest.mark.parametrize("lines", [9, 64])
def test_gather_rows_gloo_world2(lines):
    world, samples = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), lines, samples, ret), nprocs=world, join=True)
    assert ret.get("ok") is True


def _tiled_worker(rank, world, port, lines, samples, ret):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)  # the same full rasters on every rank
        inc = rng.uniform(20, 45, (lines, samples)).astype(np.float32)
        s_co = rng.uniform(0.01, 0.2, (lines, samples)).astype(np.float32)
        s_cr = rng.uniform(0.001, 0.01, (lines, samples)).astype(np.float32)
        anc = (rng.normal(0, 8, (lines, samples)) + 1j * rng.normal(0, 8, (lines, samples))).astype(np.complex64)
        dsig = rng.uniform(0.05, 0.5, (lines, samples)).astype(np.float32)
        seen = {}

        def fake_invert(i, a, b=None, /, ancillary_wind=None, dsig_cr=0.1, model=None):  # stands in for the GPU call
            seen["shape"] = a.shape
            assert i.shape == a.shape == ancillary_wind.shape and (b is None or b.shape == a.shape)
            co = (i + 2 * a).astype(np.float64) * ancillary_wind.astype(np.complex128)
            if b is None:
                return co
            return co, (b * np.asarray(dsig_cr)).astype(np.complex128)

        mono = multi_gpu.invert_from_model_tiled(inc, s_co, ancillary_wind=anc, model="m", invert=fake_invert)
        l0, l1 = multi_gpu.tile_bounds(lines, world, rank)
        ok = seen["shape"] == (l1 - l0, samples)
        dual = multi_gpu.invert_from_model_tiled(inc, s_co, s_cr, ancillary_wind=anc, dsig_cr=dsig, model=("a", "b"), invert=fake_invert)
        if rank == 0:
            ok = ok and np.array_equal(mono, fake_invert(inc, s_co, ancillary_wind=anc))
            e0, e1 = fake_invert(inc, s_co, s_cr, ancillary_wind=anc, dsig_cr=dsig)
            ok = ok and isinstance(dual, tuple) and np.array_equal(dual[0], e0) and np.array_equal(dual[1], e1)
            ret["ok"] = bool(ok)
        else:
            assert mono is None and dual is None
            ret[f"ok{rank}"] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lines", [(2, 11), (3, 8)])
def test_invert_from_model_tiled_gloo(world, lines):
    """User-level entry: every rank passes the full rasters, inverts its own lines (stand-in callable: no GPU here), rank 0
    gets the full mono / dual results, the others None; uneven tiles, raster-valued dsig_cr sliced with the rest."""
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_tiled_worker, args=(world, _free_port(), lines, 37, ret), nprocs=world, join=True)
        assert ret.get("ok") is True and all(ret.get(f"ok{r}") for r in range(1, world))


def test_invert_from_model_tiled_without_a_process_group():
    import numpy as np
    a = np.ones((4, 5), np.float32)
    out = multi_gpu.invert_from_model_tiled(a, a, ancillary_wind=a.astype(np.complex64), invert=lambda i, s, ancillary_wind=None: s * 3)
    assert np.array_equal(out, a * 3)


def _tiled_edge_worker(rank, world, port, ret):
    """a 1-D incidence row of a SQUARE raster is not a raster (not cut); a rank whose inversion fails does not
    leave the others hanging in the gather: every rank raises."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 6  # square: lines == samples
        rng = np.random.default_rng(9)
        inc_row = rng.uniform(20, 45, n).astype(np.float32)
        s_co = rng.uniform(0.01, 0.2, (n, n)).astype(np.float32)
        anc = (rng.normal(0, 8, (n, n)) + 1j * rng.normal(0, 8, (n, n))).astype(np.complex64)
        seen = {}

        def fake(i, a, /, ancillary_wind=None, model=None):
            seen["inc_shape"] = np.shape(i)
            return (np.asarray(i)[None, :] + 2 * a).astype(np.float64) * ancillary_wind.astype(np.complex128)

        out = multi_gpu.invert_from_model_tiled(inc_row, s_co, ancillary_wind=anc, model="m", invert=fake)
        ok = seen["inc_shape"] == (n,)
        if rank == 0:
            ok = ok and np.array_equal(out, fake(inc_row, s_co, ancillary_wind=anc))

        def failing(i, a, /, ancillary_wind=None, model=None):
            if dist.get_rank() == 1:
                raise ValueError("boom on rank 1")
            return a.astype(np.complex128)

        try:
            multi_gpu.invert_from_model_tiled(inc_row, s_co, ancillary_wind=anc, model="m", invert=failing)
            ok = False  # must not return
        except ValueError as e:
            ok = ok and rank == 1 and "boom" in str(e)
        except RuntimeError as e:
            ok = ok and rank != 1 and "another rank" in str(e)
        # gather=False: every rank keeps its own lines (what a dask consumer of row blocks does): (l0, l1, its tile's result)
        l0, l1, mine = multi_gpu.invert_from_model_tiled(inc_row, s_co, ancillary_wind=anc, model="m", invert=fake, gather=False)
        ok = ok and (l0, l1) == multi_gpu.tile_bounds(n, world, rank) and np.array_equal(mine, fake(inc_row, s_co[l0:l1], ancillary_wind=anc[l0:l1]))
        # a device-resident (torch) ancillary wind is looked at where it lives; an all-NaN one is "no valid ancillary wind" on every rank
        seen_flag = {}

        def flagged(i, a, /, ancillary_wind=None, model=None):
            return a.astype(np.complex128)

        out2 = multi_gpu.invert_from_model_tiled(inc_row, s_co, ancillary_wind=torch.from_numpy(anc), model="m", invert=flagged)
        ok = ok and ((rank == 0 and out2.shape == (n, n)) or (rank != 0 and out2 is None))

        # a rank that fails BEFORE its inversion (cutting its tile) does not leave the others in a collective either
        class Bad:
            shape, ndim = (n, n), 2

            def __getitem__(self, k):
                if dist.get_rank() == 1:
                    raise IndexError("cannot cut on rank 1")
                return s_co[k]

        try:
            multi_gpu.invert_from_model_tiled(inc_row, s_co, ancillary_wind=Bad(), model="m", invert=fake)
            ok = False
        except IndexError as e:
            ok = ok and rank == 1 and "cannot cut" in str(e)
        except RuntimeError as e:
            ok = ok and rank != 1 and "another rank failed while cutting" in str(e)
        ret[f"ok{rank}"] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_invert_from_model_tiled_edges_gloo():
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_tiled_edge_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert ret.get("ok0") is True and ret.get("ok1") is True


def _coded_gather_worker(rank, world, port, lines, samples, n_chunks, ret):
    """bench.py's N > 1 step since round 3: int32 grid codes travel; rank 0 produces its own rows directly in the gathered
    raster (self_copy=False) and nothing is copied for it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = (torch.arange(lines * samples, dtype=torch.int64).reshape(lines, samples) % 2147483647).to(torch.int32)
        l0, l1 = multi_gpu.tile_bounds(lines, world, rank)
        glob = torch.full_like(full, -7) if rank == 0 else None
        tile = glob[l0:l1] if rank == 0 else torch.zeros((l1 - l0, samples), dtype=torch.int32)
        reqs = []
        for k in range(n_chunks):
            c0, c1 = multi_gpu.chunk_bounds(l1 - l0, n_chunks, k)
            tile[c0:c1] = full[l0 + c0:l0 + c1]  # "invert chunk k": rank 0 writes straight into the gathered raster
            reqs += multi_gpu.gather_chunk_async(tile, lines, k, n_chunks, dst=0, out=glob, self_copy=False)
        for q in reqs:
            q.wait()
        if rank == 0:
            ret["ok"] = bool(torch.equal(glob, full))
            ret["bytes"] = multi_gpu.gather_bytes_into(lines, samples, world, 0, 4)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lines", [(2, 25), (3, 10)])
def test_coded_gather_gloo(world, lines):
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_coded_gather_worker, args=(world, _free_port(), lines, 7, 8, ret), nprocs=world, join=True)
        assert ret.get("ok") is True
        assert ret["bytes"] == (lines - lines // world) * 7 * 4


def _pipeline_worker(rank, world, port, lines, samples, n_chunks, dual, ret):
    """`multi_gpu.TiledPipeline` -- the object bench.py --gpus N and `invert_from_model_tiled(gather=True)` run -- on CPU tensors
    with stand-in device operations: chunked "inversion" to int32 codes, chunk-by-chunk gather, expansion on the destination."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        code_of = lambda g0, g1: (torch.arange(g0 * samples, g1 * samples, dtype=torch.int64) % 1000003).to(torch.int32).reshape(g1 - g0, samples)
        pipe = multi_gpu.TiledPipeline(lines, samples, dual=dual, device="cpu", dst=0, n_chunks=n_chunks, out_dtype=torch.complex64)
        assert (pipe.g0, pipe.g1) == multi_gpu.tile_bounds(lines, world, rank) and pipe.side is None
        seen = []

        def invert_chunk(k, r0, r1):
            seen.append((k, r0, r1))
            pipe.codes[r0:r1] = code_of(pipe.g0 + r0, pipe.g0 + r1)
            if dual:
                pipe.codes_dual[r0:r1] = -code_of(pipe.g0 + r0, pipe.g0 + r1)

        expanded = []

        def expand_rows(g0, g1, stream):
            assert stream is None and rank == 0
            expanded.append((g0, g1))
            pipe.full[g0:g1] = torch.complex(pipe.full_codes[g0:g1].float(), pipe.full_codes[g0:g1].float() * 2)
            if dual:
                pipe.full_dual[g0:g1] = torch.complex(pipe.full_codes_dual[g0:g1].float(), torch.zeros(g1 - g0, samples))

        ok = True
        for _ in range(2):  # reusable: the second run overwrites the same buffers
            del seen[:], expanded[:]
            pipe.run(invert_chunk, expand_rows)
            out = pipe.finish()
            ok = ok and [c[1:] for c in seen] == [c for c in (multi_gpu.chunk_bounds(pipe.rows, n_chunks, k) for k in range(n_chunks)) if c[1] > c[0]]
        if rank == 0:
            full, full_dual = out
            want = code_of(0, lines).float()
            ok = ok and torch.equal(full, torch.complex(want, want * 2)) and len(set(expanded)) == len(expanded)
            ok = ok and sum(g1 - g0 for g0, g1 in expanded) == lines
            ok = ok and ((full_dual is None) if not dual else torch.equal(full_dual, torch.complex(-want, torch.zeros_like(want))))
            ret["ok"] = bool(ok)
        else:
            ret[f"ok{rank}"] = bool(ok and out is None and pipe.full is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lines,n_chunks,dual", [(2, 25, 8, False), (3, 10, 4, True), (3, 2, 8, False)])
def test_tiled_pipeline_gloo(world, lines, n_chunks, dual):
    """Uneven tiles, more chunks than rows, a rank with an empty tile (2 lines over 3 ranks), mono and dual code rasters."""
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_pipeline_worker, args=(world, _free_port(), lines, 7, n_chunks, dual, ret), nprocs=world, join=True)
        assert ret.get("ok") is True and all(ret.get(f"ok{r}") for r in range(1, world))


def test_code_sink_geometry():
    """`_CodeSink.begin`: the gathered raster's geometry from the leading-axis tiling -- middle axes folded into the rows, 1-D
    rasters as rows of one pixel, the same chunk count on every rank."""
    import numpy as np
    sink = multi_gpu._CodeSink(10, 10, 0, None, 8)
    with pytest.raises(ValueError):
        sink.begin((9, 5), True, False, "cpu", torch.complex64)  # the leading axis is not the tiled one
    assert multi_gpu.chunk_bounds(10, 4, 3) == (7, 10)
    assert np.prod([3, 4]) == 12
