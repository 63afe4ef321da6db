"""Row tiling of a raster across the GPUs of one node: one process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Pixels are independent given the replicated LUT, so the path shards with NO data-path collective:
every rank inverts a contiguous block of lines (the reference's dask strategy: row blocks with the
sample axis unchunked, windspeed/windspeed.py:356-364).  The only exchange is the final gather of the
output tiles on one rank: grouped point-to-point receives straight into the destination raster's row
slices (no padding, no staging copy), each sender on its own xGMI link.
"""
import torch
import torch.distributed as dist


def tile_bounds(lines, world, rank):
    """[l0, l1) of rank's tile: `lines // world` lines each, the last rank takes the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base = lines // world
    l0 = rank * base
    l1 = lines if rank == world - 1 else l0 + base
    return l0, l1


def chunk_bounds(n_rows, n_chunks, k):
    """[c0, c1) of chunk k when n_rows tile-local rows are cut into n_chunks nearly equal chunks."""
    return n_rows * k // n_chunks, n_rows * (k + 1) // n_chunks


def gather_chunk_async(tile, lines, k, n_chunks, dst=0, group=None, out=None, self_copy=True):
    """Start gathering chunk k (of n_chunks, `chunk_bounds` of each rank's OWN tile height) of every rank's tile into
    `out` on `dst`; tiles may be uneven (`tile_bounds`: the last rank takes the remainder).  Returns the requests to
    `wait()` on.  Lets a caller pipeline: invert chunk k, start its gather, invert chunk k+1 while chunk k travels
    over xGMI (RCCL orders each transfer after the work already queued on the current stream).
    Any dtype: complex winds, or the 4-byte grid codes of xsw_invert (`out_code_*`: a quarter / half of the bytes, expanded
    on `dst` by xsw_expand_codes).  self_copy=False: `dst` produced its own rows directly in `out` (nothing to copy)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ops = []
    if rank == dst:
        for r in range(world):
            t0, t1 = tile_bounds(lines, world, r)
            c0, c1 = chunk_bounds(t1 - t0, n_chunks, k)
            if c1 <= c0:
                continue
            if r == dst:
                if self_copy:
                    out[t0 + c0:t0 + c1].copy_(tile[c0:c1], non_blocking=True)
            else:
                ops.append(dist.P2POp(dist.irecv, out[t0 + c0:t0 + c1], r, group))
    else:
        t0, t1 = tile_bounds(lines, world, rank)
        c0, c1 = chunk_bounds(t1 - t0, n_chunks, k)
        if c1 > c0:
            ops.append(dist.P2POp(dist.isend, tile[c0:c1], dst, group))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_bytes_into(lines, samples, world, dst, bytes_per_pixel):
    """Bytes rank `dst` receives from its peers in one gather of a (lines, samples) raster of `bytes_per_pixel`."""
    t0, t1 = tile_bounds(lines, world, dst)
    return (lines - (t1 - t0)) * samples * bytes_per_pixel


def gather_rows_async(tile, lines, row0, row1, dst=0, group=None, out=None):
    """Start gathering rows [row0, row1) of every rank's tile (rows are tile-local, the same for all
    ranks: equal tiles) into `out` on `dst`; returns the list of requests to `wait()` on.  Lets a caller
    pipeline: invert chunk k, start its gather, invert chunk k+1 while chunk k travels over xGMI."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ops = []
    if rank == dst:
        for r in range(world):
            t0, _ = tile_bounds(lines, world, r)
            if r == dst:
                out[t0 + row0:t0 + row1].copy_(tile[row0:row1], non_blocking=True)
            else:
                ops.append(dist.P2POp(dist.irecv, out[t0 + row0:t0 + row1], r, group))
    else:
        ops.append(dist.P2POp(dist.isend, tile[row0:row1], dst, group))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_rows(tile, lines, dst=0, group=None, out=None):
    """Gather row tiles (shape (l1-l0, samples, ...)) into the full raster on rank `dst`.

    Returns the full tensor on `dst`, None elsewhere.  One batch of isend/irecv: rank `dst` posts one
    receive per peer directly into `out[l0:l1]`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        if out is None:
            return tile
        out.copy_(tile)
        return out
    ops = []
    if rank == dst:
        if out is None:
            out = torch.empty((lines,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        l0, l1 = tile_bounds(lines, world, rank)
        out[l0:l1].copy_(tile)
        for r in range(world):
            if r != dst:
                r0, r1 = tile_bounds(lines, world, r)
                if r1 > r0:
                    ops.append(dist.P2POp(dist.irecv, out[r0:r1], r, group))
    else:
        if tile.shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, tile.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out if rank == dst else None


class TiledPipeline:
    """One rank's side of a tiled inversion whose result is gathered on `dst` -- the measured N > 1 step of bench.py, as a
    library object.

    The rank's tile is inverted in `n_chunks` row chunks to 4-byte GRID CODES (xsw_invert's out_code_*: the retrieved wind is a
    grid point; a quarter of the complex64 bytes, an eighth of complex128).  Chunk k travels to `dst` (point-to-point receives
    straight into the destination rows, one xGMI link per sender under RCCL) while chunk k + 1 is being inverted, and `dst`
    expands every chunk to complex winds on a SIDE stream as soon as its codes have landed, so that only the last chunk's
    expansion is exposed.  `dst` writes its own codes straight into the gathered raster (nothing is copied for it).

    The pipeline owns the buffers, the exchange and the stream choreography; the two device operations come in as callables:
      invert_chunk(k, r0, r1)          invert tile-local rows [r0, r1) INTO `self.codes[r0:r1]` (and `self.codes_dual[r0:r1]`),
                                       launched on the current stream;
      expand_rows(g0, g1, stream)      on `dst`: codes of gathered rows [g0, g1) -> winds in `self.full` (and `self.full_dual`),
                                       launched on `stream` (a torch stream; None on a CPU device: do it now).
    `bounds(rank) -> (g0, g1)`: the rows of the gathered raster a rank owns (default `tile_bounds(total_rows, world, rank)`).
    Reusable: `run()` + `finish()` any number of times on the same buffers (rasters of one shape)."""

    def __init__(self, total_rows, samples, *, dual=False, device=None, dst=0, group=None, n_chunks=8, out_dtype=None, bounds=None,
                 want_co=True):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if _group_up() else 1
        self.rank = dist.get_rank(group) if _group_up() else 0
        self.total_rows, self.samples, self.dual, self.want_co = int(total_rows), int(samples), bool(dual), bool(want_co)
        self.bounds = bounds or (lambda r: tile_bounds(self.total_rows, self.world, r))
        self.g0, self.g1 = self.bounds(self.rank)
        self.rows = self.g1 - self.g0
        self.n_chunks = max(1, int(n_chunks))
        self.device = torch.device("cpu") if device is None else torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        backend = dist.get_backend(group) if _group_up() else None
        # RCCL orders a transfer after the work queued on the current stream and lets a stream wait for it; gloo reads and writes
        # the tensors' memory from the host with no regard for streams: synchronise around it
        self.stream_aware = backend == "nccl"
        out_dtype = out_dtype or torch.complex64
        is_dst = self.rank == dst
        mk = lambda rows, dt: torch.empty((rows, self.samples), dtype=dt, device=self.device)
        two = self.dual or not self.want_co  # a second code raster: the cross-pol search (dual-pol, or cross-pol only)
        self.full_codes = mk(self.total_rows, torch.int32) if is_dst and self.want_co else None
        self.full_codes_dual = mk(self.total_rows, torch.int32) if is_dst and two else None
        self.full = mk(self.total_rows, out_dtype) if is_dst and self.want_co else None
        self.full_dual = mk(self.total_rows, out_dtype) if is_dst and two else None
        if is_dst:
            self.codes = None if self.full_codes is None else self.full_codes[self.g0:self.g1]
            self.codes_dual = None if self.full_codes_dual is None else self.full_codes_dual[self.g0:self.g1]
        else:
            self.codes = mk(self.rows, torch.int32) if self.want_co else None
            self.codes_dual = mk(self.rows, torch.int32) if two else None
        self.side = torch.cuda.Stream(device=self.device) if (self.on_gpu and is_dst) else None
        self._pending, self._chunk_reqs, self._launch = [], {}, None

    # -- one chunk ------------------------------------------------------------------------------------------------------
    def _start_gather(self, k):
        if self.world == 1:
            return
        if self.on_gpu and not self.stream_aware:
            torch.cuda.synchronize(self.device)
        reqs = []
        for tile, out in ((self.codes, self.full_codes), (self.codes_dual, self.full_codes_dual)):
            if tile is not None:
                reqs += _gather_chunk_bounds(tile, self.bounds, k, self.n_chunks, self.dst, self.group, out)
        self._chunk_reqs[k] = reqs
        self._pending.extend(reqs)

    def _expand_chunk(self, k, expand_rows):
        """dst: chunk k of every rank's tile -> winds, on the side stream, behind the chunk's receives and dst's own kernels"""
        own_done = None
        if self.on_gpu:
            own_done = torch.cuda.Event()
            own_done.record(self._launch)

        def body():
            if own_done is not None:
                self.side.wait_event(own_done)
            for q in self._chunk_reqs.pop(k, []):
                q.wait()  # RCCL: the CURRENT (side) stream waits for the transfer; gloo: the host does
            if self.on_gpu and not self.stream_aware:
                torch.cuda.synchronize(self.device)
            for r in range(self.world):
                t0, t1 = self.bounds(r)
                c0, c1 = chunk_bounds(t1 - t0, self.n_chunks, k)
                if c1 > c0:
                    expand_rows(t0 + c0, t0 + c1, self.side)

        if self.side is not None:
            with torch.cuda.stream(self.side):
                body()
        else:
            body()

    def chunk_done(self, k, expand_rows):
        """Chunk k of this rank's tile has been queued (its codes are being written): start its gather; on `dst`, queue its expansion."""
        if self._launch is None and self.on_gpu:
            self._launch = torch.cuda.current_stream(self.device)
        self._start_gather(k)
        if self.rank == self.dst:
            self._expand_chunk(k, expand_rows)

    def run(self, invert_chunk, expand_rows):
        """Queue the whole step: every chunk's inversion, gather and (on `dst`) expansion.  Returns at once on a GPU."""
        self._launch = torch.cuda.current_stream(self.device) if self.on_gpu else None
        for k in range(self.n_chunks):
            r0, r1 = chunk_bounds(self.rows, self.n_chunks, k)
            if r1 > r0:
                invert_chunk(k, r0, r1)
            self.chunk_done(k, expand_rows)

    def gather_only(self):
        """Measurement aid: the exchange alone -- every chunk's codes as they are, no inversion, no expansion -- and the wait for
        this rank's transfers (stream-ordered under RCCL)."""
        for k in range(self.n_chunks):
            self._start_gather(k)
        self._chunk_reqs.clear()
        while self._pending:
            self._pending.pop().wait()

    def finish(self):
        """Completes the exchange: senders wait for their sends, `dst`'s launch stream continues once the last chunk is expanded.
        Returns (full, full_dual) on `dst` (None where that search did not run), None on the other ranks.  Stream-ordered on a
        GPU under RCCL (no host synchronisation)."""
        while self._pending:
            q = self._pending.pop()
            if self.rank != self.dst:
                q.wait()
        self._chunk_reqs.clear()
        if self.on_gpu and not self.stream_aware:
            torch.cuda.synchronize(self.device)
        if self.side is not None and self._launch is not None:
            done = torch.cuda.Event()
            done.record(self.side)
            self._launch.wait_event(done)
        return (self.full, self.full_dual) if self.rank == self.dst else None


def _group_up():
    return dist.is_available() and dist.is_initialized()


def _gather_chunk_bounds(tile, bounds, k, n_chunks, dst, group, out):
    """`gather_chunk_async` for any row ownership `bounds(rank) -> (g0, g1)`; `dst` produced its own rows in place."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ops = []
    if rank == dst:
        for r in range(world):
            if r == dst:
                continue
            t0, t1 = bounds(r)
            c0, c1 = chunk_bounds(t1 - t0, n_chunks, k)
            if c1 > c0:
                ops.append(dist.P2POp(dist.irecv, out[t0 + c0:t0 + c1], r, group))
    else:
        t0, t1 = bounds(rank)
        c0, c1 = chunk_bounds(t1 - t0, n_chunks, k)
        if c1 > c0:
            ops.append(dist.P2POp(dist.isend, tile[c0:c1], dst, group))
    return dist.batch_isend_irecv(ops) if ops else []


def invert_tiled_device(ctx, inc, sigma0_co, anc, total_lines, *, sigma0_cr=None, dsig_cr=None, dst=0, group=None, n_chunks=8,
                        pipeline=None, algo=None, dsig_co=0.1, dsig_cr_scalar=0.1, sigma0_is_db=False, dual_select=True,
                        out_dtype=None, wait=True):
    """The tiled inversion over DEVICE tensors, one call per rank: this rank's row tile (`inc`, `sigma0_co`, `anc` [, `sigma0_cr`,
    `dsig_cr`]: contiguous torch tensors of one (lines, samples) shape on this rank's GPU, float32 + complex64 or float64 +
    complex128; `tile_bounds(total_lines, world, rank)` says which lines they are) is inverted by libxsw context `ctx` in
    `n_chunks` row chunks on torch's CURRENT stream (the context is handed that stream and stays on it), the grid codes are
    gathered on `dst` chunk by chunk behind the kernels and expanded there on a side stream (`TiledPipeline`).
    Returns the full raster on `dst` -- a complex tensor, or (co, dual) for dual-pol -- and None on the other ranks; with
    wait=False the pipeline itself (call `.finish()` when the result is needed: lets a caller time the two phases).
    `pipeline`: a `TiledPipeline` of the same geometry to reuse (buffers allocated once).  The LUTs must be installed on `ctx`.
    The counterpart of the reference's dask row blocks (windspeed/windspeed.py:350-364) with the concatenation included."""
    from . import _device, _lib
    dual = sigma0_cr is not None
    lines, samples = int(sigma0_co.shape[0]), int(sigma0_co.shape[1])
    dev = sigma0_co.device
    pipe = pipeline or TiledPipeline(total_lines, samples, dual=dual, device=dev, dst=dst, group=group, n_chunks=n_chunks, out_dtype=out_dtype)
    if pipe.rows != lines or pipe.samples != samples or pipe.dual != dual:
        raise ValueError("the pipeline was built for another tile geometry")
    dt = _device.xsw_dtype(sigma0_co)
    item = 4 if dt == _lib.XSW_F32 else 8
    odt = _lib.XSW_F32 if (pipe.full.dtype if pipe.full is not None else (out_dtype or torch.complex64)) == torch.complex64 else _lib.XSW_F64
    oitem = 8 if odt == _lib.XSW_F32 else 16
    algo = _lib.ALGO_AUTO if algo is None else _lib.ALGOS.get(algo, algo)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    at = lambda t, off, size: None if t is None else t.data_ptr() + off * size

    def invert_chunk(k, r0, r1):
        off = r0 * samples
        ctx.invert_raw(r1 - r0, samples, dt, odt, _lib.MEM_DEVICE, at(inc, off, item), at(sigma0_co, off, item), at(sigma0_cr, off, item),
                       at(dsig_cr, off, item), at(anc, off, 2 * item), None, None, None, dsig_co, dsig_cr_scalar, sigma0_is_db, algo,
                       dual_select and dual, out_code_co=at(pipe.codes, off, 4), out_code_cr=at(pipe.codes_dual, off, 4) if dual else None)

    def expand_rows(g0, g1, stream):
        off = g0 * samples
        ctx.expand_codes_on_stream(stream.cuda_stream, (g1 - g0) * samples, odt, at(pipe.full_codes, off, 4),
                                   at(pipe.full_codes_dual, off, 4) if dual else None, at(pipe.full, off, oitem),
                                   at(pipe.full_dual, off, oitem) if dual else None)

    pipe.run(invert_chunk, expand_rows)
    if not wait:
        return pipe
    res = pipe.finish()
    if res is None:
        return None
    return res if dual else res[0]


def invert_from_model_tiled(inc, sigma0, sigma0_dual=None, /, *, dst=0, group=None, invert=None, gather=True, n_chunks=8, **kwargs):
    """`windspeed.invert_from_model` on a raster tiled over the ranks of a `torch.distributed` job (one process per GPU;
    the reference's way to parallelise the same call is dask row blocks, windspeed/windspeed.py:350-364).  (Inside ONE process,
    `xsarsea_amd.options.devices = "all"` spreads the same row tiles over the GPUs without any exchange.)

    Every rank calls this with the SAME full-size array-likes -- numpy arrays or anything sliceable along axis 0 (memory-mapped
    files, lazily loaded arrays: only the rank's own lines `tile_bounds(lines, world, rank)` are touched), or torch CUDA tensors
    / `__cuda_array_interface__` objects on the rank's own GPU (a rank may also pass full-shape rasters of which only its own
    lines are meaningful); `ancillary_wind` and a raster `dsig_cr` in `kwargs` are sliced the same way (arrays of sigma0's rank
    whose first axis has `lines` entries: a 1-D incidence row of a square raster is NOT a raster and is passed whole).  The
    rank inverts its lines on its own GPU (`options.device`, which `xsarsea_amd` sets from LOCAL_RANK) and the result is
    gathered on rank `dst`.

    gather=True, the measured path (`TiledPipeline`; bench.py --gpus N runs the same object): the tile is inverted in
    `n_chunks` row chunks to 4-byte grid codes that never leave device memory; chunk k travels to `dst` (RCCL send/recv under
    the "nccl" backend; through host memory under "gloo") while chunk k + 1 is inverted, and `dst` expands every chunk to
    complex winds on a side stream as its codes land.  Returns what `invert_from_model` returns for the full raster on `dst`
    (numpy in -> numpy out, complex128; device tensors in -> torch tensors on `dst`'s GPU, `options.device_out_dtype`; a
    tuple of two for dual-pol), None on the other ranks.  Without an initialised process group it is the plain call.

    gather=False: no exchange at all -- every rank returns `(l0, l1, result)`, its own lines' result with the return conventions
    of `invert_from_model` (what a dask consumer of row blocks does with them: write its block, reduce it, hand it on).

    Whole-raster preconditions are whole-raster: the reference's "co-pol inversion needs a valid ancillary wind" assertion
    (windspeed.py:107) holds when ANY rank's tile has a valid ancillary value (one flag all-reduced); a tile that is all NaN
    (land) or empty (fewer lines than ranks) yields NaN / no rows instead of raising.  A rank that fails -- while cutting its
    tile, or preparing its inversion (model lookup, LUT install, uploads) -- does not leave the others waiting in a transfer:
    every step that can raise runs inside a `try`, its error flag is all-reduced before the first transfer starts, and every
    rank raises.
    `invert`: the per-tile callable (default `windspeed.invert_from_model`; tests on machines without a GPU pass a stand-in,
    whose arrays are gathered whole through `gather_rows`).
    """
    import numpy as np

    default_invert = invert is None
    if default_invert:
        from .windspeed import invert_from_model as invert
    if not _group_up():
        res = invert(inc, sigma0, *(() if sigma0_dual is None else (sigma0_dual,)), **kwargs)
        return res if gather else (0, int(np.shape(sigma0)[0]) if np.ndim(sigma0) else 0, res)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def agree(*flags):
        """MAX-all-reduce of small integer flags: the one collective between the steps"""
        t = torch.tensor([int(f) for f in flags], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return [int(x) for x in t.tolist()]

    def has_valid(a):
        if a is None:
            return False
        if torch.is_tensor(a):  # device-resident ancillary wind: decided where it lives (NaN if either part is, as numpy's isnan)
            return bool(a.numel()) and bool((~torch.isnan(a)).any().item())
        if hasattr(a, "__cuda_array_interface__"):
            return has_valid(torch.as_tensor(a, device="cuda"))
        return bool(np.size(a)) and bool(np.any(~np.isnan(np.asarray(a))))

    # step 1: cut the tile, look at its ancillary wind
    failure, l0, l1, tile, kw, valid_here, lines = None, 0, 0, None, {}, False, 0
    try:
        lines = int(np.shape(sigma0)[0])
        ndim = np.ndim(sigma0)
        l0, l1 = tile_bounds(lines, world, rank)

        def cut(a):
            is_raster = a is not None and not np.isscalar(a) and np.ndim(a) == ndim and np.shape(a)[0] == lines
            return a[l0:l1] if is_raster else a

        kw = {k: (cut(v) if k in ("ancillary_wind", "dsig_cr") else v) for k, v in kwargs.items()}
        tile = (cut(inc), cut(sigma0)) + (() if sigma0_dual is None else (cut(sigma0_dual),))
        valid_here = has_valid(kw.get("ancillary_wind"))
    except Exception as exc:  # reported to every rank, then re-raised here
        failure = exc
    any_valid_ancillary, failed = agree(valid_here, failure is not None)
    if failure is not None:
        raise failure
    if failed:
        raise RuntimeError("invert_from_model_tiled: another rank failed while cutting its tile (see its traceback)")
    # step 2: the rank's inversion (gathered default call: everything BUT the launches -- those follow the agreement below)
    res, sink = None, None
    try:
        if default_invert:
            # the per-tile call skips the per-call ancillary assertion / warning (`_xsw_tile`): the whole-raster answer is passed in
            kw["_xsw_tile"] = bool(any_valid_ancillary)
            if gather:
                sink = _CodeSink(lines, l1 - l0, dst, group, n_chunks)
                kw["_xsw_codes"] = sink  # the tile's answer as 4-byte grid codes in device memory, chunk by chunk (`TiledPipeline`)
        res = invert(*tile, **kw)
    except Exception as exc:
        failure = exc
    (failed,) = agree(failure is not None)
    if failure is not None:
        raise failure
    if failed:
        raise RuntimeError("invert_from_model_tiled: the inversion failed on another rank (see its traceback); nothing was gathered")
    if not gather:
        return l0, l1, res
    from .windspeed.windspeed import CodedWinds
    if isinstance(res, CodedWinds) and res.launch is not None:
        res.launch()  # chunked inversion + gather + expansion, queued
        out = sink.pipe.finish()
        if rank != dst:
            return None
        shape = sink.full_shape
        if res.on_device:
            ws = [None if t is None else t.reshape(shape) for t in out]
        else:
            ws = [None if t is None else t.cpu().numpy().reshape(shape) for t in out]
        return res.finish_winds(*ws)
    parts = res if isinstance(res, tuple) else (res,)
    outs = []
    for p in parts:
        t = p if torch.is_tensor(p) else torch.as_tensor(np.ascontiguousarray(np.asarray(p)))
        was_tensor = torch.is_tensor(p)
        full = gather_rows(t if was_tensor else t.to(dev), lines, dst=dst, group=group)
        outs.append((full if was_tensor else full.cpu().numpy()) if rank == dst else None)
    if rank != dst:
        return None
    return tuple(outs) if isinstance(res, tuple) else outs[0]


class _CodeSink:
    """What `invert_from_model_tiled` hands to the per-tile call in place of output rasters: the engine (`_engine.invert_coded`)
    tells it the tile's broadcast shape and searches (`begin`), which builds the `TiledPipeline` -- geometry of the GATHERED
    raster from the leading-axis tiling, rows of the pipeline = lines x the middle axes, or single pixels for 1-D rasters --
    and then runs its chunks through `pipe.run`."""

    def __init__(self, total_lines, tile_lines, dst, group, n_chunks):
        self.total_lines, self.tile_lines, self.dst, self.group, self.n_chunks = total_lines, tile_lines, dst, group, n_chunks
        self.pipe, self.full_shape = None, None

    def begin(self, tile_shape, want_co, want_cr, device, out_dtype):
        tile_shape = tuple(int(x) for x in tile_shape)
        if not tile_shape or tile_shape[0] != self.tile_lines:
            raise ValueError(f"the tile's rasters broadcast to {tile_shape}: the leading axis is not the tiled one ({self.tile_lines} lines)")
        rest = tile_shape[1:]
        samples = rest[-1] if rest else 1
        mid = 1
        for x in rest[:-1]:
            mid *= x
        world = dist.get_world_size(self.group)
        # every rank cuts the same number of chunks, of 4 rows or more where the raster allows (the staging ring's minimum):
        # decided from the SMALLEST non-empty tile (`tile_bounds`: every rank but the last, or the last alone)
        base = (self.total_lines // world if self.total_lines >= world else self.total_lines) * mid
        n_chunks = max(1, min(self.n_chunks, base // 4))
        self.full_shape = (self.total_lines,) + rest
        bounds = lambda r: tuple(x * mid for x in tile_bounds(self.total_lines, world, r))
        self.pipe = TiledPipeline(self.total_lines * mid, samples, dual=want_co and want_cr, device=device, dst=self.dst, group=self.group,
                                  n_chunks=n_chunks, out_dtype=out_dtype, bounds=bounds, want_co=want_co)
