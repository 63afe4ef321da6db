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


@pytest.mark.parametrize("lines", [9, 64])
def test_gather_rows_gloo_world2(lines):
    world, samples = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), lines, samples, ret), nprocs=world, join=True)
    assert ret.get("ok") is True
