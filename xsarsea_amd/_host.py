"""Host-side helpers: a small thread pool for the elementwise numpy passes that stay on the CPU (numpy ufuncs and
slice assignments release the GIL) and parallel first-touch of large output rasters."""
import os

import numpy as np

BLOCK = 1 << 20  # elements per task
_POOL = None


def pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:  # pragma: no cover
            n = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, n)))
    return _POOL


def empty_touched(shape, dtype):
    """np.empty whose pages are already mapped: a fresh multi-GB array costs one page fault per 4 KiB when the
    device-to-host copy first writes it, serially; here the faults are taken by the pool's threads side by side
    (one store per page; the contents stay undefined, as with np.empty)."""
    out = np.empty(shape, dtype=dtype)
    if out.nbytes < (64 << 20):
        return out
    flat = out.reshape(-1).view(np.uint8)
    step = 32 << 20  # bytes per task

    def touch(i):
        flat[i:i + step:4096] = 0

    list(pool().map(touch, range(0, flat.size, step)))
    return out
