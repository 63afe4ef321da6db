"""ctypes binding of the plain-C oracle (`oracle/invert_c.c`).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


class _Luts(ctypes.Structure):
    _fields_ = [
        ("co_lut", ctypes.c_void_p), ("wspd_dim", ctypes.c_void_p), ("phi_dim", ctypes.c_void_p),
        ("inc_dim", ctypes.c_void_p), ("lut_antenna", ctypes.c_void_p), ("lut_azi", ctypes.c_void_p),
        ("n_wspd", ctypes.c_int32), ("n_phi", ctypes.c_int32), ("n_inc", ctypes.c_int32),
        ("phi_180", ctypes.c_int32), ("dsig_co", ctypes.c_double),
        ("cr_lut", ctypes.c_void_p), ("wspd_cr", ctypes.c_void_p), ("inc_cr_dim", ctypes.c_void_p),
        ("n_wspd_cr", ctypes.c_int32), ("n_inc_cr", ctypes.c_int32),
        ("co_stride_cand", ctypes.c_int64), ("co_stride_inc", ctypes.c_int64),
        ("cr_stride_cand", ctypes.c_int64), ("cr_stride_inc", ctypes.c_int64),
    ]


def build(force=False):
    src = os.path.join(_HERE, "invert_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_invert.restype = ctypes.c_int
        _lib.oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return lib().oracle_max_threads()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _inc_major(p):
    """Incidence-major copies of the LUTs (cached on `p`): contiguous slices for the fast variant."""
    if not hasattr(p, "_co_incmajor"):
        p._co_incmajor = np.ascontiguousarray(np.moveaxis(p.co_lut, -1, 0)) if p.co_lut.size else p.co_lut
        p._cr_incmajor = np.ascontiguousarray(p.cr_lut.T) if p.cr_lut.size else p.cr_lut
    return p._co_incmajor, p._cr_incmajor


def invert_numpy(p, np_inc, np_s_co_db, np_s_cr_db, np_dsig_cr, np_anc, nthreads=0, return_idx=False,
                 reference_layout=True):
    """Same contract as `oracle.invert.invert_numpy` (p is an `oracle.invert.Prepared`).

    reference_layout=True addresses the LUT as the reference does ((wspd, phi, incidence), strided
    slice gather: the CPU-baseline configuration); False uses an incidence-major copy (identical
    arithmetic and results, ~10x faster: used by the larger parity tests)."""
    shape = np.shape(np_inc)
    f = [np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape), dtype=np.float64).ravel()
         for a in (np_inc, np_s_co_db, np_s_cr_db, np_dsig_cr)]
    anc = np.ascontiguousarray(np.broadcast_to(np.asarray(np_anc), shape), dtype=np.complex128).ravel()
    n = f[0].size
    ncand, ninc, ncr, ninc_cr = len(p.wspd_dim) * len(p.phi_dim), len(p.inc_dim), len(p.wspd_cr), len(p.inc_cr_dim)
    if reference_layout:
        co, cr = p.co_lut, p.cr_lut
        strides = (ninc, 1, ninc_cr, 1)
    else:
        co, cr = _inc_major(p)
        strides = (1, ncand, 1, ncr)
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (
        co, p.wspd_dim, p.phi_dim, p.inc_dim, p.lut_co_antenna, p.lut_co_azi,
        cr, p.wspd_cr, p.inc_cr_dim)]
    L = _Luts(_p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]), _p(keep[5]),
              len(p.wspd_dim), len(p.phi_dim), len(p.inc_dim), int(p.phi_180), float(p.dsig_co),
              _p(keep[6]), _p(keep[7]), _p(keep[8]), len(p.wspd_cr), len(p.inc_cr_dim), *strides)
    out_co = np.empty(n, dtype=np.complex128)
    out_cr = np.empty(n, dtype=np.complex128)
    idx = np.empty((n, 3), dtype=np.int64) if return_idx else None
    rc = lib().oracle_invert(ctypes.byref(L), ctypes.c_int64(n), _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]),
                             _p(anc), _p(out_co), _p(out_cr), _p(idx) if return_idx else None,
                             ctypes.c_int(nthreads))
    assert rc == 0
    res = (out_co.reshape(shape), out_cr.reshape(shape))
    if return_idx:
        res = res + (idx.reshape(shape + (3,)),)
    return res
