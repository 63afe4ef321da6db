"""ctypes binding of libxsw.so (include/xsw.h).  There is no CPU fallback: if the library is missing
or no GPU is present the calls raise."""
import ctypes
import functools
import os
import threading

import numpy as np

from . import _build, _host

XSW_F32, XSW_F64 = 0, 1
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED, MEM_DEVICE_SIGMA0_HOST = 0, 1, 2, 3
CODE_NAN_RE, CODE_NAN, CODE_PICK_CO, CODE_NO_INDEX = 0xFFFFFFFF, 0xFFFFFFFE, 0x40000000, 0x3FFFFFFF
ALGO_AUTO, ALGO_PRUNED, ALGO_EXHAUSTIVE, ALGO_EXACT, ALGO_EXHAUSTIVE_F64 = 0, 1, 2, 3, 4
ALGOS = {"auto": ALGO_AUTO, "pruned": ALGO_PRUNED, "exhaustive": ALGO_EXHAUSTIVE, "exact": ALGO_EXACT,
         "exhaustive_f64": ALGO_EXHAUSTIVE_F64}

GMF_IDS = {"gmf_cmod5": 0, "gmf_cmod5n": 1, "gmf_cmod5n_pr_zhangA": 2, "gmf_cmod5n_pr_mouche1": 3, "gmf_cmodifr2": 4,
           "gmf_rs2_v2": 5, "gmf_s1_v2": 6, "gmf_rcm_noaa": 7, "gmf_s1_v3_ew_rec": 8, "gmf_rs2_v3": 9, "gmf_rcm_v3": 10,
           "gmf_rcm_v4": 11, "gmf_rs2_v4": 12}

EXPORTS = (
    "xsw_version", "xsw_device_count", "xsw_ctx_create", "xsw_ctx_destroy", "xsw_last_error", "xsw_set_stream", "xsw_use_own_stream",
    "xsw_synchronize", "xsw_lut_upload", "xsw_invert", "xsw_stats_enable", "xsw_stats_read", "xsw_stats_read_chain", "xsw_detrend", "xsw_lut_interp", "xsw_gmf_eval",
    "xsw_nesz_flatten", "xsw_lut_build", "xsw_lut_read", "xsw_timing_enable", "xsw_timing_read", "xsw_expand_codes", "xsw_expand_codes_on_stream",
    "xsw_host_alloc", "xsw_host_free", "xsw_set_host_threads",
)


class XswError(RuntimeError):
    pass


class LutStruct(ctypes.Structure):
    _fields_ = [("db", ctypes.c_void_p), ("inc", ctypes.c_void_p), ("wspd", ctypes.c_void_p),
                ("phi", ctypes.c_void_p), ("cos_phi", ctypes.c_void_p), ("sin_phi", ctypes.c_void_p),
                ("out_dir", ctypes.c_void_p), ("abs_co", ctypes.c_void_p), ("dual_dir", ctypes.c_void_p),
                ("n_inc", ctypes.c_int32), ("n_wspd", ctypes.c_int32), ("n_phi", ctypes.c_int32)]


class InvertArgs(ctypes.Structure):
    _fields_ = [("lines", ctypes.c_int64), ("samples", ctypes.c_int64), ("dtype", ctypes.c_int32),
                ("out_dtype", ctypes.c_int32), ("mem", ctypes.c_int32), ("sigma0_is_db", ctypes.c_int32),
                ("algo", ctypes.c_int32), ("dual_select", ctypes.c_int32),
                ("inc", ctypes.c_void_p), ("sigma0_co", ctypes.c_void_p), ("sigma0_cr", ctypes.c_void_p),
                ("dsig_cr", ctypes.c_void_p), ("anc", ctypes.c_void_p),
                ("dsig_co", ctypes.c_double), ("dsig_cr_scalar", ctypes.c_double),
                ("out_co", ctypes.c_void_p), ("out_cr", ctypes.c_void_p), ("out_idx", ctypes.c_void_p),
                ("out_code_co", ctypes.c_void_p), ("out_code_cr", ctypes.c_void_p),
                ("stage", ctypes.c_void_p), ("stage_user", ctypes.c_void_p)]


#: int stage(void *user, int32 which, int64 px0, int64 npx, void *dst)  (xsw_invert_args.stage)
STAGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p)
STAGE_INC, STAGE_SIGMA0_CO, STAGE_SIGMA0_CR, STAGE_DSIG_CR, STAGE_ANC = range(5)


ABI_VERSION = 4  # include/xsw.h: XSW_VERSION


class Stats(ctypes.Structure):
    _fields_ = [("pixels_co", ctypes.c_uint64), ("cand_co", ctypes.c_uint64), ("pixels_exact", ctypes.c_uint64),
                ("pixels_cr", ctypes.c_uint64)]


class ChainStats(ctypes.Structure):
    _fields_ = [("cand_band2", ctypes.c_uint64), ("cand_blocks", ctypes.c_uint64), ("cand_list", ctypes.c_uint64), ("pixels_refined", ctypes.c_uint64)]


class Timing(ctypes.Structure):
    _fields_ = [("launches", ctypes.c_int64), ("first_kernel_ms", ctypes.c_double), ("second_kernel_ms", ctypes.c_double),
                ("last_list_pixels", ctypes.c_int64), ("band2_kernel_ms", ctypes.c_double), ("last_band2_pixels", ctypes.c_int64),
                ("blocks_kernel_ms", ctypes.c_double), ("last_blocks_pixels", ctypes.c_int64)]


_cdll = None


def library_path():
    return _build.LIB


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same soname as the system
    one); if libxsw.so pulled in the system runtime first, a later `import torch` would find "no ROCm-capable
    device".  So when torch is installed but not imported yet, its bundled runtime is loaded first (libxsw.so then
    binds to it); when torch is already imported, or absent, nothing needs doing."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libxsw.so (raises if it has not been built: run `python -m xsarsea_amd._build`)."""
    global _cdll
    if _cdll is None:
        if not os.path.exists(_build.LIB):
            raise XswError(f"{_build.LIB} is missing: build it with `python -m xsarsea_amd._build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _preload_hip_runtime()
        lib = ctypes.CDLL(_build.LIB)
        lib.xsw_last_error.restype = ctypes.c_char_p
        lib.xsw_last_error.argtypes = [ctypes.c_void_p]
        lib.xsw_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.xsw_ctx_destroy.argtypes = [ctypes.c_void_p]
        lib.xsw_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.xsw_use_own_stream.argtypes = [ctypes.c_void_p]
        lib.xsw_synchronize.argtypes = [ctypes.c_void_p]
        lib.xsw_lut_upload.argtypes = [ctypes.c_void_p, ctypes.POINTER(LutStruct), ctypes.POINTER(LutStruct)]
        lib.xsw_invert.argtypes = [ctypes.c_void_p, ctypes.POINTER(InvertArgs)]
        lib.xsw_stats_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.xsw_stats_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
        lib.xsw_stats_read_chain.argtypes = [ctypes.c_void_p, ctypes.POINTER(ChainStats)]
        lib.xsw_detrend.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.xsw_lut_interp.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int32] * 3 + \
            [ctypes.c_void_p] * 3 + [ctypes.c_int32] * 3 + [ctypes.c_void_p]
        lib.xsw_gmf_eval.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4
        lib.xsw_nesz_flatten.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 3
        lib.xsw_lut_build.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(LutStruct)]
        lib.xsw_lut_read.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        lib.xsw_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.xsw_timing_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(Timing)]
        lib.xsw_expand_codes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4
        lib.xsw_expand_codes_on_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4
        lib.xsw_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.xsw_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.xsw_set_host_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        if lib.xsw_version() != ABI_VERSION:
            raise XswError(f"{_build.LIB} is version {lib.xsw_version()}, this package binds version {ABI_VERSION}: rebuild it")
        _cdll = lib
    return _cdll


def device_count():
    return load().xsw_device_count()


def device_count_safe():
    """0 when the library is not built (host-only uses such as LUT preparation in a CPU-only session)."""
    try:
        return device_count()
    except (XswError, OSError):
        return 0


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return ctypes.c_void_p(a.ctypes.data)


def _locked(method):
    """A libxsw context is not thread-safe (include/xsw.h); the Python layer serialises the calls on one context, so
    that threaded callers (dask's threaded scheduler runs the reference's function from several threads) stay correct.
    The lock is re-entrant and is also taken by `windspeed._engine` around "make sure the LUTs are up, then invert"."""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with self.lock:
            return method(self, *args, **kwargs)
    return wrapper


class Context:
    """One device, one stream (include/xsw.h: xsw_ctx)."""

    def __init__(self, device=0):
        self._lib = load()
        h = ctypes.c_void_p()
        rc = self._lib.xsw_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise XswError(f"xsw_ctx_create failed ({rc}): {self._lib.xsw_last_error(None).decode()}")
        self._h = h
        self.device = int(device)
        self.lut_key = (None, None)
        self.lock = threading.RLock()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.xsw_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise XswError(f"{what} failed ({rc}): {self._lib.xsw_last_error(self._h).decode()}")

    @_locked
    def set_stream(self, stream_handle):
        """Launch on this hipStream_t handle (int; 0 = the device's default stream)."""
        self._check(self._lib.xsw_set_stream(self._h, ctypes.c_void_p(stream_handle or 0)), "xsw_set_stream")

    @_locked
    def use_own_stream(self):
        self._check(self._lib.xsw_use_own_stream(self._h), "xsw_use_own_stream")

    @_locked
    def synchronize(self):
        self._check(self._lib.xsw_synchronize(self._h), "xsw_synchronize")

    @staticmethod
    def _lut_struct(db, inc, wspd, phi=None, cos_phi=None, sin_phi=None, out_dir=None, abs_co=None, dual_dir=None):
        keep = [None if db is None else _f64(db), _f64(inc), _f64(wspd)] + [None if v is None else _f64(v) for v in
                                                                            (phi, cos_phi, sin_phi, out_dir, abs_co, dual_dir)]
        n_phi = 0 if phi is None else len(keep[3])
        n_w = len(keep[2])
        expect = (len(keep[1]), n_w) + ((n_phi,) if phi is not None else ())
        if keep[0] is not None and keep[0].shape != expect:
            raise ValueError(f"LUT shape {keep[0].shape} does not match axes {expect}")
        for a, shp, nm in ((keep[4], (n_phi,), "cos_phi"), (keep[5], (n_phi,), "sin_phi"),
                           (keep[6], (2, n_phi, 2), "out_dir"), (keep[7], (n_w, n_phi), "abs_co"),
                           (keep[8], (2, n_w, n_phi, 2), "dual_dir")):
            if a is not None and a.shape != shp:
                raise ValueError(f"{nm} has shape {a.shape}, expected {shp}")
        s = LutStruct(*[_ptr(k) for k in keep], len(keep[1]), n_w, n_phi)
        return s, keep

    @_locked
    def upload_luts(self, co=None, cr=None):
        """co = dict(db[inc,wspd,phi], inc, wspd, phi[, cos_phi, sin_phi]); cr = dict(db[inc,wspd], inc, wspd)."""
        sco = scr = None
        keep = []
        if co is not None:
            sco, k = self._lut_struct(**co)
            keep.append(k)
        if cr is not None:
            scr, k = self._lut_struct(**cr)
            keep.append(k)
        self._check(self._lib.xsw_lut_upload(self._h, ctypes.byref(sco) if sco else None,
                                             ctypes.byref(scr) if scr else None), "xsw_lut_upload")

    @_locked
    def build_lut(self, gmf_id, raw_axes, target):
        """xsw_lut_build: raw_axes = (inc, wspd, phi or None) of the model's own grid; target = dict(inc, wspd[, phi, cos_phi,
        sin_phi, out_dir, abs_co, dual_dir]) of the grid to search on.  The table is built and kept on the device."""
        inc_r, wspd_r = _f64(raw_axes[0]), _f64(raw_axes[1])
        phi_r = None if raw_axes[2] is None else _f64(raw_axes[2])
        st, keep = self._lut_struct(None, **target)
        self._check(self._lib.xsw_lut_build(self._h, int(gmf_id), _ptr(inc_r), len(inc_r), _ptr(wspd_r), len(wspd_r), _ptr(phi_r),
                                            0 if phi_r is None else len(phi_r), ctypes.byref(st)), "xsw_lut_build")
        del keep

    @_locked
    def read_lut(self, shape, cross=False):
        """xsw_lut_read: the context's current dB table as a host array of `shape` (the caller knows the axes)."""
        out = np.empty(shape, dtype=np.float64)
        self._check(self._lib.xsw_lut_read(self._h, int(bool(cross)), _ptr(out)), "xsw_lut_read")
        return out

    @_locked
    def lut_interp(self, raw, inc_raw, wspd_raw, phi_raw, inc, wspd, phi):
        """xsw_lut_interp: (incidence, wspd[, phi]) table -> finer axes, bit-identical to three interp1d passes."""
        raw, inc_raw, wspd_raw, inc, wspd = map(_f64, (raw, inc_raw, wspd_raw, inc, wspd))
        has_phi = phi_raw is not None
        phi_raw = _f64(phi_raw) if has_phi else None
        phi = _f64(phi) if has_phi else None
        out = np.empty((len(inc), len(wspd)) + ((len(phi),) if has_phi else ()), dtype=np.float64)
        self._check(self._lib.xsw_lut_interp(
            self._h, _ptr(raw), _ptr(inc_raw), _ptr(wspd_raw), _ptr(phi_raw), len(inc_raw), len(wspd_raw),
            len(phi_raw) if has_phi else 0, _ptr(inc), _ptr(wspd), _ptr(phi), len(inc), len(wspd),
            len(phi) if has_phi else 0, _ptr(out)), "xsw_lut_interp")
        return out

    @_locked
    def gmf_eval(self, gmf_id, inc, wspd, phi=None):
        """xsw_gmf_eval on host float64 arrays of one common shape."""
        inc = _f64(inc)
        wspd = _f64(wspd)
        phi = None if phi is None else _f64(phi)
        if wspd.shape != inc.shape or (phi is not None and phi.shape != inc.shape):
            raise ValueError("gmf_eval wants already-broadcast arrays of one shape")
        out = np.empty(inc.shape, dtype=np.float64)
        self._check(self._lib.xsw_gmf_eval(self._h, int(gmf_id), inc.size, MEM_HOST, _ptr(inc), _ptr(wspd), _ptr(phi),
                                           _ptr(out)), "xsw_gmf_eval")
        return out

    @_locked
    def gmf_eval_raw(self, gmf_id, n, mem, inc_ptr, wspd_ptr, phi_ptr, out_ptr):
        """Thin call of xsw_gmf_eval (pointers are ints or None: float64 arrays of n already-broadcast elements; device
        pointers with MEM_DEVICE: asynchronous on the context's stream)."""
        self._check(self._lib.xsw_gmf_eval(self._h, int(gmf_id), int(n), mem, _ptr(inc_ptr), _ptr(wspd_ptr), _ptr(phi_ptr), _ptr(out_ptr)),
                    "xsw_gmf_eval")

    @_locked
    def timing_enable(self, on=True):
        self._check(self._lib.xsw_timing_enable(self._h, int(bool(on))), "xsw_timing_enable")

    @_locked
    def timing(self):
        """xsw_timing_read: dict(launches, first_kernel_ms, second_kernel_ms) summed since the last read."""
        t = Timing()
        self._check(self._lib.xsw_timing_read(self._h, ctypes.byref(t)), "xsw_timing_read")
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    @_locked
    def stats_enable(self, on=True):
        """on = True: the statistics instantiation (every window swept in k_invert_band, every candidate counted); on = 2: the
        production chain with per-kernel counters (`stats_chain`)."""
        self._check(self._lib.xsw_stats_enable(self._h, 2 if on == 2 else int(bool(on))), "xsw_stats_enable")

    @_locked
    def stats_chain(self):
        s = ChainStats()
        self._check(self._lib.xsw_stats_read_chain(self._h, ctypes.byref(s)), "xsw_stats_read_chain")
        return {k: int(getattr(s, k)) for k, _ in ChainStats._fields_}

    @_locked
    def stats(self):
        s = Stats()
        self._check(self._lib.xsw_stats_read(self._h, ctypes.byref(s)), "xsw_stats_read")
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    @_locked
    def invert_raw(self, lines, samples, dtype, out_dtype, mem, inc, sigma0_co, sigma0_cr, dsig_cr, anc, out_co,
                   out_cr, out_idx=None, dsig_co=0.1, dsig_cr_scalar=0.1, sigma0_is_db=False, algo=ALGO_AUTO,
                   dual_select=False, out_code_co=None, out_code_cr=None, stage=None):
        """Thin call of xsw_invert; pointer arguments are ints (device or host addresses) or None.
        stage: python callable (which, px0, npx, dst_address) -> 1 filled / 0 default copy (xsw_invert_args.stage); an
        exception inside it aborts the call and is re-raised here."""
        cb, failure = None, []
        if stage is not None:
            def _cb(_user, which, px0, npx, dst):
                try:
                    return 1 if stage(which, px0, npx, dst) else 0
                except BaseException as exc:  # nothing propagates through the C frames
                    failure.append(exc)
                    return -1
            cb = STAGE_FN(_cb)
        a = InvertArgs(int(lines), int(samples), dtype, out_dtype, mem, int(bool(sigma0_is_db)), int(algo),
                       int(bool(dual_select)), inc, sigma0_co, sigma0_cr, dsig_cr, anc, float(dsig_co),
                       float(dsig_cr_scalar), out_co, out_cr, out_idx, out_code_co, out_code_cr,
                       ctypes.cast(cb, ctypes.c_void_p) if cb is not None else None, None)
        rc = self._lib.xsw_invert(self._h, ctypes.byref(a))
        if failure:
            raise failure[0]
        self._check(rc, "xsw_invert")

    @_locked
    def expand_codes_raw(self, n, mem, out_dtype, code_co, code_cr, out_co, out_cr):
        """Thin call of xsw_expand_codes (pointers are ints or None): grid codes -> the complex winds xsw_invert stores."""
        self._check(self._lib.xsw_expand_codes(self._h, int(n), mem, out_dtype, code_co, code_cr, out_co, out_cr), "xsw_expand_codes")

    def expand_codes_on_stream(self, stream, n, out_dtype, code_co, code_cr, out_co, out_cr):
        """xsw_expand_codes_on_stream: device codes -> device winds on `stream` (a HIP stream handle as an int), the context's
        launch stream untouched."""
        self._check(self._lib.xsw_expand_codes_on_stream(self._h, ctypes.c_void_p(int(stream)), int(n), out_dtype, code_co, code_cr, out_co, out_cr),
                    "xsw_expand_codes_on_stream")

    def expand_codes_host(self, codes_co, codes_cr, out_dtype=np.complex128):
        """Grid codes (uint32 arrays of one shape; either may be None) -> (ws_co, ws_cr) on the host, block-wise on the host
        thread pool (xsw_expand_codes with XSW_MEM_HOST reads the context's host tables only: the blocks run side by side; the
        context's lock is held for the whole expansion so that no LUT is installed meanwhile)."""
        ref = codes_co if codes_co is not None else codes_cr
        shape, n = ref.shape, ref.size
        od = XSW_F32 if np.dtype(out_dtype) == np.complex64 else XSW_F64
        cc = None if codes_co is None else np.ascontiguousarray(codes_co, dtype=np.uint32).reshape(-1)
        cr = None if codes_cr is None else np.ascontiguousarray(codes_cr, dtype=np.uint32).reshape(-1)
        o_co = None if cc is None else np.empty(n, out_dtype)
        o_cr = None if cr is None else np.empty(n, out_dtype)
        item = np.dtype(out_dtype).itemsize
        at = lambda a, i, size: None if a is None else ctypes.c_void_p(a.ctypes.data + i * size)

        def work(i):
            m = min(_host.BLOCK, n - i)
            rc = self._lib.xsw_expand_codes(self._h, m, MEM_HOST, od, at(cc, i, 4), at(cr, i, 4), at(o_co, i, item), at(o_cr, i, item))
            if rc != 0:
                raise XswError(f"xsw_expand_codes failed ({rc}): {self._lib.xsw_last_error(self._h).decode()}")

        with self.lock:
            if n:
                list(_host.pool().map(work, range(0, n, _host.BLOCK)))
        return (None if o_co is None else o_co.reshape(shape)), (None if o_cr is None else o_cr.reshape(shape))

    @_locked
    def set_host_threads(self, n):
        """Worker threads of the host-memory paths (0 = default: XSW_HOST_THREADS or 12)."""
        self._check(self._lib.xsw_set_host_threads(self._h, int(n)), "xsw_set_host_threads")

    @_locked
    def pinned_empty(self, shape, dtype):
        """numpy array in page-locked memory of this context (xsw_host_alloc): rasters filled here can be handed to the
        XSW_MEM_HOST_PINNED paths, which DMA straight out of them.  The memory lives until `pinned_free(arr)` or the context closes."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = ctypes.c_void_p()
        self._check(self._lib.xsw_host_alloc(self._h, max(nbytes, 1), ctypes.byref(p)), "xsw_host_alloc")
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    @_locked
    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self._lib.xsw_host_free(self._h, ctypes.c_void_p(p)), "xsw_host_free")

    @_locked
    def invert_host(self, inc, sigma0_co=None, sigma0_cr=None, dsig_cr=None, anc=None, dsig_co=0.1,
                    sigma0_is_db=False, algo="auto", dual_select=False, out_dtype=np.complex128, want_idx=False, want_codes=False,
                    pinned=False, out_co=None, out_cr=None, stage=None, want_complex=True):
        """numpy-in / numpy-out wrapper of xsw_invert for host rasters of one dtype (float32 or float64).
        want_codes: also return the uint32 grid codes (co, cr) as a 4th element.  pinned: the rasters are page-locked
        (`pinned_empty`): XSW_MEM_HOST_PINNED.  out_co / out_cr: C-contiguous arrays of the broadcast shape and `out_dtype` to
        write into (row tiles of one raster inverted by several contexts land in place).  stage: see `invert_raw`.
        want_complex=False (with want_codes): only the grid codes are produced (nothing is expanded on the host)."""
        inc = np.asarray(inc)
        dt = inc.dtype
        if dt not in (np.float32, np.float64):
            raise TypeError("raster dtype must be float32 or float64")
        cdt = np.complex64 if dt == np.float32 else np.complex128
        shape = np.broadcast_shapes(*(np.shape(a) for a in (inc, sigma0_co, sigma0_cr, anc, None if np.isscalar(dsig_cr) else dsig_cr)
                                      if a is not None))
        inc = np.ascontiguousarray(np.broadcast_to(inc, shape))
        n = inc.size
        lines, samples = (int(np.prod(shape[:-1])), shape[-1]) if inc.ndim >= 1 and n else (0, 0)
        if inc.ndim == 0:
            lines, samples = 1, 1

        def prep(a, t):
            if a is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape), dtype=t)
            return a

        s_co, s_cr, anc_ = prep(sigma0_co, dt), prep(sigma0_cr, dt), prep(anc, cdt)
        dsig_scalar = 0.1
        dsig_arr = None
        if dsig_cr is not None:
            if np.isscalar(dsig_cr):
                dsig_scalar = float(dsig_cr)
            else:
                dsig_arr = prep(dsig_cr, dt)
        out_dtype = np.dtype(out_dtype)
        # plain np.empty: the library's worker threads write (and so first-touch) their own chunks side by side
        for o in (out_co, out_cr):
            if o is not None and (o.shape != tuple(shape) or o.dtype != out_dtype or not o.flags.c_contiguous):
                raise ValueError("out_co / out_cr must be C-contiguous arrays of the broadcast shape and out_dtype")
        if s_co is not None and out_co is None and want_complex:
            out_co = np.empty(shape, out_dtype)
        if s_cr is not None and out_cr is None and want_complex:
            out_cr = np.empty(shape, out_dtype)
        if s_co is None:
            out_co = None
        if s_cr is None:
            out_cr = None
        idx = np.empty(shape + (3,), dtype=np.int32) if want_idx else None
        codes = (np.empty(shape, np.uint32) if s_co is not None else None,
                 np.empty(shape, np.uint32) if s_cr is not None else None) if want_codes else (None, None)
        if n:
            self.invert_raw(lines, samples, XSW_F32 if dt == np.float32 else XSW_F64,
                            XSW_F32 if out_dtype == np.complex64 else XSW_F64, MEM_HOST_PINNED if pinned else MEM_HOST,
                            _ptr(inc), _ptr(s_co), _ptr(s_cr), _ptr(dsig_arr), _ptr(anc_), _ptr(out_co), _ptr(out_cr),
                            _ptr(idx), dsig_co, dsig_scalar, sigma0_is_db, ALGOS.get(algo, algo), dual_select,
                            _ptr(codes[0]), _ptr(codes[1]), stage)
        if want_codes:
            return out_co, out_cr, idx, codes
        return out_co, out_cr, idx

    @_locked
    def nesz_flatten_raw(self, lines, samples, dtype, mem, noise_ptr, inc_ptr, out_ptr):
        """Thin call of xsw_nesz_flatten (pointers are ints: device or host addresses)."""
        self._check(self._lib.xsw_nesz_flatten(self._h, int(lines), int(samples), dtype, mem, ctypes.c_void_p(noise_ptr),
                                               ctypes.c_void_p(inc_ptr), ctypes.c_void_p(out_ptr)), "xsw_nesz_flatten")

    @_locked
    def nesz_flatten_host(self, noise, inc):
        """xsw_nesz_flatten on host rasters of one dtype (float32 or float64) and one 2-D shape -> float64."""
        noise, inc = np.asarray(noise), np.asarray(inc)
        # one raster dtype on the device: the wider of the two (a float64 incidence is never narrowed to a float32 noise raster)
        dt = np.result_type(noise.dtype, inc.dtype)
        if dt not in (np.float32, np.float64):
            dt = np.dtype(np.float64)
        noise = np.ascontiguousarray(noise, dtype=dt)
        inc = np.ascontiguousarray(inc, dtype=dt)
        if noise.ndim != 2 or inc.shape != noise.shape:
            raise ValueError("noise and inc must be 2-D rasters of one shape")
        out = np.empty(noise.shape, np.float64)
        if noise.size:
            self.nesz_flatten_raw(noise.shape[0], noise.shape[1], XSW_F32 if noise.dtype == np.float32 else XSW_F64, MEM_HOST,
                                  noise.ctypes.data, inc.ctypes.data, out.ctypes.data)
        return out

    @_locked
    def detrend_raw(self, lines, samples, dtype, out_dtype, mem, sigma0_ptr, ratio_row, out_ptr):
        """Thin call of xsw_detrend (raster pointers are ints; ratio_row is a host float64 array)."""
        ratio_row = _f64(ratio_row)
        self._check(self._lib.xsw_detrend(self._h, int(lines), int(samples), dtype, out_dtype, mem,
                                          ctypes.c_void_p(sigma0_ptr), _ptr(ratio_row), ctypes.c_void_p(out_ptr)),
                    "xsw_detrend")

    @_locked
    def detrend_host(self, sigma0, ratio_row, out_dtype=np.float64):
        sigma0 = np.ascontiguousarray(sigma0)
        if sigma0.dtype not in (np.float32, np.float64):
            sigma0 = sigma0.astype(np.float64)
        ratio_row = _f64(ratio_row)
        lines, samples = int(np.prod(sigma0.shape[:-1])), sigma0.shape[-1]
        if ratio_row.shape != (samples,):
            raise ValueError("ratio_row must have one value per sample")
        out = np.empty(sigma0.shape, out_dtype)
        self._check(self._lib.xsw_detrend(self._h, lines, samples, XSW_F32 if sigma0.dtype == np.float32 else XSW_F64,
                                          XSW_F32 if out.dtype == np.float32 else XSW_F64, MEM_HOST,
                                          _ptr(sigma0), _ptr(ratio_row), _ptr(out)), "xsw_detrend")
        return out


_default_ctx = {}
_default_ctx_lock = threading.Lock()


def default_context(device=0, replica=0):
    """Process-wide context per device (created on first use).  replica > 0: further contexts on the same device
    (`options.devices = [0, 0]`: the single-process multi-device path rehearsed on one GPU)."""
    key = (int(device), int(replica))
    with _default_ctx_lock:
        if key not in _default_ctx:
            _default_ctx[key] = Context(device)
        return _default_ctx[key]


def contexts_for(devices):
    """One context per entry of `devices` (a device listed twice gets two contexts)."""
    seen = {}
    out = []
    for d in devices:
        k = seen.get(d, 0)
        seen[d] = k + 1
        out.append(default_context(d, k))
    return out
