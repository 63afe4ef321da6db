"""Minimal duck-typed stand-ins for `xarray` and `dask.array` (neither is installed in the build or GPU images), so that
the xarray-in/xarray-out and dask branches of the drop-in API execute in CI.  They implement exactly the surface the
product (and the reference, windspeed/windspeed.py:333-439, detrend.py:55-66) touches:

  xarray:      DataArray(data, dims=, coords=, name=, attrs=) with .data (settable) .values .dims .shape .dtype .name .attrs
               .coords, attribute access to coordinates (`sigma0.pol.values.item()`), .isel(), .copy(data=), .astype(),
               numpy ufuncs and the comparison / bit-wise operators (attrs kept, as xarray's defaults do for ufuncs), and the
               module functions zeros_like (TypeError for anything that is not a DataArray -- the reference relies on it)
               and where (attrs dropped, as xarray's default).
  dask.array:  a lazy Array (nothing runs before .compute() / np.asarray), from_array(x, chunks=(rows, -1)) and
               apply_gufunc(func, signature, *args, output_dtypes=) evaluating `func` once per row block.

This is test infrastructure: `conftest.xr_env` binds them into the product modules only when the real packages are absent.
"""
import operator
import types

import numpy as np


# ----------------------------------------------------------------------------------------------- dask.array
class Array:
    """Lazy array: `thunk()` produces the numpy value on first use."""

    def __init__(self, thunk, shape, dtype, chunks=None):
        self._thunk, self._value = thunk, None
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.ndim = len(self.shape)
        self.chunks = chunks if chunks is not None else tuple((n,) for n in self.shape)
        self.computed = False

    def compute(self):
        if self._value is None:
            self._value = np.asarray(self._thunk())
            self.computed = True
            self._thunk = None
        return self._value

    def __array__(self, dtype=None, copy=None):
        v = self.compute()
        return v if dtype is None else v.astype(dtype)

    def astype(self, dtype):
        return Array(lambda: self.compute().astype(dtype), self.shape, dtype, self.chunks)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return getattr(ufunc, method)(*(np.asarray(x) if isinstance(x, Array) else x for x in inputs), **kwargs)
        return _lazy(lambda *xs: ufunc(*xs, **kwargs), *inputs)

    def __invert__(self):
        return _lazy(np.invert, self)

    def __getitem__(self, key):
        shape = np.empty(self.shape, dtype=np.bool_)[key].shape
        return Array(lambda: self.compute()[key], shape, self.dtype)


def _lazy(func, *operands):
    shape = np.broadcast_shapes(*(np.shape(o) for o in operands))
    probe = func(*(np.zeros((1,) * np.ndim(o), dtype=o.dtype if hasattr(o, "dtype") else np.asarray(o).dtype) for o in operands))
    chunks = next((o.chunks for o in operands if isinstance(o, Array) and o.shape == shape), None)
    return Array(lambda: func(*(o.compute() if isinstance(o, Array) else o for o in operands)), shape,
                 np.asarray(probe).dtype, chunks)


for _name, _op in (("__lt__", operator.lt), ("__le__", operator.le), ("__gt__", operator.gt), ("__ge__", operator.ge),
                   ("__or__", operator.or_), ("__and__", operator.and_), ("__add__", operator.add), ("__mul__", operator.mul),
                   ("__sub__", operator.sub), ("__truediv__", operator.truediv), ("__pow__", operator.pow)):
    setattr(Array, _name, (lambda op: lambda self, other: _lazy(op, self, other))(_op))
    if _name in ("__add__", "__mul__", "__sub__", "__truediv__", "__or__", "__and__"):
        setattr(Array, "__r" + _name[2:], (lambda op: lambda self, other: _lazy(op, other, self))(_op))


def from_array(x, chunks):
    x = np.asarray(x)
    rows = chunks[0] if isinstance(chunks, (tuple, list)) else chunks
    if x.ndim == 0 or rows in (-1, None):
        ch = tuple((n,) for n in x.shape)
    else:
        first = tuple(min(rows, x.shape[0] - r) for r in range(0, x.shape[0], rows)) or (0,)
        ch = (first,) + tuple((n,) for n in x.shape[1:])
    return Array(lambda: x, x.shape, x.dtype, ch)


def apply_gufunc(func, signature, *args, output_dtypes=None, **_kw):
    """`func` maps row blocks of the inputs (core dimension = last axis, never chunked) to row blocks of the outputs."""
    n_out = len(signature.split("->")[1].split(","))
    shape = np.broadcast_shapes(*(np.shape(a) for a in args))
    rows = next((a.chunks[0] for a in args if isinstance(a, Array) and a.ndim == len(shape) and a.ndim >= 2), (shape[0],) if shape else ())
    cache = {}

    def evaluate():
        if "out" not in cache:
            full = [np.broadcast_to(a.compute() if isinstance(a, Array) else np.asarray(a), shape) for a in args]
            pieces, r0 = [], 0
            for n in (rows if len(shape) >= 2 else (None,)):
                sl = slice(None) if n is None else slice(r0, r0 + n)
                pieces.append(func(*(f[sl] for f in full)))
                r0 += 0 if n is None else n
            cache["out"] = [np.concatenate([p[k] for p in pieces]) if len(shape) >= 2 else pieces[0][k] for k in range(n_out)]
            cache["calls"] = len(pieces)
        return cache["out"]

    dts = output_dtypes if isinstance(output_dtypes, (tuple, list)) else (output_dtypes,) * n_out
    outs = tuple(Array((lambda k: lambda: evaluate()[k])(k), shape, dts[k], (tuple(rows),) + tuple((n,) for n in shape[1:]) if len(shape) >= 2 else None)
                 for k in range(n_out))
    for o in outs:
        o.block_calls = lambda: cache.get("calls", 0)
    return outs if n_out > 1 else outs[0]


def make_dask_array_module():
    m = types.ModuleType("dask.array")
    m.Array, m.from_array, m.apply_gufunc = Array, from_array, apply_gufunc
    return m


# ----------------------------------------------------------------------------------------------- xarray
class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.data = data if isinstance(data, Array) else np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(np.ndim(self.data)))
        self.coords = dict(coords or {})
        self.name = name
        self.attrs = dict(attrs or {})

    shape = property(lambda self: tuple(self.data.shape))
    dtype = property(lambda self: self.data.dtype)
    ndim = property(lambda self: len(self.data.shape))
    values = property(lambda self: np.asarray(self.data))

    def __array__(self, dtype=None, copy=None):
        v = np.asarray(self.data)
        return v if dtype is None else v.astype(dtype)

    def __getattr__(self, name):  # coordinates as attributes: sigma0.pol
        coords = self.__dict__.get("coords", {})
        if name in coords:
            return DataArray(coords[name])
        raise AttributeError(name)

    def __getitem__(self, key):
        if isinstance(key, str):
            return DataArray(self.coords[key], dims=(key,))
        return self._like(self.data[key])

    def item(self):
        return np.asarray(self.data).item()

    def _like(self, data, keep_attrs=True, name=True):
        return DataArray(data, dims=self.dims if np.ndim(data) == len(self.dims) else None, coords=self.coords,
                         name=self.name if name else None, attrs=self.attrs if keep_attrs else None)

    def isel(self, **idx):
        data, dims = self.data, list(self.dims)
        for d, i in idx.items():
            ax = dims.index(d)
            data = data[(slice(None),) * ax + (i,)]
            if np.isscalar(i) or isinstance(i, (int, np.integer)):
                dims.pop(ax)
        return DataArray(data, dims=dims, coords=self.coords, name=self.name, attrs=self.attrs)

    def copy(self, deep=True, data=None):
        d = self.data if data is None else data
        if data is None and not isinstance(d, Array):
            d = d.copy()
        if tuple(np.shape(d)) != self.shape:
            raise ValueError("copy(data=) must keep the shape")
        return self._like(d)

    def astype(self, dtype):
        return self._like(self.data.astype(dtype))

    def __invert__(self):
        return self.__array_ufunc__(np.invert, "__call__", self)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":  # reductions (np.any, np.all, ...): computed eagerly on the values
            return getattr(ufunc, method)(*(np.asarray(x) if isinstance(x, (DataArray, Array)) else x for x in inputs), **kwargs)
        raw = [x.data if isinstance(x, DataArray) else x for x in inputs]
        res = _lazy(lambda *xs: ufunc(*xs, **kwargs), *raw) if any(isinstance(x, Array) for x in raw) else ufunc(*raw, **kwargs)
        return self._like(res)  # xarray keeps attrs for ufuncs (keep_attrs default of __array_ufunc__)

    def _binary(self, other, op, reflected=False):
        a, b = self.data, (other.data if isinstance(other, DataArray) else other)
        if reflected:
            a, b = b, a
        lazy = isinstance(a, Array) or isinstance(b, Array)
        return self._like(_lazy(op, a, b) if lazy else op(a, b), keep_attrs=False)  # binary ops drop attrs by default


for _name, _op in (("__lt__", operator.lt), ("__le__", operator.le), ("__gt__", operator.gt), ("__ge__", operator.ge),
                   ("__or__", operator.or_), ("__and__", operator.and_), ("__add__", operator.add), ("__mul__", operator.mul),
                   ("__sub__", operator.sub), ("__truediv__", operator.truediv), ("__pow__", operator.pow)):
    setattr(DataArray, _name, (lambda op: lambda self, other: self._binary(other, op))(_op))
    if _name in ("__add__", "__mul__", "__sub__", "__truediv__"):
        setattr(DataArray, "__r" + _name[2:], (lambda op: lambda self, other: self._binary(other, op, True))(_op))


def zeros_like(other, dtype=None):
    if not isinstance(other, DataArray):
        raise TypeError(f"Expected DataArray, Dataset, or Variable, got {type(other)}")  # xarray's behaviour; the reference relies on it
    dt = other.dtype if dtype is None else dtype
    if isinstance(other.data, Array):
        shape = other.shape
        data = Array(lambda: np.zeros(shape, dtype=dt), shape, dt, other.data.chunks)
    else:
        data = np.zeros(other.shape, dtype=dt)
    return other._like(data)


def where(cond, x, y):
    tmpl = next(v for v in (cond, x, y) if isinstance(v, DataArray))
    raw = [v.data if isinstance(v, DataArray) else v for v in (cond, x, y)]
    res = _lazy(np.where, *raw) if any(isinstance(v, Array) for v in raw) else np.where(*raw)
    return tmpl._like(res, keep_attrs=False)


def make_xarray_module():
    m = types.ModuleType("xarray")
    m.DataArray, m.zeros_like, m.where = DataArray, zeros_like, where
    m.__standin__ = True
    return m
