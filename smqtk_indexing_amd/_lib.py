"""
ctypes binding of ``libsmqtk_hip.so`` (C ABI: ``include/smqtk_hip.h``).

There is deliberately NO fallback: if the shared library is missing or a call
fails, a :class:`HipError` is raised.  Plugin classes report
``is_usable() == False`` when the library or a GPU is absent.

Buffers: host calls take C-contiguous numpy arrays; device calls take raw
device pointers (``int``), e.g. ``torch.Tensor.data_ptr()``, plus a HIP stream
handle (``torch.cuda.current_stream().cuda_stream``).  ctypes releases the GIL
for the duration of every call.
"""
import ctypes
import os
import threading
from typing import Optional, Tuple

import numpy as np

SQ_OK = 0
SQ_MEM_HOST = 0
SQ_MEM_DEVICE = 1
SQ_MEM_DEVICE_ASYNC = 2
SQ_METRIC_L2 = 0
SQ_METRIC_COSINE = 1
SQ_DTYPE_F32 = 0
SQ_DTYPE_F64 = 1
SQ_NORM_NONE = -1
SQ_NORM_L0 = 0
SQ_NORM_L1 = 1
SQ_NORM_L2 = 2
SQ_NORM_INF = 1000
SQ_NORM_NEG_INF = -1000
SQ_MAX_K = 16384

LIB_NAME = "libsmqtk_hip.so"
# SMQTK_HIP_LIBRARY: another build of the same library (measurement: kernel variants side by side)
LIB_PATH = os.environ.get("SMQTK_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# every symbol include/smqtk_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "sq_last_error", "sq_version", "sq_device_count", "sq_device_name",
    "sq_set_option", "sq_handle_set_option", "sq_handle_reset_options", "sq_get_stats", "sq_itq_hash",
    "sq_itq_model_create", "sq_itq_model_hash", "sq_itq_model_destroy",
    "sq_hamming_create", "sq_hamming_search", "sq_hamming_sync", "sq_hamming_append", "sq_hamming_remove", "sq_hamming_info", "sq_hamming_destroy",
    "sq_dense_create", "sq_dense_create_opts", "sq_dense_info", "sq_dense_append", "sq_dense_search", "sq_dense_sync", "sq_dense_destroy",
    "sq_dense_distances", "sq_merge_topk", "sq_merge_topk_strided",
    "sq_rows_create", "sq_rows_append", "sq_rows_rerank", "sq_rows_set_buckets", "sq_lsh_query", "sq_rows_destroy",
    "sq_itqfit_create", "sq_itqfit_set_mean", "sq_itqfit_cov", "sq_itqfit_project", "sq_itqfit_iterate",
    "sq_itqfit_destroy",
)


class HipError(RuntimeError):
    """A libsmqtk_hip call failed (or the library could not be loaded)."""


class SqStats(ctypes.Structure):
    _fields_ = [
        ("scan_ms", ctypes.c_double),
        ("total_ms", ctypes.c_double),
        ("scan_launches", ctypes.c_int64),
        ("candidates", ctypes.c_int64),
        ("fallback_queries", ctypes.c_int64),
        ("bytes_scanned", ctypes.c_int64),
        ("rerank_ms", ctypes.c_double),
        ("mid_tier_queries", ctypes.c_int64),
    ]


_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def _declare(lib: ctypes.CDLL) -> None:
    c_int, c_i64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
    lib.sq_last_error.restype = ctypes.c_char_p
    lib.sq_last_error.argtypes = []
    lib.sq_version.restype = c_int
    lib.sq_version.argtypes = []
    lib.sq_device_count.argtypes = [ctypes.POINTER(c_int)]
    lib.sq_device_name.argtypes = [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]
    lib.sq_set_option.argtypes = [ctypes.c_char_p, c_i64]
    lib.sq_handle_set_option.argtypes = [c_i64, ctypes.c_char_p, c_i64]
    lib.sq_handle_reset_options.argtypes = [c_i64]
    lib.sq_get_stats.argtypes = [c_i64, ctypes.POINTER(SqStats)]
    lib.sq_itq_hash.argtypes = [c_vp, c_int, c_i64, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp]
    lib.sq_itq_model_create.argtypes = [c_vp, c_int, c_vp, c_int, c_int, c_int, ctypes.POINTER(c_i64)]
    lib.sq_itq_model_hash.argtypes = [c_i64, c_vp, c_int, c_i64, c_vp, c_int, c_vp]
    lib.sq_itq_model_destroy.argtypes = [c_i64]
    lib.sq_hamming_create.argtypes = [c_vp, c_i64, c_int, c_int, c_i64, ctypes.POINTER(c_i64)]
    lib.sq_hamming_search.argtypes = [c_i64, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp]
    lib.sq_hamming_sync.argtypes = [c_i64]
    lib.sq_hamming_append.argtypes = [c_i64, c_vp, c_i64, c_vp]
    lib.sq_hamming_remove.argtypes = [c_i64, c_vp, c_i64]
    lib.sq_hamming_destroy.argtypes = [c_i64]
    lib.sq_dense_create.argtypes = [c_vp, c_i64, c_int, c_int, c_int, c_i64, ctypes.POINTER(c_i64)]
    lib.sq_dense_create_opts.argtypes = [c_vp, c_i64, c_int, c_int, c_int, c_i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64), c_int,
                                         ctypes.POINTER(c_i64)]
    lib.sq_dense_info.argtypes = [c_i64, ctypes.POINTER(c_i64), c_int]
    lib.sq_dense_append.argtypes = [c_i64, c_vp, c_i64, c_int]
    lib.sq_dense_search.argtypes = [c_i64, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp]
    lib.sq_dense_sync.argtypes = [c_i64]
    lib.sq_dense_destroy.argtypes = [c_i64]
    lib.sq_dense_distances.argtypes = [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_vp, c_int, c_vp]
    lib.sq_merge_topk.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]
    lib.sq_merge_topk_strided.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_vp, c_vp]
    lib.sq_rows_create.argtypes = [c_vp, c_int, c_i64, c_int, c_int, ctypes.POINTER(c_i64)]
    lib.sq_rows_append.argtypes = [c_i64, c_vp, c_i64, c_int]
    lib.sq_rows_rerank.argtypes = [c_i64, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]
    lib.sq_rows_set_buckets.argtypes = [c_i64, c_vp, c_i64, c_vp, c_int]
    lib.sq_lsh_query.argtypes = [c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]
    lib.sq_hamming_info.argtypes = [c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]
    lib.sq_rows_destroy.argtypes = [c_i64]
    lib.sq_itqfit_create.argtypes = [c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, ctypes.POINTER(c_i64)]
    lib.sq_itqfit_set_mean.argtypes = [c_i64, c_vp]
    lib.sq_itqfit_cov.argtypes = [c_i64, c_vp]
    lib.sq_itqfit_project.argtypes = [c_i64, c_vp, c_int]
    lib.sq_itqfit_iterate.argtypes = [c_i64, c_vp, c_vp]
    lib.sq_itqfit_destroy.argtypes = [c_i64]
    for name in EXPORTS:
        if name not in ("sq_last_error",):
            getattr(lib, name).restype = c_int


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``, the same as ROCm's).  If
    this library pulls in ``/opt/rocm/lib/libamdhip64.so.7`` first and torch is
    imported later, the process ends up with two runtimes and the second one
    to initialise finds no GPU.  Loading torch's copy first (without importing
    torch) makes both bind to the same runtime in either import order.  Set
    ``SMQTK_HIP_RUNTIME=system`` to skip this."""
    if os.environ.get("SMQTK_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.isfile(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises HipError when it is missing."""
    global _lib
    with _lock:
        if _lib is None:
            _share_torch_hip_runtime()
            if not os.path.isfile(LIB_PATH):
                raise HipError(
                    f"{LIB_NAME} not found at {LIB_PATH}; build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` or "
                    "`make -C smqtk_indexing_amd/csrc`")
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as ex:
                raise HipError(f"cannot load {LIB_PATH}: {ex}") from ex
            _declare(lib)
            _lib = lib
        return _lib


def _check(rc: int, what: str) -> None:
    if rc != SQ_OK:
        msg = load().sq_last_error()
        raise HipError(f"{what} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


def library_present() -> bool:
    return os.path.isfile(LIB_PATH)


def device_count() -> int:
    """Number of visible HIP devices; 0 when the library or driver is absent."""
    if not library_present():
        return 0
    try:
        lib = load()
    except HipError:
        return 0
    n = ctypes.c_int(0)
    rc = lib.sq_device_count(ctypes.byref(n))
    return int(n.value) if rc == SQ_OK else 0


def usable() -> bool:
    return device_count() > 0


def device_name(device: int = 0) -> Tuple[str, int, int]:
    buf = ctypes.create_string_buffer(256)
    mem = ctypes.c_int64(0)
    cus = ctypes.c_int(0)
    _check(load().sq_device_name(device, buf, 256, ctypes.byref(mem), ctypes.byref(cus)), "sq_device_name")
    return buf.value.decode(), int(mem.value), int(cus.value)


def set_option(name: str, value: int) -> None:
    _check(load().sq_set_option(name.encode(), int(value)), f"sq_set_option({name})")


def get_stats(handle: int) -> dict:
    st = SqStats()
    _check(load().sq_get_stats(handle, ctypes.byref(st)), "sq_get_stats")
    return {k: getattr(st, k) for k, _ in SqStats._fields_}


def _ptr(a) -> ctypes.c_void_p:
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    return ctypes.c_void_p(int(a))


def _host(a: np.ndarray, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


# ----------------------------------------------------------------------- ITQ
def itq_hash(x: np.ndarray, mean: np.ndarray, rotation: np.ndarray, norm_ord: int = SQ_NORM_NONE) -> np.ndarray:
    """Packed codes uint64[n, ceil(bits/64)] of the rows of ``x`` (host arrays)."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("x must be [n, d]")
    if x.dtype == np.float32:
        dt = SQ_DTYPE_F32
    else:
        x = x.astype(np.float64, copy=False)
        dt = SQ_DTYPE_F64
    x = np.ascontiguousarray(x)
    # dtype of the model's mean vector: `x - mean` is evaluated in numpy's promoted dtype (itq.py:404)
    mdt = SQ_DTYPE_F32 if np.asarray(mean).dtype == np.float32 else SQ_DTYPE_F64
    mean = _host(mean, np.float64)
    rotation = _host(rotation, np.float64)
    n, d = x.shape
    if mean.shape != (d,) or rotation.ndim != 2 or rotation.shape[0] != d:
        raise ValueError("mean must be [d] and rotation [d, bits]")
    bits = rotation.shape[1]
    out = np.empty((n, (bits + 63) // 64), dtype=np.uint64)
    if n == 0:
        return out
    _check(load().sq_itq_hash(_ptr(x), dt, n, d, _ptr(mean), mdt, _ptr(rotation), bits, int(norm_ord), _ptr(out),
                              SQ_MEM_HOST, None), "sq_itq_hash")
    return out


def itq_hash_device(x_ptr: int, x_dtype: int, n: int, d: int, mean_ptr: int, rot_ptr: int, bits: int,
                    norm_ord: int, out_ptr: int, stream: int = 0, mean_dtype: int = SQ_DTYPE_F64) -> None:
    """Device pointers throughout; ``mean_ptr``: float64[d] values, ``mean_dtype``: the model's dtype."""
    _check(load().sq_itq_hash(_ptr(x_ptr), x_dtype, n, d, _ptr(mean_ptr), int(mean_dtype), _ptr(rot_ptr), bits,
                              int(norm_ord), _ptr(out_ptr), SQ_MEM_DEVICE, ctypes.c_void_p(stream or None)),
           "sq_itq_hash")


# ------------------------------------------------------------------- handles
class _Handle:
    _destroy_name = ""

    def __init__(self) -> None:
        self.handle = 0
        self._keepalive = None

    def close(self) -> None:
        if self.handle:
            h, self.handle = self.handle, 0
            try:
                getattr(load(), self._destroy_name)(h)
            except Exception:
                pass
            self._keepalive = None

    def __del__(self) -> None:
        self.close()

    def stats(self) -> dict:
        return get_stats(self.handle)

    def set_option(self, name: str, value: int) -> None:
        """An option of THIS handle only (``sq_handle_set_option``): wins over the process-wide :func:`set_option`."""
        _check(load().sq_handle_set_option(self.handle, name.encode(), int(value)), f"sq_handle_set_option({name})")

    def reset_options(self) -> None:
        _check(load().sq_handle_reset_options(self.handle), "sq_handle_reset_options")


class ItqModel(_Handle):
    """ItqFunctor's model resident on the device (``sq_itq_model_*``): repeated small ``hash`` calls move
    only the rows and the codes."""
    _destroy_name = "sq_itq_model_destroy"

    def __init__(self, mean: np.ndarray, rotation: np.ndarray, norm_ord: int = SQ_NORM_NONE):
        super().__init__()
        # dtype of the model's mean vector: `x - mean` is evaluated in numpy's promoted dtype (itq.py:404)
        mdt = SQ_DTYPE_F32 if np.asarray(mean).dtype == np.float32 else SQ_DTYPE_F64
        mean = _host(mean, np.float64)
        rotation = _host(rotation, np.float64)
        if mean.ndim != 1 or rotation.ndim != 2 or rotation.shape[0] != mean.shape[0]:
            raise ValueError("mean must be [d] and rotation [d, bits]")
        self.d, self.bits = int(rotation.shape[0]), int(rotation.shape[1])
        h = ctypes.c_int64(0)
        _check(load().sq_itq_model_create(_ptr(mean), mdt, _ptr(rotation), self.d, self.bits, int(norm_ord),
                                          ctypes.byref(h)), "sq_itq_model_create")
        self.handle = int(h.value)

    def hash(self, x: np.ndarray) -> np.ndarray:
        """Packed codes uint64[n, ceil(bits/64)] of the rows of ``x`` (host array, float32 or float64)."""
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("x must be [n, d]")
        if x.dtype == np.float32:
            dt = SQ_DTYPE_F32
        else:
            x = x.astype(np.float64, copy=False)
            dt = SQ_DTYPE_F64
        x = np.ascontiguousarray(x)
        out = np.empty((x.shape[0], (self.bits + 63) // 64), dtype=np.uint64)
        if x.shape[0]:
            _check(load().sq_itq_model_hash(self.handle, _ptr(x), dt, int(x.shape[0]), _ptr(out), SQ_MEM_HOST, None),
                   "sq_itq_model_hash")
        return out


class HammingIndex(_Handle):
    """Device-resident array of unique packed codes + top-k Hamming search."""
    _destroy_name = "sq_hamming_destroy"

    def __init__(self, codes, n: Optional[int] = None, words: Optional[int] = None, device_ptr: bool = False,
                 id_base: int = 0, keepalive=None):
        super().__init__()
        if device_ptr:
            assert n is not None and words is not None
            ptr, mem = _ptr(codes), SQ_MEM_DEVICE
            self._keepalive = keepalive
        else:
            codes = _host(codes, np.uint64)
            if codes.ndim != 2:
                raise ValueError("codes must be uint64[n, words]")
            n, words = codes.shape
            ptr, mem = _ptr(codes), SQ_MEM_HOST
        self.n, self.words, self.id_base = int(n), int(words), int(id_base)
        h = ctypes.c_int64(0)
        _check(load().sq_hamming_create(ptr, self.n, self.words, mem, self.id_base, ctypes.byref(h)),
               "sq_hamming_create")
        self.handle = int(h.value)

    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _host(queries, np.uint64)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.words:
            raise ValueError("query code width does not match the index")
        nq = q.shape[0]
        dist = np.empty((nq, k), dtype=np.int32)
        idx = np.empty((nq, k), dtype=np.int64)
        _check(load().sq_hamming_search(self.handle, _ptr(q), nq, int(k), _ptr(dist), _ptr(idx), SQ_MEM_HOST, None),
               "sq_hamming_search")
        return dist, idx

    def append(self, new_codes: np.ndarray, insert_pos: np.ndarray) -> None:
        """``sq_hamming_append``: ``new_codes`` ``uint64[m, words]`` ascending and not in the index, ``insert_pos[m]``
        = number of indexed codes smaller than each; row ids become the ranks in the merged order."""
        c = _host(new_codes, np.uint64)
        pos = _host(insert_pos, np.int64).reshape(-1)
        if c.ndim != 2 or c.shape[1] != self.words or pos.shape[0] != c.shape[0]:
            raise ValueError("new_codes must be uint64[m, words] with one insert position each")
        if c.shape[0] == 0:
            return
        _check(load().sq_hamming_append(self.handle, _ptr(c), int(c.shape[0]), _ptr(pos)), "sq_hamming_append")
        self.n += int(c.shape[0])

    def remove(self, ranks: np.ndarray) -> None:
        """``sq_hamming_remove``: strictly ascending current row ids that leave the index."""
        r = _host(ranks, np.int64).reshape(-1)
        if r.shape[0] == 0:
            return
        _check(load().sq_hamming_remove(self.handle, _ptr(r), int(r.shape[0])), "sq_hamming_remove")
        self.n -= int(r.shape[0])

    def search_device(self, q_ptr: int, nq: int, k: int, out_dist_ptr: int, out_idx_ptr: int, stream: int = 0) -> None:
        _check(load().sq_hamming_search(self.handle, _ptr(q_ptr), int(nq), int(k), _ptr(out_dist_ptr),
                                        _ptr(out_idx_ptr), SQ_MEM_DEVICE, ctypes.c_void_p(stream or None)),
               "sq_hamming_search")

    async_option_prefix = "hamming_async"      # (HipSearcher: the names of this index's pipeline options)

    def search_device_async(self, q_ptr: int, nq: int, k: int, out_dist_ptr: int, out_idx_ptr: int, stream: int = 0) -> None:
        """``SQ_MEM_DEVICE_ASYNC``: enqueue and return; final ``depth - 1`` calls later (or after :meth:`sync`)."""
        _check(load().sq_hamming_search(self.handle, _ptr(q_ptr), int(nq), int(k), _ptr(out_dist_ptr),
                                        _ptr(out_idx_ptr), SQ_MEM_DEVICE_ASYNC, ctypes.c_void_p(stream or None)),
               "sq_hamming_search")

    def sync(self) -> None:
        """Finish every asynchronous search in flight (``sq_hamming_sync``)."""
        _check(load().sq_hamming_sync(self.handle), "sq_hamming_sync")


class DenseIndex(_Handle):
    """Device-resident float32 matrix + exact L2 / cosine top-k search."""
    _destroy_name = "sq_dense_destroy"

    def __init__(self, db, n: Optional[int] = None, d: Optional[int] = None, metric: int = SQ_METRIC_L2,
                 device_ptr: bool = False, id_base: int = 0, keepalive=None, options: Optional[dict] = None):
        """``options``: options of THIS index known at create (``sq_dense_create_opts``), e.g. ``{"dense_int8": 0}`` --
        no int8 copy is built or kept; they stay the handle's overrides afterwards."""
        super().__init__()
        if device_ptr:
            assert n is not None and d is not None
            ptr, mem = _ptr(db), SQ_MEM_DEVICE
            self._keepalive = keepalive
        else:
            db = _host(db, np.float32)
            if db.ndim != 2:
                raise ValueError("db must be float32[n, d]")
            n, d = db.shape
            ptr, mem = _ptr(db), SQ_MEM_HOST
        self.n, self.d, self.metric, self.id_base = int(n), int(d), int(metric), int(id_base)
        h = ctypes.c_int64(0)
        if options:
            names = (ctypes.c_char_p * len(options))(*[str(k_).encode() for k_ in options])
            values = (ctypes.c_int64 * len(options))(*[int(v_) for v_ in options.values()])
            _check(load().sq_dense_create_opts(ptr, self.n, self.d, self.metric, mem, self.id_base, names, values, len(options),
                                               ctypes.byref(h)), "sq_dense_create_opts")
        else:
            _check(load().sq_dense_create(ptr, self.n, self.d, self.metric, mem, self.id_base, ctypes.byref(h)),
                   "sq_dense_create")
        self.handle = int(h.value)

    def info(self) -> dict:
        """What the index keeps resident (bytes per copy) and what its build cost (``sq_dense_info``)."""
        out = (ctypes.c_int64 * 10)()
        _check(load().sq_dense_info(self.handle, out, 10), "sq_dense_info")
        return {"rows": int(out[0]), "d": int(out[1]), "f32_rows_bytes": int(out[2]), "f32_rows_owned": bool(out[3]),
                "bf16_copy_bytes": int(out[4]), "int8_copy_bytes": int(out[5]), "row_stats_bytes": int(out[6]),
                "int8_in_use": bool(out[7]), "build_ms": out[8] / 1e3, "build_int8_ms": out[9] / 1e3}

    @property
    def dist_dtype(self):
        return np.float64 if self.metric == SQ_METRIC_COSINE else np.float32

    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _host(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.d:
            raise ValueError("query dimension does not match the index")
        nq = q.shape[0]
        dist = np.empty((nq, k), dtype=self.dist_dtype)
        idx = np.empty((nq, k), dtype=np.int64)
        _check(load().sq_dense_search(self.handle, _ptr(q), nq, int(k), _ptr(dist), _ptr(idx), SQ_MEM_HOST, None),
               "sq_dense_search")
        return dist, idx

    def append(self, rows, n: Optional[int] = None, device_ptr: bool = False) -> None:
        """Append rows (``[m, d]`` float32 host array, or a device pointer + ``n``) to an index created from host
        memory; they get the next row ids.  ``sq_dense_append``."""
        if device_ptr:
            assert n is not None
            ptr, mem, m = _ptr(rows), SQ_MEM_DEVICE, int(n)
        else:
            rows = _host(rows, np.float32)
            if rows.ndim != 2 or rows.shape[1] != self.d:
                raise ValueError("rows must be float32[m, d]")
            ptr, mem, m = _ptr(rows), SQ_MEM_HOST, int(rows.shape[0])
        if m == 0:
            return
        _check(load().sq_dense_append(self.handle, ptr, m, mem), "sq_dense_append")
        self.n += m

    def search_device(self, q_ptr: int, nq: int, k: int, out_dist_ptr: int, out_idx_ptr: int, stream: int = 0) -> None:
        _check(load().sq_dense_search(self.handle, _ptr(q_ptr), int(nq), int(k), _ptr(out_dist_ptr),
                                      _ptr(out_idx_ptr), SQ_MEM_DEVICE, ctypes.c_void_p(stream or None)),
               "sq_dense_search")


    async_option_prefix = "dense_async"

    def search_device_async(self, q_ptr: int, nq: int, k: int, out_dist_ptr: int, out_idx_ptr: int, stream: int = 0) -> None:
        """``SQ_MEM_DEVICE_ASYNC``: enqueue and return.  The results are final when the next call on this index
        (another search, :meth:`sync`, ``append``, ``close``) returns; alternate between two output buffers."""
        _check(load().sq_dense_search(self.handle, _ptr(q_ptr), int(nq), int(k), _ptr(out_dist_ptr),
                                      _ptr(out_idx_ptr), SQ_MEM_DEVICE_ASYNC, ctypes.c_void_p(stream or None)),
               "sq_dense_search")

    def sync(self) -> None:
        """Finish every asynchronous search in flight (``sq_dense_sync``)."""
        _check(load().sq_dense_sync(self.handle), "sq_dense_sync")


def dense_distances(query: np.ndarray, rows: np.ndarray, metric: int = SQ_METRIC_L2) -> np.ndarray:
    """Reference-arithmetic distances from ``query`` to each of ``rows`` (host
    arrays).  Computed in float32 when both operands are float32, otherwise in
    float64 (numpy's promotion in metrics.py:86); cosine is always float64."""
    query, rows = np.asarray(query), np.asarray(rows)
    dt = np.float32 if (query.dtype == np.float32 and rows.dtype == np.float32) else np.float64
    q = _host(query, dt).reshape(-1)
    r = _host(rows, dt)
    if r.ndim != 2 or r.shape[1] != q.shape[0]:
        raise ValueError("rows must be [n, d] with d == len(query)")
    out = np.empty(r.shape[0], dtype=np.float64 if metric == SQ_METRIC_COSINE else dt)
    if r.shape[0] == 0:
        return out
    _check(load().sq_dense_distances(_ptr(q), _ptr(r), SQ_DTYPE_F32 if dt == np.float32 else SQ_DTYPE_F64,
                                     r.shape[0], r.shape[1], int(metric), _ptr(out), SQ_MEM_HOST, None),
           "sq_dense_distances")
    return out


def merge_topk(dist: np.ndarray, idx: np.ndarray, k_out: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host merge of per-shard lists: dist/idx [nshards, nq, k_in] -> [nq, k_out]."""
    idx = _host(idx, np.int64)
    if dist.dtype == np.float32:
        dt = 0
    elif dist.dtype == np.float64:
        dt = 1
    elif dist.dtype == np.int32:
        dt = 2
    else:
        raise ValueError("dist must be float32, float64 or int32")
    dist = np.ascontiguousarray(dist)
    ns, nq, k_in = dist.shape
    od = np.empty((nq, k_out), dtype=dist.dtype)
    oi = np.empty((nq, k_out), dtype=np.int64)
    _check(load().sq_merge_topk(_ptr(dist), _ptr(idx), dt, ns, nq, k_in, int(k_out), _ptr(od), _ptr(oi)),
           "sq_merge_topk")
    return od, oi


def merge_topk_gathered(buf: np.ndarray, nshards: int, nq: int, k_in: int, k_out: int,
                        dist_dtype) -> Tuple[np.ndarray, np.ndarray]:
    """Host merge straight from the receive buffer of ONE all-gather: ``buf`` is a C-contiguous
    uint8 array of ``nshards`` blocks, each ``[ids int64 nq*k_in][dist nq*k_in]`` padded to a multiple of 8 bytes
    (``distributed.packed_block_bytes``)."""
    dist_dtype = np.dtype(dist_dtype)
    dt = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2}[dist_dtype]
    per = (nq * k_in * (8 + dist_dtype.itemsize) + 7) // 8 * 8    # blocks padded to 8 bytes: id blocks stay aligned
    if buf.dtype != np.uint8 or not buf.flags.c_contiguous or buf.size != nshards * per:
        raise ValueError("buf must be a contiguous uint8 array of nshards blocks of nq * k_in * (8 + dist size) "
                         "bytes, each rounded up to a multiple of 8")
    od = np.empty((nq, k_out), dtype=dist_dtype)
    oi = np.empty((nq, k_out), dtype=np.int64)
    base = buf.ctypes.data
    _check(load().sq_merge_topk_strided(base + nq * k_in * 8, base, dt, int(nshards), int(nq), int(k_in), int(k_out), per,
                                        per, _ptr(od), _ptr(oi)), "sq_merge_topk_strided")
    return od, oi


class RowMatrix:
    """Descriptor rows resident on the device for the LSH re-rank stage
    (``sq_rows_*``; lsh.py:499-519).  ``rows``: ``[n, d]`` float32 or float64."""

    def __init__(self, rows, n: Optional[int] = None, d: Optional[int] = None, dtype=None, device_ptr: bool = False,
                 keepalive=None):
        if device_ptr:      # a device matrix [n, d] of `dtype`, borrowed (kept alive by `keepalive`)
            assert n is not None and d is not None and dtype is not None
            self.dtype = np.dtype(dtype)
            self.n, self.d = int(n), int(d)
            ptr, mem = _ptr(rows), SQ_MEM_DEVICE
            self._keepalive = keepalive
        else:
            rows = np.asarray(rows)
            if rows.ndim != 2 or rows.dtype not in (np.float32, np.float64):
                raise ValueError("rows must be a [n, d] float32 or float64 matrix")
            rows = np.ascontiguousarray(rows)
            self.dtype = rows.dtype
            self.n, self.d = int(rows.shape[0]), int(rows.shape[1])
            ptr, mem = _ptr(rows), SQ_MEM_HOST
        h = ctypes.c_int64(0)
        _check(load().sq_rows_create(ptr, SQ_DTYPE_F32 if self.dtype == np.float32 else SQ_DTYPE_F64,
                                     self.n, self.d, mem, ctypes.byref(h)), "sq_rows_create")
        self._h: Optional[int] = h.value

    def set_buckets(self, csr_off, csr_rows, n_codes: Optional[int] = None, device_ptr: bool = False, keepalive=None) -> None:
        """The hash -> rows store as a CSR map (``sq_rows_set_buckets``): ``csr_off[n_codes + 1]``, ``csr_rows[n]``."""
        if device_ptr:
            assert n_codes is not None
            self._csr_keepalive = keepalive
            _check(load().sq_rows_set_buckets(self._h, _ptr(csr_off), int(n_codes), _ptr(csr_rows), SQ_MEM_DEVICE),
                   "sq_rows_set_buckets")
            return
        off = _host(np.asarray(csr_off), np.int64).reshape(-1)
        rows = _host(np.asarray(csr_rows), np.int64).reshape(-1)
        if rows.shape[0] > self.n or off.shape[0] < 2 or int(off[-1]) != rows.shape[0]:
            raise ValueError("csr_rows must list rows of the matrix at most once each, csr_off[-1] of them")
        _check(load().sq_rows_set_buckets(self._h, _ptr(off), int(off.shape[0] - 1), _ptr(rows), SQ_MEM_HOST),
               "sq_rows_set_buckets")

    def lsh_query(self, hamming: "HammingIndex", itq: "ItqModel", queries: np.ndarray, n_codes: int, metric: int,
                  k: int) -> Tuple[np.ndarray, np.ndarray]:
        """``sq_lsh_query``: hash -> nearest ``n_codes`` codes -> bucket expansion -> exact re-rank, all on the device.
        Returns (dist ``[nq, k]``, rows ``[nq, k]``; -1 / +inf padding)."""
        q = _host(np.asarray(queries), self.dtype)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("queries must be [nq, d]")
        nq = q.shape[0]
        k64 = self.dtype == np.float32 and metric == SQ_METRIC_L2
        dist = np.empty((nq, k), dtype=np.float32 if k64 else np.float64)
        rows = np.empty((nq, k), dtype=np.int64)
        _check(load().sq_lsh_query(self._h, hamming.handle, itq.handle, _ptr(q), nq, int(n_codes), int(metric), int(k),
                                   _ptr(dist), _ptr(rows), SQ_MEM_HOST, None), "sq_lsh_query")
        return dist, rows

    def lsh_query_device(self, hamming: "HammingIndex", itq: "ItqModel", q_ptr: int, nq: int, n_codes: int, metric: int,
                         k: int, out_dist_ptr: int, out_rows_ptr: int, stream: int = 0) -> None:
        _check(load().sq_lsh_query(self._h, hamming.handle, itq.handle, _ptr(q_ptr), int(nq), int(n_codes), int(metric),
                                   int(k), _ptr(out_dist_ptr), _ptr(out_rows_ptr), SQ_MEM_DEVICE,
                                   ctypes.c_void_p(stream or None)), "sq_lsh_query")

    def append(self, rows: np.ndarray) -> None:
        """Rows ``[m, d]`` of the matrix's dtype behind the resident ones (``sq_rows_append``)."""
        if self._h is None:
            raise HipError("RowMatrix is closed")
        rows = np.ascontiguousarray(np.asarray(rows), dtype=self.dtype)
        if rows.ndim != 2 or rows.shape[1] != self.d:
            raise ValueError("rows must be [m, d]")
        if rows.shape[0] == 0:
            return
        _check(load().sq_rows_append(self._h, _ptr(rows), int(rows.shape[0]), SQ_MEM_HOST), "sq_rows_append")
        self.n += int(rows.shape[0])

    def rerank(self, queries: np.ndarray, metric: int, cand_rows: np.ndarray, cand_offsets: np.ndarray,
               k: int) -> Tuple[np.ndarray, np.ndarray]:
        """Per query the ``k`` smallest distances among its candidate rows, in
        (distance, position in the candidate list) order.  Returns
        (dist ``[nq, k]`` -- float32 for float32 rows with L2, else float64; +inf
        padding, pos ``[nq, k]`` int64 positions into the query's candidate list; -1 padding)."""
        if self._h is None:
            raise HipError("RowMatrix is closed")
        q = _host(np.asarray(queries), self.dtype)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("queries must be [nq, d]")
        cand = _host(np.asarray(cand_rows), np.int64).reshape(-1)
        off = _host(np.asarray(cand_offsets), np.int64).reshape(-1)
        nq = q.shape[0]
        if off.shape[0] != nq + 1 or off[0] != 0 or off[-1] != cand.shape[0]:
            raise ValueError("cand_offsets must be [nq + 1], start at 0 and end at len(cand_rows)")
        k64 = self.dtype == np.float32 and metric == SQ_METRIC_L2
        dist = np.empty((nq, k), dtype=np.float32 if k64 else np.float64)
        pos = np.empty((nq, k), dtype=np.int64)
        if cand.shape[0] == 0:
            cand = np.zeros(1, dtype=np.int64)
        _check(load().sq_rows_rerank(self._h, _ptr(q), nq, int(metric), _ptr(cand), _ptr(off), int(k), _ptr(dist),
                                     _ptr(pos), None), "sq_rows_rerank")
        return dist, pos

    def close(self) -> None:
        if self._h is not None and _lib is not None:
            _lib.sq_rows_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ItqFit:
    """Device side of ``ItqFunctor.fit`` (``sq_itqfit_*``; itq.py:239-387): the descriptor matrix
    stays on the device, the caller does the small eigen / SVD problems in numpy."""

    def __init__(self, x: np.ndarray, norm_ord: int = SQ_NORM_NONE):
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("x must be [n, d]")
        if x.dtype != np.float32:
            x = x.astype(np.float64, copy=False)
        x = np.ascontiguousarray(x)
        self.n, self.d = int(x.shape[0]), int(x.shape[1])
        self.mean = np.empty(self.d, dtype=np.float64)
        h = ctypes.c_int64(0)
        _check(load().sq_itqfit_create(_ptr(x), SQ_DTYPE_F32 if x.dtype == np.float32 else SQ_DTYPE_F64, self.n, self.d,
                                       int(norm_ord), SQ_MEM_HOST, _ptr(self.mean), ctypes.byref(h)), "sq_itqfit_create")
        self._h: Optional[int] = h.value
        self.bits = 0

    def set_mean(self, mean: np.ndarray) -> None:
        m = _host(np.asarray(mean), np.float64)
        _check(load().sq_itqfit_set_mean(self._h, _ptr(m)), "sq_itqfit_set_mean")

    def cov(self) -> np.ndarray:
        out = np.empty((self.d, self.d), dtype=np.float64)
        _check(load().sq_itqfit_cov(self._h, _ptr(out)), "sq_itqfit_cov")
        return out

    def project(self, pc: np.ndarray) -> None:
        pc = _host(np.asarray(pc), np.float64)
        if pc.ndim != 2 or pc.shape[0] != self.d:
            raise ValueError("pc must be [d, bits]")
        self.bits = int(pc.shape[1])
        _check(load().sq_itqfit_project(self._h, _ptr(pc), self.bits), "sq_itqfit_project")

    def iterate(self, r: np.ndarray) -> np.ndarray:
        r = _host(np.asarray(r), np.float64)
        out = np.empty((self.bits, self.bits), dtype=np.float64)
        _check(load().sq_itqfit_iterate(self._h, _ptr(r), _ptr(out)), "sq_itqfit_iterate")
        return out

    def close(self) -> None:
        if self._h is not None and _lib is not None:
            _lib.sq_itqfit_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
