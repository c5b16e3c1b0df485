"""ctypes binding of libdoppel_amd.so (C ABI: include/doppel_amd.h).  Fails loudly when the library is missing."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SOURCES = ("ds_runtime.hip", "ds_jaccard.hip", "ds_features.hip", "ds_build.hip", "ds_forest.hip")
_lib = None


class DoppelError(Exception):
    """An error reported by libdoppel_amd.so (message from ds_last_error())."""


def library_path():
    # DS_LIBRARY selects another build of the same sources (tuning sweeps: scripts/sweep_variants.sh)
    return os.environ.get("DS_LIBRARY") or os.path.join(_HERE, "libdoppel_amd.so")


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libdoppel_amd.so next to this file (hipcc cross-compiles without a GPU)."""
    sources = [os.path.join(_HERE, "csrc", name) for name in _SOURCES]
    headers = [os.path.join(_HERE, "csrc", "ds_common.h"), os.path.join(_ROOT, "include", "doppel_amd.h")]
    target = library_path()
    if not force and os.path.exists(target):
        newest = max(os.path.getmtime(path) for path in sources + headers)
        if os.path.getmtime(target) >= newest:
            return target
    command = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-I", os.path.join(_ROOT, "include"), "-o", target] + os.environ.get("DS_BUILD_FLAGS", "").split() + sources
    if verbose:
        print(" ".join(command))
    subprocess.check_call(command)
    return target


def _declare(handle):
    c = ctypes
    p = c.c_void_p
    handle.ds_last_error.restype = c.c_char_p
    handle.ds_last_error.argtypes = []
    signatures = {
        "ds_version": [],
        "ds_device_count": [c.POINTER(c.c_int)],
        "ds_device_name": [c.c_int, c.c_char_p, c.c_size_t],
        "ds_index_create": [p, p, p, p, c.c_int64, c.c_int64, c.c_int, c.POINTER(p)],
        "ds_index_info": [p, c.POINTER(c.c_int64)],
        "ds_jaccard_topk": [p, p, p, p, c.c_int64, c.c_int32, p],
        "ds_jaccard_topk_device": [p, p, p, p, c.c_int64, c.c_int32, p, p],
        "ds_jaccard_sync": [p, p, c.POINTER(c.c_int64)],
        "ds_construct_features": [p, p, p, p, p, c.c_uint8, c.c_uint32, c.c_int64, c.c_int64, c.c_int, p],
        "ds_titles_create": [p, c.c_int64, p, p, c.c_int64, c.c_int, c.POINTER(p)],
        "ds_construct_features_indexed": [p, p, p, p, c.c_uint8, c.c_uint32, c.c_int64, p],
        "ds_construct_features_indexed_device": [p, p, p, p, c.c_int64, c.c_int32, c.c_uint8, c.c_uint32, c.c_int64,
                                                 p, p],
        "ds_levenshtein_ratio_batch": [p, p, p, p, c.c_int64, c.c_int, c.c_int, p],
        "ds_close_matches": [p, p, p, c.c_int32, c.c_int64, c.c_uint8, p, c.c_int32, p, p],
        "ds_close_matches_device": [p, p, p, c.c_int64, c.c_int32, c.c_int64, c.c_uint8, p, c.c_int32, p, p, p],
        "ds_problem_create": [p, p, c.c_int64, p, p, c.c_int64, c.c_int32, c.POINTER(p)],
        "ds_problem_info": [p, c.POINTER(c.c_int64)],
        "ds_problem_arrays": [p] + [c.POINTER(p)] * 9,
        "ds_transform_titles": [p, p, c.c_int64, c.c_int32, c.c_int32, p, p],
        "ds_forest_create": [p, p, p, p, p, p, c.c_int32, c.c_int32, c.c_float, c.c_int, c.POINTER(p)],
        "ds_forest_predict": [p, p, c.c_int64, p, p],
        "ds_forest_predict_device": [p, p, c.c_int64, p, p, p],
        "ds_malloc": [c.POINTER(p), c.c_size_t, c.c_int],
        "ds_free": [p, c.c_int],
        "ds_memcpy_h2d": [p, p, c.c_size_t, c.c_int],
        "ds_memcpy_d2h": [p, p, c.c_size_t, c.c_int],
        "ds_memset": [p, c.c_int, c.c_size_t, c.c_int],
        "ds_stream_sync": [p, c.c_int],
        "ds_timer_create": [c.c_int, c.POINTER(p)],
        "ds_timer_start": [p, p],
        "ds_timer_stop": [p, p],
        "ds_timer_elapsed_ms": [p, c.POINTER(c.c_float)],
    }
    for name, argtypes in signatures.items():
        function = getattr(handle, name)
        function.argtypes = argtypes
        function.restype = c.c_int
    for name in ("ds_index_destroy", "ds_titles_destroy", "ds_timer_destroy", "ds_problem_destroy", "ds_forest_destroy"):
        function = getattr(handle, name)
        function.argtypes = [p]
        function.restype = None
    return handle


EXPORTED_SYMBOLS = (
    "ds_last_error", "ds_version", "ds_device_count", "ds_device_name", "ds_index_create", "ds_index_destroy",
    "ds_index_info", "ds_jaccard_topk", "ds_jaccard_topk_device", "ds_jaccard_sync", "ds_construct_features",
    "ds_titles_create", "ds_titles_destroy", "ds_construct_features_indexed", "ds_construct_features_indexed_device",
    "ds_levenshtein_ratio_batch", "ds_close_matches", "ds_close_matches_device", "ds_problem_create",
    "ds_problem_destroy", "ds_problem_info", "ds_problem_arrays", "ds_transform_titles", "ds_forest_create", "ds_forest_destroy",
    "ds_forest_predict", "ds_forest_predict_device", "ds_malloc", "ds_free", "ds_memcpy_h2d", "ds_memcpy_d2h", "ds_memset",
    "ds_stream_sync", "ds_timer_create", "ds_timer_destroy", "ds_timer_start", "ds_timer_stop",
    "ds_timer_elapsed_ms")


def lib():
    """The loaded library.  Raises DoppelError (never falls back to a CPU path) when it has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise DoppelError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  doppel-speller_amd has no CPU fallback.")
        _lib = _declare(ctypes.CDLL(path))
    return _lib


def check(status, what=""):
    if status != 0:
        message = lib().ds_last_error().decode("utf-8", "replace")
        if status == -3:  # DS_E_TOP_N: same exception text as match_maker.py:189
            raise Exception("top_matches.shape[0] != self.top_n")
        raise DoppelError(f"{what}: {message} (status {status})")


def device_count():
    count = ctypes.c_int(0)
    status = lib().ds_device_count(ctypes.byref(count))
    return int(count.value) if status == 0 else 0


def pointer(array):
    return ctypes.c_void_p(array.ctypes.data)


class DeviceArray:
    """A typed buffer in HBM owned through ds_malloc/ds_free (plumbing for tests and bench.py)."""

    def __init__(self, shape, dtype, device=0):
        import numpy as np
        self.shape = tuple(shape) if hasattr(shape, "__len__") else (int(shape),)
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        raw = ctypes.c_void_p()
        check(lib().ds_malloc(ctypes.byref(raw), self.nbytes, device), "ds_malloc")
        self.ptr = raw

    @classmethod
    def from_host(cls, array, device=0):
        import numpy as np
        array = np.ascontiguousarray(array)
        out = cls(array.shape, array.dtype, device)
        check(lib().ds_memcpy_h2d(out.ptr, pointer(array), array.nbytes, device), "ds_memcpy_h2d")
        return out

    def to_host(self):
        import numpy as np
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().ds_memcpy_d2h(pointer(out), self.ptr, self.nbytes, self.device), "ds_memcpy_d2h")
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().ds_free(self.ptr, self.device)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Timer:
    """HIP events recorded on a given stream."""

    def __init__(self, device=0):
        self.handle = ctypes.c_void_p()
        check(lib().ds_timer_create(device, ctypes.byref(self.handle)), "ds_timer_create")

    def start(self, stream=None):
        check(lib().ds_timer_start(self.handle, ctypes.c_void_p(stream or 0)), "ds_timer_start")

    def stop(self, stream=None):
        check(lib().ds_timer_stop(self.handle, ctypes.c_void_p(stream or 0)), "ds_timer_stop")

    def elapsed_ms(self):
        ms = ctypes.c_float(0)
        check(lib().ds_timer_elapsed_ms(self.handle, ctypes.byref(ms)), "ds_timer_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                lib().ds_timer_destroy(self.handle)
        except Exception:
            pass
