"""ctypes binding of libdoppel_amd.so (C ABI: include/doppel_amd.h).  Fails loudly when the library is missing."""
import ctypes
import hashlib
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SOURCES = ("ds_runtime.hip", "ds_jaccard.hip", "ds_jaccard_wide.hip", "ds_jaccard_narrow.hip", "ds_features.hip",
            "ds_build.hip", "ds_forest.hip", "ds_pairs.hip")
_lib = None


class DoppelError(Exception):
    """An error reported by libdoppel_amd.so (message from ds_last_error())."""


def library_path():
    # DS_LIBRARY selects another build of the same sources (tuning sweeps: scripts/sweep_variants.sh)
    return os.environ.get("DS_LIBRARY") or os.path.join(_HERE, "libdoppel_amd.so")


def source_id():
    """First 16 hex digits of the SHA-256 over csrc/* and include/* (name + contents, names ascending): the identity
    baked into the library as ds_build_id().  None when the sources are not next to the package."""
    files = []
    for directory in (os.path.join(_HERE, "csrc"), os.path.join(_ROOT, "include")):
        if not os.path.isdir(directory):
            return None
        files += [os.path.join(directory, name) for name in sorted(os.listdir(directory))
                  if name.endswith((".hip", ".h", ".inc"))]
    digest = hashlib.sha256()
    for path in files:
        digest.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as handle:
            digest.update(handle.read())
        digest.update(b"\0")
    return digest.hexdigest()[:16]


def binary_id(path):
    """The id baked into an existing library, read from its bytes (no dlopen); None when absent."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as handle:
        found = re.search(rb"DS_BUILD_ID=([0-9a-f]{16}|unidentified)\0", handle.read())
    return found.group(1).decode() if found else None


# Diagnostic builds of the same sources, next to the product library (selected with DS_LIBRARY=<path> by the tests that
# use them): "boundscheck" checks every data-dependent global index of the fast Jaccard kernel (ds_jaccard_sync
# stats[28..30]).
VARIANTS = {"boundscheck": ["-DDS_BOUNDS_CHECK"]}


def variant_path(variant):
    return os.path.join(_HERE, f"libdoppel_amd_{variant}.so")


def build_library(force=False, verbose=False, variant=None):
    """Compile csrc/*.hip for gfx950 into libdoppel_amd.so next to this file (hipcc cross-compiles without a GPU).
    Rebuilds whenever the id baked into the existing binary differs from the sources' (not on mtimes).
    variant: one of VARIANTS -> libdoppel_amd_<variant>.so with the variant's extra flags."""
    sources = [os.path.join(_HERE, "csrc", name) for name in _SOURCES]
    target = library_path() if variant is None else variant_path(variant)
    extra = [] if variant is None else VARIANTS[variant]
    wanted = source_id()
    if not force and binary_id(target) == wanted:
        return target
    # One builder at a time (the ranks of a multi-GPU job may all find the same stale library at once): the others wait
    # for the lock, find the finished binary and return.  The link step writes a scratch file that replaces the target
    # atomically, so a process that is loading the old library never sees a half-written one.
    import fcntl
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and binary_id(target) == wanted:   # somebody else built it while this process waited
            return target
        _compile_and_link(sources, target, wanted, verbose, extra)
    return target


def _compile_and_link(sources, target, wanted, verbose, extra=()):
    # one hipcc per translation unit, side by side (the two geometries of the Jaccard kernels take a minute each), then
    # one link step
    import concurrent.futures
    import tempfile
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f'-DDS_BUILD_ID="{wanted}"', "-I",
             os.path.join(_ROOT, "include")] + list(extra) + os.environ.get("DS_BUILD_FLAGS", "").split()
    with tempfile.TemporaryDirectory(prefix="ds_build_") as scratch:
        objects = [os.path.join(scratch, os.path.basename(source) + ".o") for source in sources]

        def compile_one(pair):
            source, output = pair
            command = ["hipcc"] + flags + ["-c", source, "-o", output]
            if verbose:
                print(" ".join(command), flush=True)
            subprocess.check_call(command)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(sources))) as pool:
            list(pool.map(compile_one, zip(sources, objects)))
        linked = f"{target}.{os.getpid()}.tmp"
        # -Bsymbolic-functions: calls between the library's own entry points bind inside the library, whatever else a process
        # maps that exports the same names (the tests' CPU twin of this ABI does, on purpose)
        link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic-functions", "-o", linked] + objects
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.check_call(link)
        os.replace(linked, target)


def _declare(handle):
    c = ctypes
    p = c.c_void_p
    handle.ds_last_error.restype = c.c_char_p
    handle.ds_last_error.argtypes = []
    handle.ds_build_id.restype = c.c_char_p
    handle.ds_build_id.argtypes = []
    signatures = {
        "ds_version": [],
        "ds_device_count": [c.POINTER(c.c_int)],
        "ds_device_name": [c.c_int, c.c_char_p, c.c_size_t],
        "ds_index_create": [p, p, p, p, c.c_int64, c.c_int64, c.c_int, c.POINTER(p)],
        "ds_index_info": [p, c.POINTER(c.c_int64)],
        "ds_index_duplicate_ranks": [p, p, p, c.c_int64, c.c_int64, p],
        "ds_index_image_digest": [p, p, p, p, c.c_int64, c.c_int64, c.c_int64, p],
        "ds_index_option": [p, c.c_char_p, c.c_int64],
        "ds_jaccard_topk": [p, p, p, p, c.c_int64, c.c_int32, p],
        "ds_jaccard_topk_device": [p, p, p, p, c.c_int64, c.c_int32, p, p],
        "ds_jaccard_sync": [p, p, c.POINTER(c.c_int64)],
        "ds_jaccard_status": [p, p, p, c.c_int64],
        "ds_construct_features": [p, p, p, p, p, c.c_uint8, c.c_uint32, c.c_int64, c.c_int64, c.c_int, p],
        "ds_titles_create": [p, c.c_int64, p, p, c.c_int64, c.c_int, c.POINTER(p)],
        "ds_titles_option": [p, c.c_char_p, c.c_int64],
        "ds_construct_features_indexed": [p, p, p, p, c.c_uint8, c.c_uint32, c.c_int64, p],
        "ds_construct_features_indexed_device": [p, p, p, p, c.c_int64, c.c_int32, c.c_uint8, c.c_uint32, c.c_int64,
                                                 p, p],
        "ds_levenshtein_ratio_batch": [p, p, p, p, c.c_int64, c.c_int, c.c_int, p],
        "ds_levenshtein_ratio": [p, c.c_int, p, c.c_int],
        "ds_close_matches": [p, p, p, c.c_int32, c.c_int64, c.c_uint8, p, c.c_int32, p, p],
        "ds_close_matches_device": [p, p, p, c.c_int64, c.c_int32, c.c_int64, c.c_uint8, p, c.c_int32, p, p, p],
        "ds_remaining_pairs_device": [p, p, c.c_int64, c.c_int32, c.c_int64, p, p, p, p],
        "ds_select_matches_device": [p, p, p, c.c_int64, c.c_int32, c.c_float, p, p, p],
        "ds_problem_create": [p, p, c.c_int64, p, p, c.c_int64, c.c_int32, c.POINTER(p)],
        "ds_problem_info": [p, c.POINTER(c.c_int64)],
        "ds_problem_arrays": [p] + [c.POINTER(p)] * 9,
        "ds_transform_titles": [p, p, c.c_int64, c.c_int32, c.c_int32, p, p],
        "ds_encode_titles": [p, p, c.c_int64, p, c.c_int64, p, p],
        "ds_truth_word_counts": [p, p, c.c_int64, p, p],
        "ds_forest_create": [p, p, p, p, p, p, c.c_int32, c.c_int32, c.c_float, c.c_int, c.POINTER(p)],
        "ds_forest_predict": [p, p, c.c_int64, p, p],
        "ds_forest_predict_device": [p, p, c.c_int64, p, p, p],
        "ds_malloc": [c.POINTER(p), c.c_size_t, c.c_int],
        "ds_free": [p, c.c_int],
        "ds_memcpy_h2d": [p, p, c.c_size_t, c.c_int],
        "ds_memcpy_d2h": [p, p, c.c_size_t, c.c_int],
        "ds_memset": [p, c.c_int, c.c_size_t, c.c_int],
        "ds_stream_sync": [p, c.c_int],
        "ds_memcpy_d2d_async": [p, p, c.c_size_t, c.c_int, p],
        "ds_stream_create": [c.c_int, c.POINTER(p)],
        "ds_stream_destroy": [p, c.c_int],
        "ds_timer_create": [c.c_int, c.POINTER(p)],
        "ds_timer_start": [p, p],
        "ds_timer_stop": [p, p],
        "ds_timer_elapsed_ms": [p, c.POINTER(c.c_float)],
    }
    for name, argtypes in signatures.items():
        if not hasattr(handle, name) and os.environ.get("DS_ALLOW_STALE_LIBRARY") == "1":
            continue  # an older library named by DS_LIBRARY for an A/B measurement: entry points added since are simply absent
        function = getattr(handle, name)
        function.argtypes = argtypes
        function.restype = c.c_int
    handle.ds_remaining_pairs_counts_size.argtypes = [c.c_int64]
    handle.ds_remaining_pairs_counts_size.restype = c.c_int64
    for name in ("ds_index_destroy", "ds_titles_destroy", "ds_timer_destroy", "ds_problem_destroy", "ds_forest_destroy"):
        function = getattr(handle, name)
        function.argtypes = [p]
        function.restype = None
    return handle


EXPORTED_SYMBOLS = (
    "ds_last_error", "ds_version", "ds_build_id", "ds_device_count", "ds_device_name", "ds_index_create", "ds_index_destroy",
    "ds_index_duplicate_ranks", "ds_index_image_digest", "ds_index_option",
    "ds_index_info", "ds_jaccard_topk", "ds_jaccard_topk_device", "ds_jaccard_sync", "ds_jaccard_status", "ds_construct_features",
    "ds_titles_create", "ds_titles_destroy", "ds_titles_option", "ds_construct_features_indexed", "ds_construct_features_indexed_device",
    "ds_levenshtein_ratio_batch", "ds_levenshtein_ratio", "ds_close_matches", "ds_close_matches_device", "ds_remaining_pairs_counts_size", "ds_remaining_pairs_device",
    "ds_select_matches_device", "ds_problem_create",
    "ds_problem_destroy", "ds_problem_info", "ds_problem_arrays", "ds_transform_titles", "ds_encode_titles", "ds_truth_word_counts", "ds_forest_create", "ds_forest_destroy",
    "ds_forest_predict", "ds_forest_predict_device", "ds_malloc", "ds_free", "ds_memcpy_h2d", "ds_memcpy_d2h", "ds_memset",
    "ds_stream_sync", "ds_memcpy_d2d_async", "ds_stream_create", "ds_stream_destroy", "ds_timer_create", "ds_timer_destroy", "ds_timer_start", "ds_timer_stop",
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
        sources = source_id()
        if sources is not None and binary_id(path) != sources and os.environ.get("DS_ALLOW_STALE_LIBRARY") != "1":
            # never run a binary that does not match the sources next to it: rebuild it where hipcc exists (same
            # image on the GPU box), refuse it otherwise
            import shutil
            import sys
            if os.environ.get("DS_AUTO_REBUILD", "1") != "0" and shutil.which("hipcc") and "DS_LIBRARY" not in os.environ:
                print(f"doppel-speller_amd: {path} was built from other sources ({binary_id(path)} != {sources}); "
                      "rebuilding", file=sys.stderr, flush=True)
                build_library()
            else:
                raise DoppelError(
                    f"{path} was built from sources {binary_id(path)}, but csrc/ + include/ are now {sources}: rebuild "
                    "it (`python -c 'import __graft_entry__ as g; g.build()'`); a stale library is refused, not used.")
        handle = _declare(ctypes.CDLL(path))
        if sources is not None and handle.ds_build_id().decode() != sources and \
                os.environ.get("DS_ALLOW_STALE_LIBRARY") != "1":
            raise DoppelError(f"{path}: ds_build_id() = {handle.ds_build_id().decode()} does not match the sources "
                              f"({sources})")
        _lib = handle
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
    def view(cls, pointer, shape, dtype, device=0):
        """A typed view of device memory owned by somebody else (never freed here)."""
        import numpy as np
        out = cls.__new__(cls)
        out.shape = tuple(shape)
        out.dtype = np.dtype(dtype)
        out.device = device
        out.nbytes = int(np.prod(out.shape, dtype=np.int64)) * out.dtype.itemsize
        out.ptr = pointer if isinstance(pointer, ctypes.c_void_p) else ctypes.c_void_p(int(pointer))
        out.owned = False
        return out

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
        if getattr(self, "owned", True) and self.ptr is not None and self.ptr.value:
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
