"""Query sharding across the GPUs of one node (one process per GPU) and the single gather of the top-k.

Queries are independent (match_maker.py:192-203 is evaluated per row), so rank r of W owns the contiguous query range
[Q*r/W, Q*(r+1)/W) and the truth index, title tables and word counts are replicated on every GPU.  The only
collective on the data path is one all-gather of the int32[Q/W, k] row indexes over RCCL (xGMI).

No PyTorch anywhere: `Rendezvous` is a small TCP star through rank 0 (unique-id broadcast, host barrier, max of a
float: what `bench.py` needs around the timed region) and `RcclCommunicator` binds librccl.so with ctypes
(`ncclGetUniqueId`, `ncclCommInitRank`, `ncclAllGather` on the caller's HIP stream).  `HostCommunicator` has the same
`all_gather` over host arrays through the rendezvous: the communicator the CPU tests inject.
"""
import ctypes
import os
import socket
import struct
import time

import numpy as np

from . import _lib


def shard_range(n_queries, rank, world_size):
    """Contiguous query range [begin, end) of `rank`."""
    begin = (n_queries * rank) // world_size
    end = (n_queries * (rank + 1)) // world_size
    return begin, end


def shard_sizes(n_queries, world_size):
    return [shard_range(n_queries, r, world_size)[1] - shard_range(n_queries, r, world_size)[0]
            for r in range(world_size)]


def slice_queries(q_rowptr, q_cols, q_maxint, begin, end):
    """The CSR slice of queries [begin, end) (host arrays)."""
    q_rowptr = np.asarray(q_rowptr)
    first, last = int(q_rowptr[begin]), int(q_rowptr[end])
    return (q_rowptr[begin:end + 1] - first).astype(np.int64), np.asarray(q_cols)[first:last], \
        np.asarray(q_maxint)[begin:end]


# ---- rendezvous: TCP star through rank 0 ------------------------------------------------------------------------------
def _send(sock, payload):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


def _receive(sock):
    def exactly(count):
        chunks = []
        while count:
            chunk = sock.recv(min(count, 1 << 20))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            chunks.append(chunk)
            count -= len(chunk)
        return b"".join(chunks)
    (length,) = struct.unpack("<q", exactly(8))
    return exactly(length)


def private_directory(base=None):
    """<base or tmp>/ds_<uid>: created 0700 and verified to be a real directory of this user that nobody else can write
    to (never a planted link, never somebody else's directory)."""
    import stat
    import tempfile
    path = os.path.join(base or tempfile.gettempdir(), f"ds_{os.getuid()}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    info = os.lstat(path)
    if not stat.S_ISDIR(info.st_mode) or info.st_uid != os.getuid():
        raise _lib.DoppelError(f"{path} is not a directory owned by this user")
    if info.st_mode & 0o077:
        # somebody else could write here before this call: whatever they planted would stay -- refuse instead of tightening it
        raise _lib.DoppelError(f"{path} exists with mode {stat.S_IMODE(info.st_mode):o}: group / world access; remove it or "
                               "pass another base directory")
    return path


def _write_private(path, text):
    """Create `path` exclusively (no link is followed, nothing is overwritten in place) and publish it atomically."""
    scratch = f"{path}.{os.getpid()}.tmp"
    if os.path.lexists(scratch):
        os.unlink(scratch)
    descriptor = os.open(scratch, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(descriptor, "w") as handle:
        handle.write(text)
    os.replace(scratch, path)


class Rendezvous:
    """World-wide host-side exchange for one process per GPU: rank 0 listens on (address, port), every other rank
    connects.  All operations are collective and must be called by every rank in the same order."""

    _MAGIC = b"DSRV1"

    def __init__(self, rank, world_size, address="127.0.0.1", port=29533, timeout=600.0, port_file=None,
                 hello_timeout=5.0):
        """port: where rank 0 listens.  With `port_file`, rank 0 falls back to any free port when `port` is taken and
        publishes the port it got in that file (written atomically); the other ranks read it before every attempt."""
        self.rank, self.world_size = rank, world_size
        self.peers = {}
        self.server = None
        if world_size == 1:
            return
        if rank == 0:
            if port_file and os.path.lexists(port_file):
                os.unlink(port_file)  # a previous job's
            self.server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self.server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                self.server.bind((address, port))
            except OSError:
                if not port_file:
                    raise
                self.server.bind((address, 0))
            if port_file:
                _write_private(port_file, str(self.server.getsockname()[1]))
            self.server.listen(world_size)
            self.server.settimeout(timeout)
            while len(self.peers) < world_size - 1:
                connection, _ = self.server.accept()
                connection.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                connection.settimeout(hello_timeout)  # a stray connection that says nothing costs seconds, not `timeout`
                try:
                    hello = _receive(connection)
                    connection.settimeout(timeout)
                except (ConnectionError, OSError, struct.error):
                    connection.close()
                    continue
                if len(hello) != len(self._MAGIC) + 4 or not hello.startswith(self._MAGIC):
                    connection.close()  # not one of ours
                    continue
                _send(connection, self._MAGIC)
                self.peers[struct.unpack("<i", hello[len(self._MAGIC):])[0]] = connection
        else:
            deadline = time.time() + timeout
            while True:
                connection = None
                try:
                    target = port
                    if port_file and os.path.exists(port_file):
                        with open(port_file) as handle:
                            target = int(handle.read().strip() or port)
                    connection = socket.create_connection((address, target), timeout=5.0)
                    connection.settimeout(10.0)
                    _send(connection, self._MAGIC + struct.pack("<i", rank))
                    if _receive(connection) == self._MAGIC:
                        break
                    connection.close()
                except (OSError, ValueError, ConnectionError, struct.error):
                    if connection is not None:
                        connection.close()
                if time.time() > deadline:
                    raise TimeoutError(f"rank {rank}: no rendezvous with rank 0 at {address}:{port}")
                time.sleep(0.05)
            connection.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            connection.settimeout(timeout)
            self.peers[0] = connection

    @classmethod
    def from_environment(cls, port_offset=1):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as `torch.distributed.run` (or bench.py's own launcher) export
        them.  The launcher's store owns MASTER_PORT itself, so the star listens `port_offset` above it -- or, when
        that port is taken, wherever rank 0 finds room: the port is published in a file named after MASTER_PORT."""
        master_port = int(os.environ.get("MASTER_PORT", "29533"))
        port_file = os.path.join(private_directory(), f"rendezvous_{master_port}.port")
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), master_port + port_offset, port_file=port_file)

    def all_gather_bytes(self, payload):
        """Every rank's payload, in rank order, on every rank."""
        if self.world_size == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_receive(self.peers[r]) for r in range(1, self.world_size)]
            packed = b"".join(struct.pack("<q", len(p)) + p for p in parts)
            for r in range(1, self.world_size):
                _send(self.peers[r], packed)
            return parts
        _send(self.peers[0], payload)
        packed, parts, at = _receive(self.peers[0]), [], 0
        for _ in range(self.world_size):
            (length,) = struct.unpack_from("<q", packed, at)
            parts.append(packed[at + 8:at + 8 + length])
            at += 8 + length
        return parts

    def broadcast_bytes(self, payload=None):
        """Rank 0's payload on every rank."""
        return self.all_gather_bytes(payload if self.rank == 0 else b"")[0]

    def barrier(self):
        self.all_gather_bytes(b"")

    def max(self, value):
        return max(struct.unpack("<d", part)[0] for part in self.all_gather_bytes(struct.pack("<d", float(value))))

    def close(self):
        for connection in self.peers.values():
            connection.close()
        if self.server is not None:
            self.server.close()
        self.peers, self.server = {}, None


# ---- communicators ----------------------------------------------------------------------------------------------------
class HostCommunicator:
    """`all_gather` over host arrays through the rendezvous (CPU tests; same interface as RcclCommunicator)."""
    on_device = False

    def __init__(self, rendezvous):
        self.rendezvous = rendezvous
        self.rank, self.world_size = rendezvous.rank, rendezvous.world_size

    def all_gather(self, send, stream=None):
        """send: numpy array, same shape on every rank -> array [world_size * send.shape[0], ...] in rank order."""
        send = np.ascontiguousarray(send)
        parts = self.rendezvous.all_gather_bytes(send.tobytes())
        return np.concatenate([np.frombuffer(p, dtype=send.dtype).reshape(send.shape) for p in parts])

    def close(self):
        pass


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]


_NCCL_TYPES = {np.dtype(np.int8): 0, np.dtype(np.uint8): 1, np.dtype(np.int32): 2, np.dtype(np.uint32): 3,
               np.dtype(np.int64): 4, np.dtype(np.uint64): 5, np.dtype(np.float32): 7, np.dtype(np.float64): 8}


def _load_rccl():
    override = os.environ.get("DS_RCCL_LIBRARY")   # when set, ONLY this file is tried
    for name in ((override,) if override else ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                                               "/opt/rocm/lib/librccl.so")):
        try:
            handle = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
        handle.ncclGetErrorString.restype = ctypes.c_char_p
        handle.ncclGetErrorString.argtypes = [ctypes.c_int]
        handle.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
        handle.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        handle.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p]
        handle.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        return handle
    raise _lib.DoppelError(f"{override}: cannot be loaded (DS_RCCL_LIBRARY)" if override else
                           "librccl.so not found (set DS_RCCL_LIBRARY)")


class _StdoutToStderr:
    """RCCL prints a version banner on the PROCESS's stdout when a communicator is created; a caller whose stdout is
    a protocol (bench.py: one JSON line) must not see it.  File descriptor 1 points at stderr while this is active."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class RcclCommunicator:
    """One RCCL communicator over the ranks of the rendezvous (one process per GPU, device = LOCAL_RANK).

    The unique id is created by rank 0 and broadcast through the rendezvous; `all_gather` enqueues ONE ncclAllGather
    on the caller's HIP stream (device pointers in, device pointers out, no host staging)."""
    on_device = True

    def __init__(self, rendezvous, device, select_device=True):
        """Collective over the rendezvous.  Every rank takes part in TWO host exchanges before ncclCommInitRank -- the
        id broadcast and one readiness flag per rank -- whatever happened to it locally: a rank that failed (no
        librccl, a LOCAL_RANK beyond the visible devices ...) reports its error text, and every rank raises.  Nobody
        enters the collective ncclCommInitRank(world_size) with a rank missing (it would block forever)."""
        self.rendezvous = rendezvous
        self.rank, self.world_size, self.device = rendezvous.rank, rendezvous.world_size, device
        unique, failure = _NcclUniqueId(), None
        try:
            self.rccl = _load_rccl()
            if select_device:
                _lib.check(_lib.lib().ds_stream_sync(None, device), "select device")  # hipSetDevice(device) for RCCL
            if self.rank == 0:
                with _StdoutToStderr():
                    status = self.rccl.ncclGetUniqueId(ctypes.byref(unique))
                self._check(status, "ncclGetUniqueId")
        except Exception as error:  # noqa: BLE001 - reported to every rank below
            failure = error
        raw = rendezvous.broadcast_bytes(
            (b"" if failure else ctypes.string_at(ctypes.byref(unique), 128)) if self.rank == 0 else None)
        if failure is None and len(raw) != 128:
            failure = _lib.DoppelError("rank 0 could not create an RCCL unique id" if not raw
                                       else "rendezvous delivered a malformed RCCL unique id")
        flags = rendezvous.all_gather_bytes(b"" if failure is None else f"rank {self.rank}: {failure}".encode()[:400])
        if failure is not None:
            raise failure
        others = [flag.decode("utf-8", "replace") for flag in flags if flag]
        if others:
            raise _lib.DoppelError("RCCL communicator not created, another rank failed: " + "; ".join(others))
        ctypes.memmove(ctypes.byref(unique), raw, 128)
        self.comm = ctypes.c_void_p()
        with _StdoutToStderr():
            status = self.rccl.ncclCommInitRank(ctypes.byref(self.comm), self.world_size, unique, self.rank)
        self._check(status, "ncclCommInitRank")

    def _check(self, status, what):
        if status != 0:
            raise _lib.DoppelError(f"{what}: {self.rccl.ncclGetErrorString(status).decode()} (rccl status {status})")

    def all_gather(self, send, stream=None, out=None):
        """send: DeviceArray, same shape on every rank -> DeviceArray [world_size * send.shape[0], ...]."""
        if out is None:
            out = _lib.DeviceArray((self.world_size * send.shape[0],) + tuple(send.shape[1:]), send.dtype, self.device)
        count = int(np.prod(send.shape, dtype=np.int64))
        self._check(self.rccl.ncclAllGather(send.ptr, out.ptr, count, _NCCL_TYPES[send.dtype], self.comm,
                                            ctypes.c_void_p(stream or 0)), "ncclAllGather")
        return out

    def close(self):
        if getattr(self, "comm", None):
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None


class RowGather:
    """The one collective of the path: every rank's int32[q_local, k] top-k rows -> int32[n_queries, k] in query
    order on every rank.  Shards may differ by one row; they travel padded to the longest shard and are compacted
    afterwards.  Buffers are allocated once and reused by every `gather` (nothing is allocated in a timed region)."""

    def __init__(self, communicator, n_queries, k, device=0):
        self.communicator, self.n_queries, self.k, self.device = communicator, n_queries, k, device
        self.sizes = shard_sizes(n_queries, communicator.world_size)
        self.longest = max(self.sizes)
        self.even = all(size == self.longest for size in self.sizes)
        self.local = self.sizes[communicator.rank]
        self.padded = self.gathered = self.compact = None
        if communicator.on_device:
            world = communicator.world_size
            self.gathered = _lib.DeviceArray((world * self.longest, k), np.int32, device)
            if not self.even:
                self.padded = _lib.DeviceArray((self.longest, k), np.int32, device)
                _lib.check(_lib.lib().ds_memset(self.padded.ptr, 0xff, self.padded.nbytes, device), "pad")
                self.compact = _lib.DeviceArray((n_queries, k), np.int32, device)

    def gather(self, local_rows, stream=None):
        """local_rows: numpy int32[q_local, k] (HostCommunicator) or DeviceArray / device pointer (RcclCommunicator)."""
        if not self.communicator.on_device:
            local_rows = np.ascontiguousarray(local_rows, dtype=np.int32)
            if local_rows.shape != (self.local, self.k):
                raise ValueError("local_rows does not match this rank's shard")
            padded = np.full((self.longest, self.k), -1, dtype=np.int32)
            padded[:self.local] = local_rows
            gathered = self.communicator.all_gather(padded)
            return np.concatenate([gathered[r * self.longest:r * self.longest + size]
                                   for r, size in enumerate(self.sizes)])
        lib, row_bytes = _lib.lib(), self.k * 4
        if isinstance(local_rows, _lib.DeviceArray):
            pointer = local_rows.ptr
        else:
            pointer = local_rows if isinstance(local_rows, ctypes.c_void_p) else ctypes.c_void_p(int(local_rows))
        stream_pointer = ctypes.c_void_p(stream or 0)
        send = _lib.DeviceArray.view(pointer, (self.longest, self.k), np.int32, self.device)
        if not self.even:
            _lib.check(lib.ds_memcpy_d2d_async(self.padded.ptr, pointer, self.local * row_bytes, self.device,
                                               stream_pointer), "pad copy")
            send = self.padded
        self.communicator.all_gather(send, stream, out=self.gathered)
        if self.even:
            return self.gathered
        at = 0
        for r, size in enumerate(self.sizes):
            source = ctypes.c_void_p(self.gathered.ptr.value + r * self.longest * row_bytes)
            target = ctypes.c_void_p(self.compact.ptr.value + at * row_bytes)
            _lib.check(lib.ds_memcpy_d2d_async(target, source, size * row_bytes, self.device, stream_pointer), "compact")
            at += size
        return self.compact
