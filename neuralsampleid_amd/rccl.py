"""Direct RCCL binding (ctypes over librccl.so's C API) for the data-parallel step.

Why not torch.distributed's ProcessGroupNCCL on the data path: the whole contrastive step is replayed as ONE hipGraph,
collectives included. ProcessGroupNCCL owns a hidden stream plus a watchdog thread that polls `hipEventQuery` on the end
events of earlier eager collectives; once its stream has joined a capture those queries fail with
hipErrorCapturedEvent and the watchdog aborts the process (seen intermittently on MI355X, gpurun_out/fc.err). Here the
communicator is ours: every collective is enqueued on ONE dedicated high-priority HIP stream that forks from / joins the
compute streams with events, there is no helper thread, errors are return codes -> RuntimeError, and the same calls
work eagerly and under stream capture.

Bootstrap (ncclUniqueId, 128 bytes) travels over whatever torch.distributed group exists (gloo is enough) or, with one
rank, nowhere. librccl.so is the copy torch itself ships/loads, so the process holds exactly one RCCL."""
import ctypes
import os
from typing import Iterable, Optional

import torch

NCCL_UNIQUE_ID_BYTES = 128
_DTYPE = {torch.float32: 7, torch.float64: 8, torch.bfloat16: 9, torch.float16: 6, torch.int32: 2, torch.int64: 4,
          torch.uint8: 1, torch.int8: 0}
SUM, PROD, MAX, MIN = 0, 1, 2, 3


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _find_library() -> str:
    cands = [os.environ.get("NSID_RCCL_LIB"),
             os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"),
             "/opt/rocm/lib/librccl.so", "librccl.so"]
    for c in cands:
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("librccl.so not found (set NSID_RCCL_LIB)")


def lib():
    """the loaded librccl.so with argtypes declared; raises if RCCL is not available (no silent fallback)"""
    global _lib
    if _lib is not None:
        return _lib
    L = ctypes.CDLL(_find_library(), mode=ctypes.RTLD_GLOBAL)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ncclGetErrorString.restype = ctypes.c_char_p
    L.ncclGetErrorString.argtypes = [ci]
    for name, args in {
        "ncclGetVersion": [ctypes.POINTER(ci)],
        "ncclGetUniqueId": [ctypes.POINTER(_UniqueId)],
        "ncclCommInitRank": [ctypes.POINTER(vp), ci, _UniqueId, ci],
        "ncclCommDestroy": [vp],
        "ncclCommAbort": [vp],
        "ncclCommCount": [vp, ctypes.POINTER(ci)],
        "ncclCommUserRank": [vp, ctypes.POINTER(ci)],
        "ncclCommGetAsyncError": [vp, ctypes.POINTER(ci)],
        "ncclAllReduce": [vp, vp, sz, ci, ci, vp, vp],
        "ncclAllGather": [vp, vp, sz, ci, vp, vp],
        "ncclBroadcast": [vp, vp, sz, ci, ci, vp, vp],
        "ncclGroupStart": [],
        "ncclGroupEnd": [],
    }.items():
        fn = getattr(L, name)
        fn.restype = ci
        fn.argtypes = args
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"RCCL {what} failed: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def version() -> int:
    v = ctypes.c_int(0)
    _check(lib().ncclGetVersion(ctypes.byref(v)), "ncclGetVersion")
    return v.value


def new_unique_id() -> bytes:
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)     # (.internal would stop at the first NUL)


class _stdout_to_stderr:
    """route file descriptor 1 to 2 for the duration (a program that prints one machine-readable line on stdout must
    not have a library banner in front of it)"""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)


class RcclComm:
    """One RCCL communicator + one dedicated HIP stream. All methods are stream-ordered (never host-synchronising),
    so a step that uses them can be captured in a hipGraph.

    * `all_gather(t)`, `all_reduce_(t)`: fork from the current stream, run on the comm stream, join back — the result
      is usable on the current stream.
    * `all_reduce_async_(t, producers)`: fork from `producers` (default: the current stream) only; the compute streams
      run on. `wait_async()` joins the comm stream into the current stream (before the optimiser reads the gradients).
    """

    def __init__(self, rank: int, world: int, unique_id: bytes, device: Optional[torch.device] = None):
        if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("ncclUniqueId must be 128 bytes")
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = _UniqueId()
        ctypes.memmove(ctypes.byref(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()          # make sure the HIP context exists on this device
            with _stdout_to_stderr():                          # RCCL prints its version banner on stdout
                _check(lib().ncclCommInitRank(ctypes.byref(h), world, uid, rank), "ncclCommInitRank")
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            self.stream = torch.cuda.Stream(device=self.device, priority=min(lo, hi))
        self._h = h
        self.calls = 0

    # -- helpers
    def _fork(self, producers: Optional[Iterable[torch.cuda.Stream]]):
        for s in (producers if producers is not None else (torch.cuda.current_stream(self.device),)):
            if s != self.stream:
                self.stream.wait_stream(s)

    def _join(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def _chk_tensor(self, t: torch.Tensor):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPE):
            raise ValueError("RCCL collectives take contiguous device tensors of a supported dtype")
        if self._h is None:
            raise RuntimeError("communicator already destroyed")

    # -- collectives
    def all_reduce_async_(self, t: torch.Tensor, producers=None, op: int = SUM) -> None:
        self._chk_tensor(t)
        self._fork(producers)
        t.record_stream(self.stream)
        _check(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPE[t.dtype], op, self._h,
                                   self.stream.cuda_stream), "ncclAllReduce")
        self.calls += 1

    def wait_async(self) -> None:
        self._join()

    wait = wait_async          # so a communicator can stand in for a torch.distributed Work handle

    def all_reduce_(self, t: torch.Tensor, op: int = SUM) -> torch.Tensor:
        self.all_reduce_async_(t, None, op)
        self._join()
        return t

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """(n, ...) -> (world*n, ...), rank-major"""
        t = t.contiguous()
        self._chk_tensor(t)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self._fork(None)
        t.record_stream(self.stream)
        out.record_stream(self.stream)
        _check(lib().ncclAllGather(t.data_ptr(), out.data_ptr(), t.numel(), _DTYPE[t.dtype], self._h,
                                   self.stream.cuda_stream), "ncclAllGather")
        self.calls += 1
        self._join()
        return out

    def barrier(self) -> None:
        """device-side barrier (1-element all-reduce) followed by a host wait; not for use under capture"""
        t = torch.zeros(1, device=self.device)
        self.all_reduce_(t)
        torch.cuda.current_stream(self.device).synchronize()

    def async_error(self) -> int:
        e = ctypes.c_int(0)
        _check(lib().ncclCommGetAsyncError(self._h, ctypes.byref(e)), "ncclCommGetAsyncError")
        return e.value

    def count(self) -> int:
        """ncclCommCount: the number of ranks RCCL itself believes are in this communicator"""
        n = ctypes.c_int(0)
        _check(lib().ncclCommCount(self._h, ctypes.byref(n)), "ncclCommCount")
        return n.value

    def abort(self) -> None:
        """ncclCommAbort: tear the communicator down WITHOUT waiting for outstanding collectives (a peer died or hung);
        the only safe next step for the process is to exit"""
        if self._h is not None:
            lib().ncclCommAbort(self._h)
            self._h = None

    def destroy(self) -> None:
        if self._h is not None:
            self.stream.synchronize()
            lib().ncclCommDestroy(self._h)
            self._h = None


def init_comm(rank: int, world: int, device: Optional[torch.device] = None, group=None) -> RcclComm:
    """Create the communicator; with world > 1 the unique id is broadcast from rank 0 over torch.distributed `group`
    (any backend — it is 128 bytes of host data)."""
    if world == 1:
        return RcclComm(0, 1, new_unique_id(), device)
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("world > 1: initialise a torch.distributed group (gloo is enough) to carry the ncclUniqueId")
    box = [new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group, device=torch.device("cpu")
                               if dist.get_backend(group) == "gloo" else None)
    return RcclComm(rank, world, box[0], device)
