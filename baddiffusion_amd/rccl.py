"""RCCL called directly (ctypes on the librccl.so that PyTorch-ROCm ships): the data-parallel gradient exchange of SURVEY 8(e).

Replaces nn.DataParallel's gradient reduction (/root/reference/baddiffusion.py:325) by one `ncclAllReduce` per finished range
of the flat fp32 gradient, enqueued on OUR communication stream: its place in time is fixed by plain HIP stream order (events
recorded by the backward plan), not by a process-group wrapper.  Measured reason: the same
collectives issued through torch.distributed (c10d ProcessGroupNCCL: per-collective event bookkeeping + a watchdog thread that
polls them) cost 0.45 ms EACH beside the backward kernels even on a 1-rank group -- 33.5 instead of 18.3 ms/step.

Bootstrap: rank 0 creates the ncclUniqueId, the 128 bytes travel through whatever torch.distributed group exists (object
broadcast; gloo or nccl) or through nothing at world 1.  torch.distributed stays the launcher-facing layer (rendezvous, barriers,
the scalar MAX of the bench); RCCL moves the gradients.
"""
import ctypes as C
import os

import torch

_NCCL_FLOAT32 = 7
_NCCL_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("BD_RCCL_LIB"), os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1",
             "/opt/rocm/lib/librccl.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError(f"librccl.so not found ({err}); set BD_RCCL_LIB")
    vp = C.c_void_p
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclBroadcast.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetErrorString.restype = C.c_char_p
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclBroadcast", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd"):
        getattr(lib, f).restype = C.c_int
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_load().ncclGetErrorString(rc).decode()} ({rc})")


def unique_id():
    """ncclGetUniqueId as 128 bytes (rank 0 calls it, the launcher's group broadcasts it)"""
    uid = _UniqueId()
    _check(_load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.byref(uid), 128)      # NOT bytes(uid.internal): a c_char array reads as a C string, cut at the first NUL byte


class RcclComm:
    """One RCCL communicator over the ranks of the (already initialised) torch.distributed default group, or a 1-rank
    communicator when there is none.  All calls enqueue on the HIP stream they are given and return at once."""

    def __init__(self, device=None, uid=None):
        """uid: the 128 bytes of unique_id() already shared by the caller (TrainEngine._open_rccl stages the host-side calls with an agreement
        behind each); None: obtain and broadcast it here."""
        import torch.distributed as dist
        lib = _load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        have = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if have else 0
        self.world = dist.get_world_size() if have else 1
        if uid is None:
            box = [unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)
            uid = box[0]
        raw, uid = bytes(uid), _UniqueId()
        if len(raw) != 128:
            raise RuntimeError(f"ncclUniqueId must be 128 bytes, got {len(raw)}")
        C.memmove(C.byref(uid), raw, 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, t, stream):
        """in-place sum of the fp32 tensor / slice `t` over the ranks, enqueued on `stream` (a torch.cuda.Stream)"""
        assert t.dtype == torch.float32 and t.is_contiguous()
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_FLOAT32, _NCCL_SUM, self._comm, stream.cuda_stream), "ncclAllReduce")

    def all_reduce_ranges_(self, flat, ranges, stream):
        """several ranges of one flat fp32 buffer as ONE grouped launch (ncclGroupStart / End)"""
        lib = _load()
        ranges = [(lo, hi) for lo, hi in ranges if hi > lo]
        if not ranges:
            return 0
        if len(ranges) > 1:
            _check(lib.ncclGroupStart(), "ncclGroupStart")
        base, n = flat.data_ptr(), 0
        for lo, hi in ranges:
            p = base + 4 * lo
            _check(lib.ncclAllReduce(p, p, hi - lo, _NCCL_FLOAT32, _NCCL_SUM, self._comm, stream.cuda_stream), "ncclAllReduce")
            n += 4 * (hi - lo)
        if len(ranges) > 1:
            _check(lib.ncclGroupEnd(), "ncclGroupEnd")
        return n

    def broadcast_(self, t, root, stream):
        assert t.dtype == torch.float32 and t.is_contiguous()
        _check(_load().ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_FLOAT32, root, self._comm, stream.cuda_stream), "ncclBroadcast")

    def self_test(self, n=4099):
        """All-reduce a rank-dependent vector (v[i] = (rank + 1) * (i % 7 + 1)) and compare with the closed form
        world * (world + 1) / 2 * (i % 7 + 1) -- exact in fp32.  Returns None when it matches, else a reason string.  Runs once per
        communicator, before the first gradient is trusted to it (bench.py prints the outcome; a multi-GPU node is leased rarely)."""
        s = torch.cuda.current_stream(self.device)
        base = (torch.arange(n, device=self.device) % 7 + 1).float()
        v = base * float(self.rank + 1)
        self.all_reduce_(v, s)
        s.synchronize()
        want = base * (self.world * (self.world + 1) / 2.0)
        if not torch.equal(v, want):
            bad = int((v != want).sum())
            return f"ncclAllReduce self-test: {bad} of {n} elements differ from the closed form at world {self.world} (rank {self.rank})"
        return None

    def destroy(self):
        if self._comm:
            _load().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
