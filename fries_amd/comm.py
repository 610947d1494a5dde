"""Ranks for the FRI engine: the two collectives of include/fries_hip.h's ``fries_comm`` on top of
torch.distributed, one process per GPU.

The reference is an MPI program (FRIES/vec_utils.hpp:991-1019 MPI_Alltoallv of the pending adds,
FRIES/compress_utils.hpp:170-231 MPI_Allgather behind every sum_mpi).  Here the same exchanges run
as RCCL collectives over xGMI (backend "nccl") on staging tensors the engine fills and reads with
its own kernels; the callbacks enqueue them under the engine's HIP stream (``ExternalStream``), so
the data path needs no host synchronisation.  Backend "gloo" takes the same code path and is what the
tests use to run several ranks on one GPU.

``TorchComm`` only moves bytes: every sum over ranks is done by the engine in rank order.
"""
from __future__ import annotations

import ctypes as C
import sys
import traceback

SMALL_BYTES = 2048          # FRIES_COMM_SMALL_BYTES
REC_BYTES = 16              # (determinant, value) record of the spawn exchange

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)


class CommStruct(C.Structure):
    """struct fries_comm (include/fries_hip.h)"""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("size", C.c_int32),
                ("small_send", C.c_void_p), ("small_recv", C.c_void_p), ("big_send", C.c_void_p), ("big_recv", C.c_void_p),
                ("big_bytes", C.c_uint64), ("allgather", ALLGATHER_FN), ("alltoallv", ALLTOALLV_FN)]


def big_bytes_for(mat_nonz: int) -> int:
    """Staging size fries_frisys_setup asks for: one record per possible spawn (mat_nonz + 4096)."""
    return REC_BYTES * (int(mat_nonz) + 4096)


class TorchComm:
    """fries_comm over a torch.distributed process group.  ``device``: the torch device holding the staging
    tensors (a CUDA device in production; "cpu" exercises the same byte movement under gloo without a GPU)."""

    def __init__(self, mat_nonz: int, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.device = torch.device(device)
        self.big_bytes = big_bytes_for(mat_nonz)
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.small_send = torch.zeros(SMALL_BYTES, **u8)
        self.small_recv = torch.zeros(SMALL_BYTES * self.size, **u8)
        self.big_send = torch.zeros(self.big_bytes, **u8)
        self.big_recv = torch.zeros(self.big_bytes, **u8)
        self._ext = {}
        self._bound = None          # engine stream made torch's current stream (see _enter)
        self._views = {}            # byte count -> (recv view, send view): slicing costs microseconds per call otherwise
        self.n_allgather = 0
        self.n_alltoallv = 0
        self.error = None
        # keep the CFUNCTYPE objects alive as long as the struct
        self._ag = ALLGATHER_FN(self._allgather)
        self._a2a = ALLTOALLV_FN(self._alltoallv)
        self.struct = CommStruct(None, self.rank, self.size, self.small_send.data_ptr(), self.small_recv.data_ptr(),
                                 self.big_send.data_ptr(), self.big_recv.data_ptr(), self.big_bytes, self._ag, self._a2a)

    # The engine's stream as torch's current stream, so that collectives order themselves after the engine's kernels.  The
    # callbacks run on the thread that called into the engine, ~50 times per iteration: the stream is made current once per
    # engine stream instead of entering a context manager on every call.
    def _enter(self, stream_ptr):
        if self.device.type != "cuda" or not stream_ptr or self._bound == stream_ptr:
            return
        torch = self.torch
        ext = self._ext.get(stream_ptr)
        if ext is None:
            ext = torch.cuda.ExternalStream(stream_ptr, device=self.device)
            self._ext[stream_ptr] = ext
        torch.cuda.set_stream(ext)
        self._bound = stream_ptr

    def n_collectives(self) -> int:
        return self.n_allgather + self.n_alltoallv

    def release(self):
        """Called before the engine's stream is destroyed: torch must not keep it as its current stream."""
        if self._bound is not None and self.device.type == "cuda":
            self.torch.cuda.set_stream(self.torch.cuda.default_stream(self.device))
        self._bound = None
        self._ext.clear()

    def _allgather(self, user, nbytes, stream):
        try:
            n = int(nbytes)
            self._enter(stream)
            v = self._views.get(n)
            if v is None:
                v = (self.small_recv[:n * self.size], self.small_send[:n])
                self._views[n] = v
            self.dist.all_gather_into_tensor(v[0], v[1], group=self.group)
            self.n_allgather += 1
            return 0
        except Exception as e:       # an exception must not unwind through the C frames
            self.error = e
            traceback.print_exc(file=sys.stderr)
            return 1

    def _alltoallv(self, user, send_bytes, recv_bytes, stream):
        try:
            sb = [int(send_bytes[i]) for i in range(self.size)]
            rb = [int(recv_bytes[i]) for i in range(self.size)]
            self._enter(stream)
            self.dist.all_to_all_single(self.big_recv[:sum(rb)], self.big_send[:sum(sb)], rb, sb, group=self.group)
            self.n_alltoallv += 1
            return 0
        except Exception as e:
            self.error = e
            traceback.print_exc(file=sys.stderr)
            return 1


class _NativeComm:
    """A transport of csrc/comm_native.hip behind the same attributes TorchComm offers the engine (struct, big_bytes,
    counters, release): no Python in the collectives."""

    def __init__(self, handle, lib, size, rank, big_bytes):
        self.lib, self.h, self.size, self.rank, self.big_bytes = lib, handle, size, rank, big_bytes
        self.struct = CommStruct()
        if lib.fries_transport_comm(self.h, C.byref(self.struct)) != 0:
            raise RuntimeError(lib.fries_last_error().decode())
        self.error = None

    def _counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.fries_transport_counts(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def n_allgather(self):
        return self._counts()[0]

    @property
    def n_alltoallv(self):
        return self._counts()[1]

    def n_collectives(self) -> int:
        return sum(self._counts())

    def release(self):
        pass

    def destroy(self):
        if self.h:
            self.lib.fries_transport_destroy(self.h)
            self.h = None


class RcclComm(_NativeComm):
    """fries_comm on librccl directly (ncclAllGather / ncclAllToAllv on the engine's stream), one process per GPU.  The
    128-byte communicator id is made by rank 0 and broadcast through the already initialised torch.distributed group
    (bootstrap only: no torch call remains on the data path)."""

    def __init__(self, mat_nonz: int, device: int, dist=None, group=None):
        from .engine import load_library
        lib = load_library()
        rank, size = (dist.get_rank(group), dist.get_world_size(group)) if dist is not None else (0, 1)
        idb = (C.c_uint8 * 128)()
        if rank == 0 and lib.fries_rccl_unique_id(idb) != 0:
            raise RuntimeError(lib.fries_last_error().decode())
        if dist is not None:
            import torch
            dev = torch.device("cuda", device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.tensor(list(idb), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0, group=group)
            idb = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        bb = big_bytes_for(mat_nonz)
        if lib.fries_rccl_create(C.byref(h), idb, rank, size, device, bb) != 0:
            raise RuntimeError(lib.fries_last_error().decode())
        super().__init__(h, lib, size, rank, bb)


class LocalGroup:
    """`size` ranks as threads of this process (csrc/comm_native.hip, "local"): group.comm(rank, device) gives rank's
    fries_comm; each rank's engine must then be driven from its own thread."""

    def __init__(self, size: int, mat_nonz: int):
        from .engine import load_library
        self.lib = load_library()
        self.size = size
        self.big_bytes = big_bytes_for(mat_nonz)
        self.g = C.c_void_p()
        if self.lib.fries_local_group_create(C.byref(self.g), size, self.big_bytes) != 0:
            raise RuntimeError(self.lib.fries_last_error().decode())
        self.members = []

    def comm(self, rank: int, device: int = 0) -> _NativeComm:
        h = C.c_void_p()
        if self.lib.fries_local_create(C.byref(h), self.g, rank, device) != 0:
            raise RuntimeError(self.lib.fries_last_error().decode())
        c = _NativeComm(h, self.lib, self.size, rank, self.big_bytes)
        self.members.append(c)
        return c

    def destroy(self):
        for c in self.members:
            c.destroy()
        self.members = []
        if self.g:
            self.lib.fries_local_group_destroy(self.g)
            self.g = None
