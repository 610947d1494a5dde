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
