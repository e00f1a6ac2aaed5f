"""ctypes binding of the four RCCL entry points the data-parallel step needs (rccl.h: ncclGetUniqueId, ncclCommInitRank,
ncclAllReduce, ncclCommDestroy) on the librccl.so that PyTorch-ROCm itself loaded -- the same library, the same xGMI rings,
without torch.distributed's per-collective bookkeeping (zhusuan.dataparallel.DirectAllReduce says why).

(torch.cuda.nccl has Python bindings for these, but its ``init_rank`` raises under this interpreter -- "PY_SSIZE_T_CLEAN
macro must be defined for '#' formats" -- so the C API is bound here.)
"""
import ctypes
import os

import torch

UNIQUE_ID_BYTES = 128          # NCCL_UNIQUE_ID_BYTES
SUM = 0                        # ncclSum
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.float16: 6, torch.bfloat16: 9, torch.int32: 2, torch.int64: 4}


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * UNIQUE_ID_BYTES)]


_LIB = None


def lib():
    """librccl.so as loaded by torch (``torch/lib``), else the system's."""
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclAllReduce, L.ncclCommDestroy):
            f.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        msg = lib().ncclGetErrorString(rc)
        raise RuntimeError("%s failed with RCCL error %d: %s" % (what, rc, msg.decode() if msg else "?"))


def unique_id():
    """A fresh ncclUniqueId as 128 bytes (made by ONE rank, handed to all)."""
    uid = UniqueId()
    _check(lib().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(ctypes.byref(uid), UNIQUE_ID_BYTES)          # (the field itself reads as a NUL-terminated string)


def comm_init_rank(world, uid_bytes, rank):
    """Join the communicator (collective over the ranks holding ``uid_bytes``; on the CURRENT device).  Returns the handle."""
    if len(uid_bytes) != UNIQUE_ID_BYTES:
        raise ValueError("an ncclUniqueId is %d bytes" % UNIQUE_ID_BYTES)
    uid = UniqueId()
    ctypes.memmove(ctypes.byref(uid), uid_bytes, UNIQUE_ID_BYTES)
    comm = ctypes.c_void_p()
    _check(lib().ncclCommInitRank(ctypes.byref(comm), int(world), uid, int(rank)), "ncclCommInitRank")
    return comm


def all_reduce_sum_(t, comm, stream):
    """In-place SUM of the contiguous tensor ``t`` over the communicator's ranks, enqueued on ``stream`` (a raw hipStream_t)."""
    _check(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], SUM, comm, stream), "ncclAllReduce")


def comm_destroy(comm):
    if comm:
        lib().ncclCommDestroy(comm)
