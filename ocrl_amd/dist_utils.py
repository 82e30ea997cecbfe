"""Image-batch data parallelism: one process per GPU, one all-reduce of the flat fp32 gradient buffer per step.  The reference has
no data-parallel training; because every rank's loss is sum/B_local with equal B_local, the mean of the rank gradients equals the
global-batch gradient, after which clip + Adam run identically everywhere.

Two transports for the same SUM all-reduce, both RCCL over xGMI:
  * default: torch.distributed's "nccl" backend (= RCCL on ROCm; "gloo" in the CPU tests);
  * OCRL_COMM=cabi: the library's own communicator (include/ocrl_hip.h ocrl_comm_*, librccl opened by libocrl_hip.so), bootstrapped
    by broadcasting the 128-byte unique id over the already-initialised torch.distributed group.  This is the path a host without
    torch.distributed binds."""
import ctypes
import os

import torch

_cabi = None


def active_dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class CabiComm:
    """ocrl_comm_* communicator for this process's current GPU"""

    def __init__(self, dist):
        from . import _lib
        self._lib = _lib
        self.L = _lib.lib()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(self.L.ocrl_comm_unique_id(uid, 128))
        box = [uid.raw if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = ctypes.create_string_buffer(box[0], 128)
        self.h = ctypes.c_void_p()
        _lib.check(self.L.ocrl_comm_init(ctypes.byref(self.h), self.rank, self.world, uid))

    def allreduce_(self, flat):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        st = ctypes.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
        self._lib.check(self.L.ocrl_comm_allreduce(self.h, self._lib.ptr(flat), flat.numel(), st))

    def close(self):
        if self.h:
            self.L.ocrl_comm_destroy(self.h)
            self.h = None


def allreduce_grads_(flat_g):
    """in-place SUM all-reduce of the flat gradient buffer; returns the scale (1/world) the optimiser kernel
    folds into its clip coefficient, so the mean costs no extra pass over the buffer"""
    global _cabi
    dist = active_dist()
    if dist is None:
        return 1.0
    if os.environ.get("OCRL_COMM", "") == "cabi" and flat_g.is_cuda:
        if _cabi is not None and (_cabi.rank, _cabi.world) != (dist.get_rank(), dist.get_world_size()):
            shutdown()          # the process group was destroyed and re-initialised with another shape: the old communicator is stale
        if _cabi is None:
            _cabi = CabiComm(dist)
        _cabi.allreduce_(flat_g)
    else:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def shutdown():
    global _cabi
    if _cabi is not None:
        _cabi.close()
        _cabi = None
