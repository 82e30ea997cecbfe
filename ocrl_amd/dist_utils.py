"""Image-batch data parallelism: one process per GPU, one all-reduce of the flat fp32 gradient buffer per step
(RCCL over xGMI through torch.distributed's "nccl" backend on ROCm; "gloo" in the CPU tests).  The reference has
no data-parallel training; because every rank's loss is sum/B_local with equal B_local, the mean of the rank
gradients equals the global-batch gradient, after which clip + Adam run identically everywhere."""
import torch


def active_dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def allreduce_grads_(flat_g):
    """in-place SUM all-reduce of the flat gradient buffer; returns the scale (1/world) the optimiser kernel
    folds into its clip coefficient, so the mean costs no extra pass over the buffer"""
    dist = active_dist()
    if dist is None:
        return 1.0
    dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()
