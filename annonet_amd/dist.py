"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm) for the single
exchange step of the path — the all-reduce of the flat fp32 gradient bucket — and tile sharding for inference."""
import numpy as np


class _DevicePointer:
    """Zero-copy view of library-owned HBM for torch (CUDA array interface v2)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


def device_tensor(ptr, count, typestr="<f4", device=None):
    import torch
    return torch.as_tensor(_DevicePointer(ptr, count, typestr), device=device or torch.device("cuda", torch.cuda.current_device()))


def grad_bucket_tensor(trainer):
    """The trainer's flat gradient bucket (n_params fp32 + 1 trailing slot = the loss) as a torch tensor."""
    ptr, n = trainer.grad_buffer()
    return device_tensor(ptr, n)


def data_parallel_step(trainer, bucket, d_images, d_labels, d_weights, n, h, w, world_size, group=None, force_collective=False):
    """One optimiser step of a data-parallel job.  The loss scale uses the GLOBAL batch (n * world_size), so the
    all-reduce is a plain SUM and every rank then applies the identical update."""
    trainer.forward_backward_device(d_images, d_labels, d_weights, n, h, w, n * world_size)
    if world_size > 1 or force_collective:
        import torch.distributed as dist
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    trainer.apply_update(1.0)


def shard_tiles(tiles, rank, world_size):
    """Tile sharding for multi-GPU inference: contiguous row-major chunks, so a rank's tiles are spatial neighbours."""
    n = len(tiles)
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return tiles[lo:hi]


def reduce_shard_planes(blended_np_list):
    """Sum per-rank blended planes in rank order (fixed order => reproducible overlap sums)."""
    out = np.zeros_like(blended_np_list[0])
    for b in blended_np_list:
        out = out + b
    return out
