"""Multi-GPU plumbing: one process per GPU, trajectories sharded over ranks, ONE all-reduce of a
packed [loss_sum | grad_sum] buffer per optimiser step (the jnp.mean at reference
policy/base.py:126-127 and gan/js_policy.py:55).  torch.distributed backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests of this logic."""

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(count, rank=None, world_size=None):
    """Contiguous [lo, hi) of `count` items owned by `rank` (sizes differ by at most one)."""
    if rank is None:
        rank, world_size = world()
    base, rem = divmod(count, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_from_sums(packed, local_count):
    """packed = [loss_sum | grad_sum] over this rank's `local_count` samples.  After the call every
    rank holds the global means (sum over ranks / global count), in place.  Single-process: no
    collective, just the division."""
    rank, ws = world()
    if ws > 1:
        cnt = torch.tensor([float(local_count)], dtype=packed.dtype, device=packed.device)
        buf = torch.cat([packed.reshape(-1), cnt])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        packed.copy_(buf[:-1].reshape(packed.shape))
        total = float(buf[-1].item())
    else:
        total = float(local_count)
    packed.mul_(1.0 / total)
    return packed
