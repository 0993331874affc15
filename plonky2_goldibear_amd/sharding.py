"""One process per GPU, independent circuits, no data-path collective.

The reference proves one circuit per call on one machine's Rayon pool; here independent proofs shard
one-per-GPU: proof `i` goes to rank `i mod world`.  torch.distributed is used only for the timing
barrier and the max-over-ranks reduction (backend "nccl" = RCCL on the GPU node, "gloo" in CPU tests).
"""
import torch
import torch.distributed as dist

_GROUP = None   # the process group that carries the barrier and the reductions (None: the default group)


def set_group(group):
    """bench.py: the default group is gloo (a control plane that cannot fail for GPU reasons); when an RCCL group came up on every
    rank it carries the barrier and the reductions instead."""
    global _GROUP
    _GROUP = group


def _device(device):
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(_GROUP) == "nccl" else torch.device("cpu")


def circuits_for_rank(num_circuits, world, rank):
    """Static partition: proof i -> device i mod world."""
    return [i for i in range(num_circuits) if i % world == rank]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=_GROUP)


def max_over_ranks(seconds, device=None):
    """Whole-job wall time = the slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_GROUP)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_GROUP)
    return float(t.item())


def aggregate_throughput(units_this_rank, seconds_this_rank):
    """value = units all ranks processed / max-over-ranks time (the bench.py contract)."""
    return sum_over_ranks(units_this_rank) / max_over_ranks(seconds_this_rank)
