"""Thin collective helpers that degrade to no-ops when no process group exists.

The reference only ever issues ``barrier()`` explicitly (``trainer/trainer.py:132,135,169,172``);
everything else is implicit inside DDP.  These helpers are what the rebuilt Trainer, reducer and
distributed validation use, so that a single-process run needs no rendezvous at all.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def barrier():
    if is_dist() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def broadcast_(tensor, src=0):
    if is_dist() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def all_reduce_sum_(tensor):
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def all_reduce_max_(tensor):
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
    return tensor


def comm_device():
    """Device on which tensors handed to collectives must live for the active backend."""
    if is_dist() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_reduce_scalars(values, op="sum"):
    """All-reduce a small list of python floats; returns python floats."""
    if not (is_dist() and dist.get_world_size() > 1):
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64 if comm_device().type == "cpu"
                     else torch.float32, device=comm_device())
    if op == "sum":
        all_reduce_sum_(t)
    elif op == "min":
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    else:
        all_reduce_max_(t)
    return [float(v) for v in t.cpu().tolist()]


def broadcast_object(obj, src=0):
    if not (is_dist() and dist.get_world_size() > 1):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, device=comm_device())
    return box[0]


def all_gather_object(obj):
    """List of every rank's ``obj`` (``[obj]`` without a process group)."""
    if not (is_dist() and dist.get_world_size() > 1):
        return [obj]
    out = [None] * dist.get_world_size()
    if dist.get_backend() == "nccl":
        torch.cuda.set_device(torch.cuda.current_device())
    dist.all_gather_object(out, obj)
    return out
