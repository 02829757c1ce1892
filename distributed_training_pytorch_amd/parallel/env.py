"""Rank / device discovery and process-group lifecycle.

Parity target: reference ``trainer/trainer.py:48-50`` (LOCAL_RANK / RANK / WORLD_SIZE from the
torchrun environment) and ``trainer/trainer.py:74-82`` (``ddp_setup(backend)`` /
``destroy_process()``).

MI355X-first differences:
  * ``backend="nccl"`` is RCCL on ROCm; ``backend="gloo"`` is a real, tested CPU path (the
    reference hard-wires ``torch.cuda.set_device``);
  * works without torchrun (single process: rank 0 of world 1, no process group needed);
  * explicit timeout on ``init_process_group`` so a dead rank surfaces as an
    error instead of a silent 10-minute watchdog wait.
"""
import datetime
import os

import torch
import torch.distributed as dist


class DistEnv(object):
    """Snapshot of the launcher-provided topology."""

    def __init__(self, local_rank=0, rank=0, world_size=1):
        self.local_rank = int(local_rank)
        self.rank = int(rank)
        self.world_size = int(world_size)

    @classmethod
    def from_env(cls):
        return cls(os.environ.get("LOCAL_RANK", 0), os.environ.get("RANK", 0),
                   os.environ.get("WORLD_SIZE", 1))

    def __repr__(self):
        return "DistEnv(local_rank=%d, rank=%d, world_size=%d)" % (
            self.local_rank, self.rank, self.world_size)


def gpu_available():
    return torch.cuda.is_available()


def apply_rccl_env_defaults(nnodes=None):
    """Export the RCCL / HIP tunables this framework wants (``launch.rccl_env_defaults``) unless the
    operator already set them.  ``run.sh`` and the self-launcher export the same values before the
    workers start; calling it here as well covers any other launcher (torchrun by hand, SLURM).
    Must run before the first RCCL communicator is created — ``ddp_setup`` does."""
    from ..launch import rccl_env_defaults
    if nnodes is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world)
        nnodes = max(1, world // max(local, 1))
    applied = {}
    for k, v in rccl_env_defaults(nnodes).items():
        if k not in os.environ:
            os.environ[k] = v
            applied[k] = v
    return applied


def default_device(local_rank=None):
    """cuda:<local_rank> when a GPU is visible and the process group (if any) is not gloo."""
    if local_rank is None:
        local_rank = DistEnv.from_env().local_rank
    if gpu_available():
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
            return torch.device("cpu")
        return torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


def ddp_setup(backend="nccl", timeout_s=None):
    """Bind this process to its GPU and join the env:// rendezvous.

    Single-process runs (no ``MASTER_ADDR`` / WORLD_SIZE==1 without torchrun) skip the process
    group entirely; every collective helper in :mod:`comm` degrades to a no-op in that case.
    """
    env = DistEnv.from_env()
    apply_rccl_env_defaults()
    if backend == "nccl" and not gpu_available():
        raise RuntimeError(
            "backend='nccl' (RCCL) needs a visible GPU; use backend='gloo' for the CPU path")
    if backend == "nccl":
        torch.cuda.set_device(env.local_rank % max(torch.cuda.device_count(), 1))
    if dist.is_initialized():
        return env
    if env.world_size == 1 and "MASTER_ADDR" not in os.environ:
        return env  # plain `python main.py`
    if timeout_s is None:
        timeout_s = float(os.environ.get("DTP_PG_TIMEOUT_S", "600"))
    kwargs = dict(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl":
        try:
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
        except Exception:
            pass
    try:
        dist.init_process_group(**kwargs)
    except TypeError:
        kwargs.pop("device_id", None)
        dist.init_process_group(**kwargs)
    return env


def destroy_process():
    try:     # native RCCL communicators go first, while the process group can still be used
        from .rccl_native import _close_all
        _close_all()
    except Exception:
        pass
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
