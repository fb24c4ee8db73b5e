"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

The reference has no parallelism at all (single TF session); meshes of a batch are
independent everywhere on the path (per-sample conv/pool/group-norm, batch-mean losses), so the
only exchange is the mean of the flat gradient buffer -- ONE all-reduce per variable group per
step (65 MB fp32 for the affine-nz64 generator).  Global-norm clipping runs after the reduce.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op for world 1)."""
    world, rank, local = env_world()
    if world == 1:
        return world, rank, local
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("CAPE_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_range(total, world, rank):
    """Contiguous [begin, end) slice of ``total`` independent units owned by ``rank``."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class GradAverager(object):
    """``hook(flat_grad)``: in-place mean over ranks of a flat gradient bucket."""

    def __init__(self, group=None, always=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = bool(always) and dist.is_initialized()      # issue the collectives even for one rank (self-test)

    def __call__(self, flat_grad):
        if self.world > 1 or self.always:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.div_(self.world)
        return flat_grad

    def start(self, flat_grad):
        """Asynchronous form: enqueue the sum on the collective stream (it starts once the work queued so far on the
        current stream is done and then runs beside whatever is queued next); ``finish`` makes the current stream wait
        for it and completes the mean."""
        if self.world <= 1 and not self.always:
            return None
        if dist.get_backend(self.group) != "nccl":
            self(flat_grad)                  # host-staged transports (gloo) cannot overlap with device work anyway
            return None
        return (dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat_grad)

    def finish(self, handle):
        if handle is not None:
            work, flat_grad = handle
            work.wait()
            flat_grad.div_(self.world)


def broadcast_flat(flat, src=0, group=None):
    """Make every rank start from rank ``src``'s variables."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
