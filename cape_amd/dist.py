"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests).

The reference has no parallelism at all (single TF session); meshes of a batch are
independent everywhere on the path (per-sample conv/pool/group-norm, batch-mean losses), so the
only exchange is the mean of the flat gradient buffer -- ONE exchange per variable group per
step (65 MB fp32 for the affine-nz64 generator).  Global-norm clipping runs after the reduce.

The exchange itself comes in three forms (``CAPE_DP_COLLECTIVE`` / ``GradAverager(mode=...)``; SURVEY 8(e) sized the bucket for
the third):
    allreduce   one RCCL all-reduce (RCCL picks ring / tree; a ring is bound by ONE xGMI link: ~2 S (N-1)/N per link)
    rsag        RCCL reduce-scatter + all-gather (the same volume as two collectives; lets the library pick per phase)
    direct      all-to-all of the N slices (every pair of GPUs has its own xGMI link: S/N per link, all links at once), a
                fixed-order local sum of the N received slices, then an all-gather of the reduced slices (S/N per link again)
All three leave the SUM (or, without ``defer_mean``, the mean) of the ranks' buckets in place on every rank, bit-identically
across ranks.  `bench.py --gpus N` times each of them on the real bucket before the timed steps and reports the figures.
"""
import os

import torch
import torch.distributed as dist

MODES = ("allreduce", "rsag", "direct")


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
    """``hook(flat_grad)``: in-place mean over ranks of a flat gradient bucket.

    ``defer_mean`` (set by the step runner): leave the SUM in the bucket; the optimiser kernels multiply by ``grad_scale``
    = 1 / world themselves (cape_flat_gradnorm / cape_flat_*_update), which removes the bucket-sized division launch."""

    def __init__(self, group=None, always=False, mode=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.always = bool(always) and dist.is_initialized()      # issue the collectives even for one rank (self-test)
        self.mode = mode or os.environ.get("CAPE_DP_COLLECTIVE", "allreduce")
        if self.mode not in MODES:
            raise ValueError("CAPE_DP_COLLECTIVE must be one of %s, not %r" % (", ".join(MODES), self.mode))
        self.defer_mean = False
        self.enabled = True               # bench.py switches the exchange off for its exposed-time measurement
        self._scratch = {}
        self._parked = []                 # outgrown scratch buffers, kept alive (see _buf)
        self._side = None

    @property
    def grad_scale(self):
        return 1.0 / self.world if self.defer_mean else 1.0

    def active(self):
        return self.enabled and (self.world > 1 or self.always)

    def _buf(self, key, n, like):
        # one buffer per (name, STREAM): the synchronous path (current stream) and the asynchronous one (side stream) never share
        # scratch, and a buffer that has to grow is parked instead of released -- work queued on its stream may still use it
        # (the caching allocator would hand the block to another stream's allocation) -- ADVICE r05
        sid = torch.cuda.current_stream(like.device).cuda_stream if like.is_cuda else 0
        k = (key, sid)
        b = self._scratch.get(k)
        if b is None or b.numel() < n or b.device != like.device:
            if b is not None:
                self._parked.append(b)
            b = self._scratch[k] = torch.empty(n, device=like.device, dtype=like.dtype)
        return b[:n]

    def exchange_sum(self, t, mode=None):
        """t <- sum over ranks of t, on the current stream (blocking collective semantics of torch.distributed)."""
        mode = mode or self.mode
        n, w = t.numel(), self.world
        if mode == "allreduce" or n % w or not t.is_contiguous():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t
        shard = self._buf("shard", n // w, t)
        if mode == "rsag":
            dist.reduce_scatter_tensor(shard, t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            recv = self._buf("recv", n, t)
            dist.all_to_all_single(recv, t, group=self.group)     # recv[r] = rank r's copy of MY slice
            torch.sum(recv.view(w, n // w), dim=0, out=shard)     # fixed order r = 0 .. w-1: every rank sums its own slice once
        dist.all_gather_into_tensor(t, shard, group=self.group)
        return t

    def __call__(self, flat_grad):
        if self.active():
            self.exchange_sum(flat_grad)
            if not self.defer_mean:
                flat_grad.div_(self.world)
        return flat_grad

    def start(self, flat_grad):
        """Asynchronous form: the exchange is enqueued on a side stream behind the work queued so far on the current stream
        and runs beside whatever is queued next; ``finish`` makes the current stream wait for it."""
        if not self.active():
            return None
        if dist.get_backend(self.group) != "nccl":
            self(flat_grad)                  # host-staged transports (gloo) cannot overlap with device work anyway
            return None
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self(flat_grad)
            done = torch.cuda.Event()
            done.record(self._side)
        return (done, flat_grad)

    def finish(self, handle):
        if handle is not None:
            handle[0].wait(torch.cuda.current_stream())


def probe_collectives(hook, nelem, device, iters=5, modes=MODES):
    """Time every exchange form on a bucket of ``nelem`` fp32 elements (all ranks call this together): per mode the MAX over
    ranks of the mean time of ``iters`` back-to-back exchanges, in ms, or an error string.  Never raises: a form the backend
    refuses is reported and left out."""
    import time
    out = {}
    device = torch.device(device)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    buf = torch.ones(nelem, device=device, dtype=torch.float32)
    for mode in modes:
        try:
            hook.exchange_sum(buf, mode)                # warm-up (communicator set-up, scratch allocation)
            sync()
            dist.barrier(group=hook.group)
            t0 = time.perf_counter()
            for _ in range(iters):
                hook.exchange_sum(buf, mode)
            sync()
            out[mode] = round(max_over_ranks(1e3 * (time.perf_counter() - t0) / iters, device), 4)
            buf.fill_(1.0)
        except Exception as e:                          # noqa: BLE001 -- report, do not lose the run
            out[mode] = "%s: %s" % (type(e).__name__, str(e)[:160])
    return out


def rank_inventory(device):
    """What every rank actually runs on: [(rank, local device index, device name, backend)] gathered to all ranks."""
    on_gpu = torch.device(device).type == "cuda"
    me = dict(rank=dist.get_rank() if dist.is_initialized() else 0, device=int(torch.cuda.current_device()) if on_gpu else -1,
              name=torch.cuda.get_device_name(torch.cuda.current_device()) if on_gpu else "cpu", pid=os.getpid(),
              backend=dist.get_backend() if dist.is_initialized() else None)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, me)
    return got


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
