"""Data-parallel path on CPU: world_size-2 gloo processes.  Covers sharding, the flat-bucket gradient
all-reduce (mean), clip-after-reduce and the momentum update -- the logic bench.py --gpus N runs over
RCCL -- with a small dense stand-in network (the HIP kernels need a GPU; the reduction logic does not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _flat_sgd_step(flat, flat_grad, m, lr=0.1, mom=0.9, clip=5.0):
    gnorm = torch.linalg.vector_norm(flat_grad)
    scale = clip / torch.clamp(gnorm, min=clip)
    m.mul_(mom).addcmul_(flat_grad, scale)
    flat.add_(m, alpha=-lr)
    return gnorm


def _worker(rank, world, port, out_dir, mode="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cape_amd import dist as cdist
    w, r, _ = cdist.init_from_env(backend="gloo")
    assert (w, r) == (world, rank)
    torch.manual_seed(0)
    X, Y = torch.randn(8, 6), torch.randn(8, 3)
    W0 = torch.randn(6 * 3 + 3)
    flat = W0.clone() + rank                 # deliberately different start; rank 0 wins after broadcast
    cdist.broadcast_flat(flat)
    b, e = cdist.shard_range(8, world, rank)
    hook = cdist.GradAverager(mode=mode)
    # deferred mean (what the step runner uses): the SUM stays in the bucket, grad_scale = 1 / world goes to the optimiser kernels
    probe = torch.arange(12, dtype=torch.float32) * (rank + 1)
    hook.defer_mean = True
    hook(probe)
    assert hook.grad_scale == 1.0 / world and torch.equal(probe, torch.arange(12, dtype=torch.float32) * sum(range(1, world + 1)))
    hook.defer_mean = False
    assert hook.grad_scale == 1.0
    odd = torch.ones(7) * (rank + 1)               # a length the ranks cannot split evenly: falls back to the all-reduce
    hook(odd)
    assert torch.allclose(odd, torch.full((7,), sum(range(1, world + 1)) / world))
    m = torch.zeros_like(flat)
    for _ in range(3):
        p = flat.clone().requires_grad_(True)
        pred = X[b:e] @ p[:18].view(6, 3) + p[18:]
        loss = ((pred - Y[b:e]) ** 2).mean() * 50.0          # big enough that clipping is active
        (g,) = torch.autograd.grad(loss, p)
        flat_grad = g.clone()
        if _ == 1:
            # the two-segment asynchronous form the split step runner uses (early part, then the rest)
            works = [hook.start(flat_grad[:16])]
            works.append(hook.start(flat_grad[16:]))
            for wk in works:
                hook.finish(wk)              # (gloo completes in start(); the handles are None)
        else:
            hook(flat_grad)
        _flat_sgd_step(flat, flat_grad, m)
    # what `bench.py --gpus N` reports before its timed steps: every exchange form timed on a bucket (MAX over ranks; a form
    # the backend refuses is named instead of raising) and who runs where
    probe = cdist.probe_collectives(hook, 64, torch.device("cpu"), iters=2)
    assert set(probe) == set(cdist.MODES) and all(isinstance(v, float) and v >= 0 for v in probe.values()), probe
    inv = cdist.rank_inventory(torch.device("cpu"))
    assert sorted(r["rank"] for r in inv) == list(range(world)) and all(r["backend"] == "gloo" for r in inv)
    np.save(os.path.join(out_dir, "flat_%d.npy" % rank), flat.numpy())
    assert cdist.max_over_ranks(rank + 1.0, torch.device("cpu")) == float(world)
    tdist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "rsag", "direct"])
def test_two_rank_dp_matches_single_process(tmp_path, mode):
    """Every form of the exchange (one all-reduce; reduce-scatter + all-gather; all-to-all + fixed-order local sum + all-gather,
    the direct form SURVEY 8(e) sized for the point-to-point xGMI links) gives bit-identical replicas and the single-process
    result."""
    from cape_amd import dist as cdist
    assert cdist.shard_range(10, 4, 0) == (0, 3) and cdist.shard_range(10, 4, 3) == (8, 10)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    flats = [np.load(os.path.join(str(tmp_path), "flat_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(flats[0], flats[1])                # replicas stay bit-identical
    # single-process reference on the full batch (mean of per-shard means == full mean for equal shards)
    torch.manual_seed(0)
    X, Y = torch.randn(8, 6), torch.randn(8, 3)
    flat = torch.randn(6 * 3 + 3)
    m = torch.zeros_like(flat)
    for _ in range(3):
        p = flat.clone().requires_grad_(True)
        loss = ((X @ p[:18].view(6, 3) + p[18:] - Y) ** 2).mean() * 50.0
        (g,) = torch.autograd.grad(loss, p)
        _flat_sgd_step(flat, g.clone(), m)
    assert np.allclose(flats[0], flat.numpy(), rtol=1e-5, atol=1e-6)
