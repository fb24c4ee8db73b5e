#!/usr/bin/env python3
"""One rank of the data-parallel equivalence check (tests/test_gpu_dist.py): the REAL model through the graph runner with
the gradient exchange hook, each rank on its shard of a global batch.  Launched by torch.distributed.run; with one visible
GPU every rank uses device 0 and the gloo transport (CAPE_FORCE_DEVICE=0 / backend from the command line), on a multi-GPU
node rank r uses device r and RCCL.
    python -m torch.distributed.run --nproc-per-node 2 tools/dp_equiv_worker.py OUT.npz B STEPS GAN BACKEND"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(batch, device):
    from cape_amd.configs import cape_params
    from cape_amd.load_data import load_graph_mtx
    from cape_amd.models import CAPE
    L, D, U, p, L_d, D_d, _ = load_graph_mtx(None, load_for_demo=True)
    params = cape_params('CAPE-affineconv_nz64_pose32_clotype32_male', p=p, batch_size=batch, name='dp_equiv', decay_steps=1000,
                         lr_warmup=0, lr=2e-3)
    model = CAPE(L=L, D=D, U=U, L_d=L_d, D_d=D_d, device=device, **params)
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='train')
    return model


def global_batch(total, nz, seed=7):
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(total, 6890, 3)
    clo = torch.eye(4)[torch.arange(total) % 4]
    return dict(data_g=x, gt=x + 0.1 * r(total, 6890, 3), data_d=r(total, 6890, 3), cond_g=0.5 * r(total, 126),
                cond_d=0.5 * r(total, 126), cond2_g=clo, cond2_d=clo.roll(1, 0), eps=r(total, nz))


def run(model, batch, steps, gan, hook, graph=True):
    from cape_amd.runtime import GraphedTrainStep
    runner = GraphedTrainStep(model, with_gan=gan, grad_hook=hook, use_graph=graph)
    runner.load_batch(**batch)
    torch.cuda.synchronize()
    runner.capture(preserve_state=True)
    for _ in range(steps):
        runner.step()
    torch.cuda.synchronize()
    return runner


def main():
    out, B, steps, gan, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == '1', sys.argv[5]
    from cape_amd import dist as cdist
    world, rank, local = cdist.init_from_env(backend=backend)
    dev = int(os.environ.get("CAPE_FORCE_DEVICE", local))
    torch.cuda.set_device(dev)
    model = build(B, 'cuda:%d' % dev)
    model.sync_variables(src=0)
    gb = global_batch(B * world, int(model.nz))
    shard = {k: v[rank * B:(rank + 1) * B] for k, v in gb.items()}
    runner = run(model, shard, steps, gan, cdist.GradAverager() if world > 1 else None)
    flats = {grp: model._opt_state[grp]['flat'].detach().cpu().numpy() for grp in (('g', 'd') if gan else ('g',))}
    if world > 1:
        import torch.distributed as tdist
        # replicas must stay bit-identical: compare checksums across ranks
        for grp, f in flats.items():
            t = torch.tensor([float(np.abs(f).sum()), float(f.sum())], dtype=torch.float64, device='cuda:%d' % dev)
            lo, hi = t.clone(), t.clone()
            tdist.all_reduce(lo, op=tdist.ReduceOp.MIN)
            tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)
            assert torch.equal(lo, hi), ("replicas diverged", grp)
    if rank == 0:
        np.savez(out, split=np.asarray(int(runner.split)), **{"flat_" + k: v for k, v in flats.items()})
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
