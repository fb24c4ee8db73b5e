#!/usr/bin/env python3
"""Diagnostic (GPU box): per-variable gradient error of the HIP model against the fp64 twin at a given batch, plus the
gradients w.r.t. the latent heads -- to localise where an error enters the backward pass.
    python tools/diag_grad_parity.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(N):
    from cape_amd.load_data import load_graph_mtx, load_pack
    import test_gpu_model as T
    L, D, U, p, L_d, D_d, U_d = load_graph_mtx(None, load_for_demo=True)
    mesh_ops = dict(L=L, D=D, U=U, p=p, L_d=L_d, D_d=D_d, U_d=U_d, pack=load_pack())
    P, twin, model = T._build("affine_nz64", mesh_ops, N)
    x, gt, xd, cond, cond_d, clo, clo_d, eps = T._inputs(N, P["nz"])
    xh, zm, zl, d_real, d_fake, ls = T._run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps)
    model.load_variables(twin.vs.vars)
    dev = model.device
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
    print("forward: prediction vertex err %.2e  z_mean %.2e  z_logvar %.2e" % (
        T.vertex_err(out['prediction'].detach().cpu().numpy(), xh.detach().numpy()),
        T.rel_err(out['z_mean'].detach().cpu().numpy(), zm.detach().numpy()),
        T.rel_err(out['z_logvar'].detach().cpu().numpy(), zl.detach().numpy())))
    names = model._g_names
    tg = torch.autograd.grad(ls['loss_g'], [twin.params[n] for n in names] + [zm, zl], retain_graph=True, allow_unused=True)
    hg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in names] + [out['z_mean'], out['z_logvar']],
                             retain_graph=True, allow_unused=True)
    for n, a, b in zip(names + ['<z_mean>', '<z_logvar>'], tg, hg):
        if a is None:
            continue
        a64, b64 = a.numpy(), b.cpu().numpy().astype(np.float64)
        print("%-62s |g| %.3e  L2 err/|g| %.2e  max-norm %.2e" % (n, np.sqrt((a64 ** 2).sum()),
              np.sqrt(((b64 - a64) ** 2).sum() / max((a64 ** 2).sum(), 1e-300)), T.rel_err(b64, a64)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
