#!/usr/bin/env python3
"""Diagnostic (GPU box): where the forward error of the bf16-storage model comes from.  CAPE-affineconv_nz64 at batch N on the
golden inputs, prediction error (relative L2 / worst vertex) against the fp64 twin for
  a) bf16 storage as shipped (fp32 master weights rounded to bf16 inside the kernels),
  b) the same with the conv weights PRE-ROUNDED to bf16 on both sides (what is left is activation rounding only),
  c) fp32 storage (the parity path) for scale."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import test_gpu_model as T
    from cape_amd.load_data import load_graph_mtx, load_pack
    L, D, U, p, L_d, D_d, U_d = load_graph_mtx(None, load_for_demo=True)
    mesh_ops = dict(L=L, D=D, U=U, p=p, L_d=L_d, D_d=D_d, U_d=U_d, pack=load_pack())
    x, gt, xd, cond, cond_d, clo, clo_d, eps = T._inputs(N, 64)

    def err(pred, ref):
        pred, ref = np.asarray(pred, np.float64), np.asarray(ref, np.float64)
        return float(np.sqrt(((pred - ref) ** 2).sum() / (ref ** 2).sum())), T.vertex_err(pred, ref)

    for label, act_dtype, round_w in (("a) bf16 storage, fp32 master weights", 'bf16', False),
                                      ("b) bf16 storage, conv weights pre-rounded to bf16 on both sides", 'bf16', True),
                                      ("c) fp32 storage", 'fp32', False)):
        P, twin, model = T._build("affine_nz64", mesh_ops, N, dict(act_dtype=act_dtype))
        T._run_twin(twin, x[:1], gt[:1], xd[:1], cond[:1], cond_d[:1], clo[:1], clo_d[:1], eps[:1])       # materialise variables
        if round_w:
            with torch.no_grad():
                for n, v in twin.params.items():
                    if twin.vs.kinds[n] == 'conv':
                        v.copy_(v.to(torch.bfloat16).to(v.dtype))
                        twin.vs.vars[n][...] = v.numpy()
        xh, zm, zl, _, _, ls = T._run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps)
        model.load_variables(twin.vs.vars)
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
        with torch.no_grad():
            out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
        l2, wv = err(out['prediction'].cpu().numpy(), xh.detach().numpy())
        print("%-70s prediction rel L2 %.3e  worst vertex %.3e   z_mean %.2e  z_logvar %.2e" % (
            label, l2, wv, T.rel_err(out['z_mean'].cpu().numpy(), zm.detach().numpy()),
            T.rel_err(out['z_logvar'].cpu().numpy(), zl.detach().numpy())))
        # decoder alone from the twin's latent code: the decoder's own error, without the encoder's
        zt = np.concatenate([zm.detach().numpy(), ] + [a.detach().numpy() for a in twin.cond_embeddings(cond, clo)], 1)
        y, y2 = twin.cond_embeddings(cond, clo)
        ref_dec = twin.decoder_cond_vert(zt, y, y2).detach().numpy()
        with torch.no_grad():
            with model.variable_scope('generator'):
                dec = model.decoder_cond_vert(t(zt), t(y.detach().numpy()), t(y2.detach().numpy()), use_res_block=model.use_res_block_dec)
        l2d, wvd = err(dec.cpu().numpy(), ref_dec)
        print("%-70s decoder only (exact latent code): rel L2 %.3e  worst vertex %.3e" % ("", l2d, wvd))
        del model, twin


if __name__ == "__main__":
    main()
