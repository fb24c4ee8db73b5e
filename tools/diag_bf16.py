#!/usr/bin/env python3
"""Diagnostic (GPU box): the bf16-storage kernels one by one against torch fp32 on the same bf16 inputs."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    from cape_amd import ops, _lib
    from cape_amd.graph import ConvOperators, HostCSR
    from cape_amd.load_data import load_graph_mtx
    L, D, U, p, L_d, D_d, U_d = load_graph_mtx(None, load_for_demo=True)
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    N, Ch, Fout, K = 4, 64, 64, 2
    dops = ops.DeviceConvOps(ConvOperators(L[0], K, pool=D[1]), dev)
    Mo, Mi = dops.Mo, dops.Mi
    for dt in (torch.float32, bf):
        print("==== dtype", dt)
        gy = torch.randn((N, Mo, Fout), generator=g).to(dev).to(dt)
        y = torch.randn((N, Mo, Fout), generator=g).to(dev).to(dt)
        W = (0.1 * torch.randn((Ch * K, Fout), generator=g)).to(dev)
        # 1. bwd_prep
        gya = ops.alloc_act(N, Mo, Fout, dev, dtype=dt); gya.copy_(gy)
        ya = ops.alloc_act(N, Mo, Fout, dev, dtype=dt); ya.copy_(y)
        dz, dbv, _, _ = ops.bwd_prep(gya, y=ya, act="leaky", want_bias=True)
        ref_dz = gy.float() * torch.where(y.float() > 0, 1.0, 0.2)
        print("bwd_prep dz", rel(dz, ref_dz), "dbias", rel(dbv, ref_dz.sum((0, 1))))
        # 2. G = dz W^T de-interleaved
        ChP = (Ch + 3) // 4 * 4
        Gall = ops.alloc_act(N, Mo, K * ChP, dev, dtype=dt)
        ops.gconv_fwd([dict(x=dz, csr=None, w=(W, 0, 1, Fout))], Gall, deinterleave=K, F=K * Ch)
        Gref = dz.float() @ W.t()                 # [N, Mo, Ch*K], column c*K + k
        for k in range(K):
            print("  G_%d" % k, rel(Gall[:, :, k * ChP:k * ChP + Ch], Gref[:, :, k::K]))
        # 3. dx = sum_k S_k^T G_k
        Gs = [Gall[:, :, k * ChP:k * ChP + Ch] for k in range(K)]
        dx = ops.spmm_multi(Gs, [dops.bwd[k] for k in range(K)], sum=True)
        ref = torch.zeros((N, Mi, Ch), device=dev)
        for k in range(K):
            h = dops.host.bwd[k]
            if h.identity:
                ref += Gs[k].float()
            else:
                S = torch.sparse_csr_tensor(torch.from_numpy(h.rowptr.astype(np.int64)), torch.from_numpy(h.colidx.astype(np.int64)),
                                            torch.from_numpy(h.vals), size=h.shape).to(dev)
                for n in range(N):
                    ref[n] += S @ Gs[k][n].float()
        print("spmm_multi sum", rel(dx, ref), tuple(dx.shape), dx.dtype, "ident", [dops.host.bwd[k].identity for k in range(K)])
        # separate mode
        outs = ops.spmm_multi([Gs[0], Gs[1]], [dops.bwd[1], dops.bwd[1]])
        h = dops.host.bwd[1]
        S = torch.sparse_csr_tensor(torch.from_numpy(h.rowptr.astype(np.int64)), torch.from_numpy(h.colidx.astype(np.int64)),
                                    torch.from_numpy(h.vals), size=h.shape).to(dev)
        print("spmm_multi separate", [rel(outs[i][0], S @ Gs[i][0].float()) for i in range(2)])
        # 4. dW
        x = torch.randn((N, Mo, Ch), generator=g).to(dev).to(dt)
        xa = ops.alloc_act(N, Mo, Ch, dev, dtype=dt); xa.copy_(x)
        dW = torch.empty((Ch, Fout), device=dev)
        ops.gconv_dw([dict(x=xa, csr=None, w=(dW, 0, Fout, 1))], dz)
        print("dW", rel(dW, torch.einsum('nrc,nrf->cf', x.float(), dz.float())))


if __name__ == "__main__":
    main()
