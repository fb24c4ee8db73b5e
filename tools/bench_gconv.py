"""Micro-benchmark of the gather-GEMM kernel variants on representative layer shapes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from cape_amd import ops
from cape_amd.graph import ConvOperators
from cape_amd.load_data import load_graph_mtx

L, D, U, p, Ld, Dd, Ud = load_graph_mtx(None, True)
dev = torch.device('cuda:0')
N = 16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


cases = [] if __name__ != "__main__" else [  # name, level, Cin, Fout, K, unpool idx, affine
    ("enc2 6890 64->64 K2", 1, 64, 64, 2, None, False),
    ("enc4 3445 128->128 K2", 3, 128, 128, 2, None, False),
    ("enc6 1723 256->256 K2", 5, 256, 256, 2, None, False),
    ("enc8 862 512->512 K2", 7, 512, 512, 2, None, False),
    ("enc8 plain K1 1024->512", 7, 1024, 512, 1, None, False),
    ("aff1 862 576->256", 7, 576, 256, 2, 7, True),
    ("aff5 3445 192->64 U", 3, 192, 64, 2, 3, True),
    ("aff7 6890 128->32 U", 1, 128, 32, 2, 1, True),
    ("aff8 6890 96->32", 0, 96, 32, 2, 0, True),
]
for name, lvl, Cin, Fout, K, ui, aff in cases:
    Lm = L[lvl]
    Um = U[ui] if ui is not None else None
    host = ConvOperators(Lm, K, unpool=Um)
    dops = ops.DeviceConvOps(host, dev)
    x = torch.randn(N, dops.Mi, Cin, device=dev)
    W = torch.randn(Cin * K, Fout, device=dev) * 0.1
    Wa = torch.randn(Cin, Fout, device=dev) * 0.1 if aff else None
    y = ops.alloc_act(N, dops.Mo, Fout, dev)
    ent = []
    for k in range(K):
        e = dict(x=x, csr=dops.fwd[k], w=(W, k * Fout, K * Fout, 1))
        if aff and k == 0:
            e["w2"] = (Wa, 0, Fout, 1)
        ent.append(e)
    mask = torch.empty((N, dops.Mo, (Fout + 31) // 32), device=dev, dtype=torch.int32) if aff else None
    t = timeit(lambda: ops.gconv_fwd(ent, y, mask=mask))
    fl = 2.0 * N * dops.Mo * Cin * Fout * (K + (1 if aff else 0))
    nnz = [h.nnz for h in host.fwd]
    # dX-like launch (transposed weights) and dW
    dz = torch.randn(N, dops.Mo, Fout, device=dev)
    dx = ops.alloc_act(N, dops.Mi, Cin, dev)
    entb = [dict(x=dz, csr=dops.bwd[k], w=(W, k * Fout, 1, K * Fout)) for k in range(K)]
    tb = timeit(lambda: ops.gconv_fwd(entb, dx))
    dW = torch.empty_like(W)
    entw = [dict(x=x, csr=dops.fwd[k], w=(dW, k * Fout, K * Fout, 1)) for k in range(K)]
    tw = timeit(lambda: ops.gconv_dw(entw, dz))
    flb = 2.0 * N * dops.Mo * Cin * Fout * K
    print("%-26s fwd %7.1f us %6.1f TF | dX %7.1f us %6.1f TF | dW %7.1f us %6.1f TF | nnz/row %s" % (
        name, t * 1e6, fl / t / 1e12, tb * 1e6, flb / tb / 1e12, tw * 1e6, flb / tw / 1e12,
        [round(z / dops.Mo, 1) for z in nnz]))
