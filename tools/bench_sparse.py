"""Micro-benchmark of the sparse operator application (spmm) on the layer shapes of the nz64 model: time per launch
(HIP-graph replay of 20 launches), algorithmic GB/s, and a same-size device copy as the streaming floor.
Run once per CAPE_SPMM_UNROLL value to compare the entry-loop variants:
    for u in 0 4 8; do CAPE_SPMM_UNROLL=$u python tools/bench_sparse.py; done"""
import os, sys
import torch
sys.path.insert(0, '.')
from cape_amd import ops
from cape_amd.graph import ConvOperators
from cape_amd.load_data import load_graph_mtx
from tools.bench_gconv import timeit

L, D, U, p, Ld, Dd, Ud = load_graph_mtx(None, True)
dev = torch.device('cuda:0')
N = 16
REP = 20


def graphed(fn):
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    return timeit(g.replay, iters=20) / REP


if __name__ == "__main__":
    print("CAPE_SPMM_UNROLL=%s" % os.environ.get("CAPE_SPMM_UNROLL", "(default)"))
    print("%-28s %6s %9s %9s %9s %9s" % ("shape", "dtype", "spmm us", "GB/s", "copy us", "axpy us"))
    for lvl, C in ((0, 32), (0, 64), (2, 64), (2, 128), (4, 128), (4, 256), (6, 256), (6, 512)):
        dops = ops.DeviceConvOps(ConvOperators(L[lvl], 2), dev)
        csr = dops.fwd[1]
        M = dops.Mo
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(N, M, C, device=dev).to(dt)
            z = torch.randn(N, M, C, device=dev).to(dt)
            y = ops.alloc_act(N, M, C, dev, dtype=dt)
            t = graphed(lambda: ops.spmm(x, csr, y=y, alpha=2.0, z=z, beta=-1.0))
            tc = graphed(lambda: y.copy_(x))
            ta = graphed(lambda: torch.add(x, z, alpha=-1.0, out=y))
            byts = x.element_size() * 3 * N * M * C
            print("%-28s %6s %9.2f %9.0f %9.2f %9.2f" % ("L~ %d x %d x %d" % (N, M, C), str(dt).split('.')[-1][:4], t * 1e6,
                                                         byts / t / 1e9, tc * 1e6, ta * 1e6))

    print("%-28s %6s %9s %9s %9s" % ("bwd_prep shape", "dtype", "mask+R2", "leaky+b", "GB/s(m)"))
    for lvl, F in ((0, 32), (0, 64), (2, 64), (2, 128), (4, 128), (4, 256), (6, 256), (6, 512)):
        M = L[lvl].shape[0]
        rowscale = torch.randn(3, M, device=dev)
        for dt in (torch.float32, torch.bfloat16):
            g = torch.randn(N, M, F, device=dev).to(dt)
            yy = torch.randn(N, M, F, device=dev).to(dt)
            mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, M, (F + 31) // 32), device=dev, dtype=torch.int32)
            ops.DEFERRED = []          # main pass only: the partial-sum finals run batched at the end of the backward pass
            tm = graphed(lambda: ops.bwd_prep(g, mask=mask, rowscale=rowscale, R=2, rg=2, joint=True, defer=True))
            tl = graphed(lambda: ops.bwd_prep(g, y=yy, act="leaky", want_bias=True, defer=True))
            ops.DEFERRED = None
            byts = g.element_size() * 2 * N * M * F
            print("%-28s %6s %9.2f %9.2f %9.0f" % ("%d x %d x %d" % (N, M, F), str(dt).split('.')[-1][:4], tm * 1e6, tl * 1e6, byts / tm / 1e9))
