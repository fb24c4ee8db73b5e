"""Per-layer fwd+bwd time of the conv autograd op in 'fused' vs 'twopass' mode (MI355X)."""
import sys
import torch
sys.path.insert(0, '.')
from cape_amd import ops
from cape_amd.graph import ConvOperators
from cape_amd.load_data import load_graph_mtx
from tools.bench_gconv import timeit

L, D, U, p, Ld, Dd, Ud = load_graph_mtx(None, True)
dev = torch.device('cuda:0')
N = 16
# name, L level, Ch, Fout, K, pool idx, unpool idx, affine, cond_in
layers = [("enc1", 0, 3, 64, 2, 0, None, False, 0), ("enc2", 1, 64, 64, 2, 1, None, False, 0),
          ("enc3", 2, 64, 128, 2, 2, None, False, 0), ("enc4", 3, 128, 128, 2, 3, None, False, 0),
          ("enc5", 4, 128, 256, 2, 4, None, False, 0), ("enc6", 5, 256, 256, 2, 5, None, False, 0),
          ("enc7", 6, 256, 512, 2, 6, None, False, 0), ("enc8", 7, 512, 512, 2, 7, None, False, 0),
          ("aff1", 7, 512, 256, 2, None, 7, True, 64), ("aff2", 6, 256, 256, 2, None, 6, True, 64),
          ("aff3", 5, 256, 128, 2, None, 5, True, 64), ("aff4", 4, 128, 128, 2, None, 4, True, 64),
          ("aff5", 3, 128, 64, 2, None, 3, True, 64), ("aff6", 2, 64, 64, 2, None, 2, True, 64),
          ("aff7", 1, 64, 32, 2, None, 1, True, 64), ("aff8", 0, 32, 32, 2, None, 0, True, 64),
          ("out", 0, 32, 3, 2, None, None, False, 64)]
tot = {"fused": 0.0, "twopass": 0.0, "best": 0.0}
for name, lvl, Ch, F, K, pi, ui, aff, Cc in layers:
    host = ConvOperators(L[lvl], K, unpool=U[ui] if ui is not None else None, pool=D[pi] if pi is not None else None)
    dops = ops.DeviceConvOps(host, dev)
    x = torch.randn(N, dops.Mi, Ch, device=dev, requires_grad=True)
    W = (torch.randn((Ch + Cc) * K, F, device=dev) * 0.1).requires_grad_(True)
    Wa = (torch.randn(Ch + Cc, F, device=dev) * 0.1).requires_grad_(True) if aff else None
    b = None if aff else torch.zeros(1, 1, F, device=dev, requires_grad=True)
    cin = torch.randn(N, Cc, device=dev, requires_grad=True) if Cc else None
    g = torch.randn(N, dops.Mo, F, device=dev)
    res = {}
    for mode in ("fused", "twopass"):
        ops.MODE = mode
        def step():
            y = ops.chebyshev5(x, W, dops, bias=b, activation=None if aff else "b1leakyrelu", W_affine=Wa, cond_in=cin)
            torch.autograd.grad(y, [t for t in (x, W, Wa, b, cin) if t is not None], g)
        # GPU time only: capture the layer's fwd+bwd into a HIP graph and time replays
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2): step()
        torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        res[mode] = timeit(gr.replay, iters=20) * 1e6
        tot[mode] += res[mode]
    tot["best"] += min(res.values())
    fl = 3 * 2.0 * N * dops.Mo * Ch * F * (K + (1 if aff else 0))
    print("%-5s Mi%5d Mo%5d %4d->%3d  fused %7.1f us  twopass %7.1f us   (%.1f TF at best)" % (
        name, dops.Mi, dops.Mo, Ch, F, res["fused"], res["twopass"], fl / min(res.values()) / 1e6))
print("total fused %.2f ms, twopass %.2f ms, per-layer best %.2f ms" % (tot["fused"] / 1e3, tot["twopass"] / 1e3, tot["best"] / 1e3))
