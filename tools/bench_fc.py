"""Micro-benchmark of the dense-layer kernels (csrc/fc.hip) at the nz64 model's shapes: batch 16, encoder
fc_mean / fc_logvar 55168 -> 64 (x2), decoder fc1 128 -> 55168.  Time per call from a HIP-graph replay of 20 calls;
28 MB of fp32 weights stream per pass."""
import sys
import torch
sys.path.insert(0, '.')
from cape_amd import ops
from tools.bench_sparse import graphed

dev = torch.device('cuda:0')
N, LONG, NZ, WIN = 16, 55168, 64, 128
x = torch.randn(N, LONG, device=dev)
Ws = [(0.01 * torch.randn(LONG, NZ, device=dev)).requires_grad_(True) for _ in range(2)]
bs = [torch.zeros(NZ, device=dev, requires_grad=True) for _ in range(2)]
gs = [torch.randn(N, NZ, device=dev) for _ in range(2)]
xr = x.clone().requires_grad_(True)


def long_fwd():
    with torch.no_grad():
        ops.FcLongFn.apply(x, None, Ws[0], bs[0], Ws[1], bs[1])


import ctypes as C
lib, check, _parr, _ptr, _stream = ops.lib, ops.check, ops._parr, ops._ptr, ops._stream
dWs = [torch.empty_like(W) for W in Ws]
dbs = [torch.empty(NZ, device=dev) for _ in Ws]
dxl = torch.empty_like(x)


def long_bwd():
    check(lib.cape_fc_long_bwd(C.c_void_p(x.data_ptr()), LONG, N, LONG, NZ, 2, _parr([W.detach() for W in Ws]), _parr(gs), _parr(dWs),
                               _parr(dbs), _ptr(dxl), LONG, _stream()), "cape_fc_long_bwd")


z = torch.randn(N, WIN, device=dev, requires_grad=True)
Ww = (0.01 * torch.randn(WIN, LONG, device=dev)).requires_grad_(True)
bw = torch.zeros(LONG, device=dev, requires_grad=True)
gw = torch.randn(N, LONG, device=dev)


def wide_fwd():
    with torch.no_grad():
        ops.FcWideFn.apply(z, Ww, bw, "leaky", None, None)


with torch.no_grad():
    yw = ops.FcWideFn.apply(z, Ww, bw, "leaky", None, None)
dWw, dbw, dz_ = torch.empty_like(Ww), torch.empty(LONG, device=dev), torch.empty(N, WIN, device=dev)
need = int(lib.cape_fc_wide_bwd_workspace_bytes(N, WIN, LONG))
wsw = torch.empty((need + 3) // 4, device=dev)


def wide_bwd():
    check(lib.cape_fc_wide_bwd(C.c_void_p(z.data_ptr()), WIN, C.c_void_p(gw.data_ptr()), LONG, C.c_void_p(yw.data_ptr()), LONG,
                               ops._lib.ACT["leaky"], N, WIN, LONG, C.c_void_p(Ww.data_ptr()), _ptr(dWw), _ptr(dbw), _ptr(dz_), WIN,
                               _ptr(wsw), need, _stream()), "cape_fc_wide_bwd")


MB = 4 * LONG * NZ * 2 / 1e6
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, fn, mb in (("fc_long fwd (2 launches)", long_fwd, MB), ("fc_long bwd (1 launch)", long_bwd, 2 * MB),
                     ("fc_wide fwd (1 launch)", wide_fwd, MB), ("fc_wide bwd (3 launches)", wide_bwd, 2 * MB)):
    if only and only not in name:
        continue
    t = graphed(fn)
    print("%-28s %8.2f us   %6.0f GB/s of weight traffic" % (name, t * 1e6, mb / t / 1e3))
