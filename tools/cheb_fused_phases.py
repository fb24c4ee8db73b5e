#!/usr/bin/env python3
"""Diagnostic (GPU box): s_memtime phase stamps of one workgroup of cheb_fused_fwd_kernel on BASELINE configs[1]."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cape_amd import ops, _lib                      # noqa: E402
from cape_amd.graph import ConvOperators            # noqa: E402
from cape_amd.load_data import load_graph_mtx       # noqa: E402

L = load_graph_mtx(None, True)[0]
dev = torch.device('cuda:0')
N, Cin, Fout, K = 64, 16, 32, 6
dops = ops.DeviceConvOps(ConvOperators(L[0], K), dev)
x = torch.randn(N, 6890, Cin, device=dev)
W = 0.1 * torch.randn(Cin * K, Fout, device=dev)
ts = torch.zeros(64, dtype=torch.int64, device=dev)
fn = _lib.lib.cape_cheb_fused_debug_timestamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
with torch.no_grad():
    for _ in range(2):
        ops.chebyshev5(x, W, dops)
    fn(C.c_void_p(ts.data_ptr()))
    ops.chebyshev5(x, W, dops)
    torch.cuda.synchronize()
    fn(None)
t = ts.cpu().tolist()
print("prologue (x rows, row entries) %d cycles" % (t[1] - t[0]))
prev = t[1]
for k in range(1, K):
    print("step %d: sparse %6d   contraction %6d   barrier %6d" % (k, t[2 + 3 * k] - prev, t[3 + 3 * k] - t[2 + 3 * k], t[4 + 3 * k] - t[3 + 3 * k]))
    prev = t[4 + 3 * k]
print("last contraction %d, epilogue stores %d, total %d cycles" % (t[30] - prev, t[31] - t[30], t[31] - t[0]))
# backward: the dx kernel (Clenshaw) stamps 31 .. 62
xg = x.clone().requires_grad_(True)
Wg = W.clone().requires_grad_(True)
dy = torch.randn(N, 6890, Fout, device=dev)
for _ in range(2):
    y = ops.chebyshev5(xg, Wg, dops)
    torch.autograd.grad(y, [xg, Wg], dy)
ts.zero_()
fn(C.c_void_p(ts.data_ptr()))
y = ops.chebyshev5(xg, Wg, dops)
torch.autograd.grad(y, [xg, Wg], dy)
torch.cuda.synchronize()
fn(None)
t = ts.cpu().tolist()
print("dx kernel: G_%d tiles %d cycles, barrier %d" % (K - 1, t[33] - t[32], t[34] - t[33]))
for k in range(K - 2, 0, -1):
    print("  k = %d: G tiles %6d   barrier %6d   sparse %6d   barrier %6d" % (k, t[35 + 4 * k] - (t[38 + 4 * (k + 1)] if k < K - 2 else t[34]),
                                                                          t[36 + 4 * k] - t[35 + 4 * k], t[37 + 4 * k] - t[36 + 4 * k], t[38 + 4 * k] - t[37 + 4 * k]))
print("  tail (G_0, last sparse, dx stores) %d cycles; kernel total from row load %d" % (t[62] - t[38 + 4], t[62] - t[31]))
plan = dops.patch_plan(Cin, Fout).host
print("patches %d, rmax %d, ring sizes of patch 9: %s" % (plan.P, plan.rmax, plan.pinfo[9 % plan.P, 3:3 + K].tolist()))
