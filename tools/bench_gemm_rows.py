"""Micro-benchmark of the plain-source contraction on the narrow / short-contraction launches (fine mesh levels, the
K = 6 single layer): per-launch time from a HIP-graph replay, with the kernel the plan query names and an output checksum.
(Written for the row-streaming kernel experiment, tools/ubench/experiments/gemm_rows.h; the CAPE_GEMM_ROWS switch only
exists in a build that includes it.)"""
import os, sys
import torch
sys.path.insert(0, '.')
from cape_amd import ops
from tools.bench_sparse import graphed

dev = torch.device('cuda:0')
# (name, N, Mo, source channels, F, dual, weights contraction-contiguous (data-gradient form))
CASES = [
    ("enc2 fwd   3445 [64,64]->64", 16, 3445, [64, 64], 64, False, False),
    ("aff6 fwd   3445 [64,64]->64 dual", 16, 3445, [64, 64], 64, True, False),
    ("aff8 fwd   6890 [32,32]->32 dual", 16, 6890, [32, 32], 32, True, False),
    ("aff7 Z     3445 [64]->96", 16, 3445, [64], 96, False, False),
    ("enc2 dX    3445 [64]->128", 16, 3445, [64], 128, False, True),
    ("aff8 dX    6890 [32,32]->64", 16, 6890, [32, 32], 64, False, True),
    ("1x1  fwd   1723 [128]->64", 16, 1723, [128], 64, False, False),
    ("K6   fwd   6890 [16]x6->32", 64, 6890, [16] * 6, 32, False, False),
    ("K6   dX    6890 [32]->96", 64, 6890, [32], 96, False, True),
]
print("CAPE_GEMM_ROWS=%s" % os.environ.get("CAPE_GEMM_ROWS", "(default)"))
for name, N, Mo, Cs, F, dual, kc in CASES:
    torch.manual_seed(0)
    xs = [torch.randn(N, Mo, c, device=dev) for c in Cs]
    Ktot = sum(Cs)
    if kc:      # W^T view: element (k, f) at f * Ktot + k
        W = 0.1 * torch.randn(F, Ktot, device=dev)
        ent, off = [], 0
        for x, c in zip(xs, Cs):
            ent.append(dict(x=x, csr=None, w=(W, off, 1, Ktot)))
            off += c
    else:
        W = 0.1 * torch.randn(Ktot, F, device=dev)
        ent, off = [], 0
        for x, c in zip(xs, Cs):
            ent.append(dict(x=x, csr=None, w=(W, off * F, F, 1)))
            off += c
    mask = None
    if dual:
        Wa = 0.1 * torch.randn(Cs[0], F, device=dev)
        ent[0]["w2"] = (Wa, 0, F, 1)
        mask = torch.empty((N, Mo, (F + 31) // 32), device=dev, dtype=torch.int32)
    y = ops.alloc_act(N, Mo, F, dev)
    ops.PLAN_LOG = set()
    ops.gconv_fwd(ent, y, mask=mask)
    plan = sorted(ops.PLAN_LOG)
    ops.PLAN_LOG = None
    t = graphed(lambda: ops.gconv_fwd(ent, y, mask=mask))
    byts = 4.0 * N * Mo * (Ktot + F)
    fam, bm, bn, layout, du = plan[0][1:6]
    print("%-36s %8.2f us  %6.0f GB/s  %6.1f TF  %-44s checksum %.6e" % (
        name, t * 1e6, byts / t / 1e9, 2.0 * N * Mo * Ktot * F / t / 1e12, ops.fwd_kernel_name(fam, bm, bn, layout, du),
        float(y[:, :, :F].double().abs().sum())))
