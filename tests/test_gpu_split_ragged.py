"""Ragged shapes for the bf16-split GEMM (cape_amd/csrc/gemm_split.h) through the C-ABI entry cape_gconv_fwd: row and
column counts that are not tile multiples, several sources of different widths, both weight layouts, bias +
activation and the de-interleaving epilogue, against float64 numpy.  The library must report the split family for
every case (cape_gconv_fwd_plan).

Includes the 128 x 128 tile instantiations the library selects at the benchmarked batch (N = 16)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # N, Mo, [C per source], F, layout ('nc' output-contiguous | 'kc' contraction-contiguous), act, bias, deinterleave
    (1, 37, [32], 64, 'nc', 'leaky', True, 0),
    (3, 203, [64, 96], 72, 'nc', None, False, 0),
    (2, 131, [128], 200, 'kc', None, False, 0),
    (5, 130, [64, 32, 64], 132, 'kc', 'relu', True, 0),
    (16, 260, [256], 384, 'nc', 'leaky', True, 0),          # 128 x 128 tiles (16*3*3 = 144 < 384 -> still 64 x 64)
    (16, 1000, [128, 128], 512, 'kc', None, False, 0),           # 16*8*4 = 512 tiles of 128 x 128
    (16, 1000, [128], 256, 'nc', None, False, 2),                # data-gradient form: K = 2 orders de-interleaved
    (8, 777, [96], 192, 'kc', None, False, 3),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_Mo%d_C%s_F%d_%s" % (c[0], c[1], "+".join(map(str, c[2])), c[3], c[4]))
def test_split_gemm_ragged(case):
    from cape_amd import _lib, ops
    N, Mo, Cs, F, layout, act, with_bias, deint = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(Mo * 100 + F)
    ref = np.zeros((N, Mo, F))
    entries, keep = [], []
    for C_ in Cs:
        x = rng.standard_normal((N, Mo, C_))
        hx = ops.alloc_act(N, Mo, C_, dev)
        hx.copy_(torch.tensor(x, dtype=torch.float32))
        W = rng.standard_normal((C_, F)) * 0.3
        if layout == 'nc':
            hW = torch.tensor(W, dtype=torch.float32, device=dev)
            w = (hW, 0, F, 1)
        else:
            hW = torch.tensor(np.ascontiguousarray(W.T), dtype=torch.float32, device=dev)      # [F, C]
            w = (hW, 0, 1, C_)
        ref += x.astype(np.float32).astype(np.float64) @ W.astype(np.float32).astype(np.float64)
        entries.append(dict(x=hx, csr=None, w=w))
        keep += [hx, hW]
    bias = None
    if with_bias:
        b = rng.standard_normal(F)
        bias = torch.tensor(b, dtype=torch.float32, device=dev)
        ref = ref + b.astype(np.float32)
    if act == 'leaky':
        ref = np.where(ref > 0, ref, 0.2 * ref)
    elif act == 'relu':
        ref = np.maximum(ref, 0)

    # the library must pick the split family for these arguments
    plan = (C.c_int32 * 4)()
    assert _lib.lib.cape_gconv_fwd_plan(ops._mk_srcs(entries), len(entries), N, Mo, F, plan) == 0
    assert plan[0] == 2 and plan[3] == (1 if layout == 'kc' else 0), list(plan)
    assert (plan[1], plan[2]) == ((128, 128) if N * ((Mo + 127) // 128) * ((F + 127) // 128) >= 384 else (64, 64))

    if deint:
        stride = (F // deint + 3) // 4 * 4
        y = torch.zeros((N, Mo, deint * stride), device=dev)
        ops.gconv_fwd(entries, y, bias=bias, bias_mode=_lib.BIAS_CHANNEL, act=act or "none", deinterleave=deint, F=F)
        got = y.cpu().numpy().astype(np.float64)
        want = np.zeros_like(got)
        for j in range(F):                                    # column j = c*K + k  ->  channel k*stride + c
            want[:, :, (j % deint) * stride + j // deint] = ref[:, :, j]
    else:
        y = ops.alloc_act(N, Mo, F, dev)
        ops.gconv_fwd(entries, y, bias=bias, bias_mode=_lib.BIAS_CHANNEL, act=act or "none")
        got, want = y.cpu().numpy().astype(np.float64), ref
    scale = np.sqrt((want ** 2).sum(-1)).max()
    err = np.sqrt(((got - want) ** 2).sum(-1)).max() / scale
    assert err < 2e-5, err
