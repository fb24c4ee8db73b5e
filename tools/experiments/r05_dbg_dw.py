import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cape_amd import ops
DEV = "cuda:0"
def _act(a):
    t = ops.alloc_act(a.shape[0], a.shape[1], a.shape[2], torch.device(DEV))
    t.copy_(torch.from_numpy(a)); return t
for case in [(2, 203, [64], 72), (16, 862, [256, 256], 512), (5, 330, [128, 64], 132), (16, 1723, [128, 128], 128), (16, 862, [256], 512), (4, 862, [256], 128), (16, 431, [128], 128)]:
    N, Mo, Cs, F = case
    rng = np.random.default_rng(Mo + F)
    sc = 2.0 ** (-rng.integers(0, 12, (N, Mo, 1))).astype(np.float64)
    dz = (rng.standard_normal((N, Mo, F)) * sc).astype(np.float32)
    hdz = _act(dz); ops.rowmax(hdz)
    ent, want = [], []
    for C_ in Cs:
        x = (rng.standard_normal((N, Mo, C_)) * sc[::-1]).astype(np.float32)
        hx = _act(x); ops.rowmax(hx)
        dW = torch.zeros((C_, F), device=DEV)
        ent.append(dict(x=hx, csr=None, w=(dW, 0, F, 1)))
        want.append(np.einsum('nrc,nrf->cf', x.astype(np.float64), dz.astype(np.float64)))
    ops.PLAN_LOG = set()
    ops.gconv_dw(ent, hdz); plans = set(ops.PLAN_LOG); ops.PLAN_LOG = None
    for e, w in zip(ent, want):
        got = e["w"][0].cpu().numpy().astype(np.float64)
        err = np.abs(got - w)
        i = np.unravel_index(err.argmax(), err.shape)
        print(case, plans, "rel err %.3e at %s got %.6g want %.6g; nan %d; rows bad %d cols bad %d" % (err.max() / np.abs(w).max(), i, got[i], w[i], np.isnan(got).sum(), (err.max(1) > 1e-5 * np.abs(w).max()).sum(), (err.max(0) > 1e-5 * np.abs(w).max()).sum()))
