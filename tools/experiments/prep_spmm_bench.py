"""Times cape_bwd_prep_spmm against the two launches it replaces (cape_bwd_prep + cape_spmm) on the five affine-block shapes of
the benchmarked step (batch 16), plus (PS_NOSUMS=1) the fused launch without its condition sums (what the reductions cost).  The
eager timings printed here are launch-bound (~18 us per call from Python): read the KERNEL durations from a rocprofv3 trace,
    rocprofv3 --kernel-trace --stats -d /tmp/ps -o r -- python tools/experiments/prep_spmm_bench.py; python tools/rocpd_summary.py <db> out.txt"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cape_amd import ops                                    # noqa: E402
from cape_amd.graph import HostCSR                          # noqa: E402
from cape_amd.load_data import load_graph_mtx               # noqa: E402


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    L = load_graph_mtx(None, load_for_demo=True)[0]
    rng = np.random.default_rng(0)
    N = 16
    shapes = ((3, 256), (2, 128), (1, 64), (0, 32))
    if os.environ.get("PS_SHAPE"):
        shapes = (shapes[int(os.environ["PS_SHAPE"])],)
    for lvl, F in shapes:
        Lt = sp.csr_matrix(sp.csr_matrix(L[2 * lvl], dtype=np.float64) - sp.identity(L[2 * lvl].shape[0]))
        Mo = Lt.shape[0]
        csr = ops.DeviceCSR(HostCSR(sp.csr_matrix(Lt.T)), dev)
        g = torch.tensor(rng.standard_normal((N, Mo, F)), dtype=torch.float32, device=dev)
        mask = torch.tensor(rng.integers(-2 ** 31, 2 ** 31 - 1, size=(N, Mo, F // 32), dtype=np.int64).astype(np.int32), device=dev)
        rs = torch.tensor(rng.standard_normal((3, Mo)).astype(np.float32), device=dev)
        t_prep = timed(lambda: ops.bwd_prep(g, mask=mask, rowscale=rs, R=2, rg=2, joint=True))
        dz = ops.bwd_prep(g, mask=mask, rowscale=rs, R=2, rg=2, joint=True)[0]
        t_spmm = timed(lambda: ops.spmm(dz, csr))
        t_fused = timed(lambda: ops.bwd_prep_spmm(g, mask, csr, rowscale=rs, R=2, rg=2, joint=True)) if not os.environ.get("PS_NOSUMS") else float("nan")
        t_nosum = timed(lambda: ops.bwd_prep_spmm(g, mask, csr)) if os.environ.get("PS_NOSUMS") else float("nan")
        print("Mo %5d F %4d   bwd_prep %5.1f + spmm %5.1f = %5.1f us   fused %5.1f us (incl. its finalisation launch)   fused without sums %5.1f us"
              % (Mo, F, t_prep, t_spmm, t_prep + t_spmm, t_fused, t_nosum), flush=True)


if __name__ == "__main__":
    main()
