"""How much of cape_spmm's time is the gather traffic through L2 -> CU?  Same kernel, same number of entries per row, three
operators on 16 x M x C: the SMPL Laplacian (shipped vertex order), a BANDED operator (row r gathers rows r-3 .. r+3: the
work items of a wave then share their gathered lines in the CU's vector cache), and the identity-like operator that gathers
ONE row seven times (no gather traffic at all beyond the row itself).  The gap between the first and the others is what an
LDS-staged patch form of the operator application could recover.
    gpurun -- 'python tools/experiments/spmm_locality_bound.py'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
from cape_amd import ops
from cape_amd.graph import HostCSR
from cape_amd.load_data import load_graph_mtx
from spmm_reorder import timeit

L = load_graph_mtx(None, load_for_demo=True)[0]
dev = torch.device('cuda:0')
for lvl, Cn in ((0, 32), (0, 64), (0, 128), (2, 128), (2, 256), (4, 256), (6, 512)):
    A = sp.csr_matrix(L[lvl], dtype=np.float64)
    M = A.shape[0]
    rows = np.repeat(np.arange(M), 7)
    band = sp.csr_matrix((np.full(7 * M, 0.1), (rows, np.clip(rows + np.tile(np.arange(-3, 4), M), 0, M - 1))), shape=(M, M))
    band.sum_duplicates(); band.sort_indices()
    same = sp.csr_matrix((np.full(7 * M, 0.1), (rows, rows)), shape=(M, M))       # duplicates summed: 1 entry per row
    xa = ops.alloc_act(16, M, Cn, dev); xa.copy_(torch.randn(16, M, Cn, device=dev))
    res = []
    for name, mat in (("laplacian", A), ("banded", band), ("diagonal", same)):
        c = ops.DeviceCSR(HostCSR(mat), dev)
        y = ops.spmm(xa, c)
        res.append("%s %5.1f us" % (name, timeit(lambda: ops.spmm(xa, c, y=y))))
    nbytes = 2 * 16 * M * Cn * 4
    res.append("copy-sized torch.add %5.1f us" % timeit(lambda: torch.add(xa, 1.0, out=y)))
    print("level %d M=%5d C=%3d (in+out %5.1f MB)  " % (lvl, M, Cn, nbytes / 1e6) + "  ".join(res))
