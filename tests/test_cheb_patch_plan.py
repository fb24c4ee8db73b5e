"""Host logic of the on-chip Chebyshev recurrence (cape_amd.graph.ChebPatchPlan, consumed by csrc/cheb_fused.hip): the
patches partition the vertices, the halo rings are what the K-step recurrence needs, and the patch-local algorithm the
kernel runs -- forward recurrence on shrinking rings, contraction on the patch; Clenshaw for the adjoint, weight gradient
from the recomputed recurrence -- restated here in numpy per patch, equals the dense evaluation of reference
lib/models.py:69-103 and its gradients."""
import numpy as np
import scipy.sparse as sp


def _emulate(plan, x, W, dy, K, Cin, Fout):
    """Per-patch evaluation exactly as the kernel organises it (float64)."""
    M = plan.M
    y = np.zeros((M, Fout))
    dx = np.zeros((M, Cin))
    dW = np.zeros((Cin * K, Fout))
    Wk = [W[k::K] for k in range(K)]                      # W_k [Cin, Fout]: rows c*K + k
    for p in range(plan.P):
        v0, e0 = int(plan.pinfo[p, 0]), int(plan.pinfo[p, 2])
        r0 = int(plan.csr_rowptr_off[p])
        R = [int(v) for v in plan.pinfo[p, 3:3 + K]]
        vid = plan.vid[v0:v0 + R[-1]]
        nrow = R[max(K - 2, 0)]
        rp = plan.rowptr[r0:r0 + nrow + 1]
        A = sp.csr_matrix((plan.val[e0:e0 + rp[-1]].astype(np.float64), plan.lcol[e0:e0 + rp[-1]], rp), shape=(nrow, R[-1]))
        # the ELL rows the kernel reads describe the same matrix (padding = (own row, 0))
        er = int(plan.pinfo[p, 1])
        ec, ev = plan.ell_col[er:er + nrow], plan.ell_val[er:er + nrow]
        E = sp.csr_matrix((ev.ravel().astype(np.float64), (np.repeat(np.arange(nrow), ec.shape[1]), ec.ravel())), shape=(nrow, R[-1]))
        assert abs(E - A).max() == 0.0 and plan.ell_col.shape[1] == 12
        own = R[0]
        # forward (and the recomputation of the backward's part 1)
        prev, other = x[vid].copy(), np.full((R[-1], Cin), np.nan)
        acc = prev[:own] @ Wk[0]
        dW[0::K] += prev[:own].T @ dy[vid[:own]]
        for k in range(1, K):
            Rk = R[K - 1 - k]
            new = (A[:Rk] @ prev) * (1.0 if k == 1 else 2.0) - (other[:Rk] if k > 1 else 0.0)
            other[:Rk] = new                                 # in place over T_{k-2}; rows beyond keep stale data
            prev, other = other, prev
            assert not np.isnan(prev[:Rk]).any()
            acc += prev[:own] @ Wk[k]
            dW[k::K] += prev[:own].T @ dy[vid[:own]]
        y[vid[:own]] = acc
        # adjoint by Clenshaw: b_k = G_k + 2 L b_{k+1} - b_{k+2} on the k-ring
        G = lambda k, Rk: dy[vid[:Rk]] @ Wk[k].T
        if K == 1:
            dx[vid[:own]] = G(0, own)
            continue
        bA, bB = np.full((R[-1], Cin), np.nan), np.full((R[-1], Cin), np.nan)
        bA[:R[K - 1]] = G(K - 1, R[K - 1])
        for k in range(K - 2, 0, -1):
            Rk = R[k]
            bB[:Rk] = G(k, Rk) - (bB[:Rk] if k < K - 2 else 0.0)
            bB[:Rk] += 2.0 * (A[:Rk] @ np.nan_to_num(bA))     # (columns outside the (k+1)-ring are never referenced)
            assert not np.isnan(A[:Rk] @ np.where(np.isnan(bA), np.inf, bA)).any()
            bA, bB = bB, bA
        t = G(0, own) - (bB[:own] if K > 2 else 0.0)
        dx[vid[:own]] = A[:own] @ np.nan_to_num(bA) + t
    return y, dx, dW


def test_patch_plan_and_patch_local_algorithm(mesh_ops):
    from cape_amd.graph import ChebPatchPlan, cheb_polys
    from cape_amd.mesh_sampling import rescale_L
    rng = np.random.default_rng(0)
    for level, K, Cin, Fout in ((6, 6, 16, 32), (4, 4, 8, 32), (6, 2, 16, 32), (6, 3, 24, 64)):
        L = mesh_ops["L"][level]
        M = L.shape[0]
        Lt = sp.csr_matrix(rescale_L(sp.csr_matrix(L), lmax=2), dtype=np.float64)
        plan = ChebPatchPlan(Lt, K, Cin, reserve_bytes=8 * 4 * Cin * Fout)
        own = np.concatenate([plan.vid[plan.pinfo[p, 0]:plan.pinfo[p, 0] + plan.pinfo[p, 3]] for p in range(plan.P)])
        assert np.array_equal(np.sort(own), np.arange(M)) and plan.own_max <= 256
        assert 2 * plan.rmax * (Cin + 4) * 4 + 8 * 4 * Cin * Fout <= 160 * 1024
        x = rng.standard_normal((M, Cin))
        W = rng.standard_normal((Cin * K, Fout))
        dy = rng.standard_normal((M, Fout))
        T = cheb_polys(L, K)
        ref_y = sum((T[k] @ x) @ W[k::K] for k in range(K))
        ref_dx = sum(T[k].T @ (dy @ W[k::K].T) for k in range(K))
        ref_dW = np.zeros_like(W)
        for k in range(K):
            ref_dW[k::K] = (T[k] @ x).T @ dy
        y, dx, dW = _emulate(plan, x, W, dy, K, Cin, Fout)
        for a, b in ((y, ref_y), (dx, ref_dx), (dW, ref_dW)):
            assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (level, K, np.abs(a - b).max())
    # the plan is a pure function of its inputs (the solver's start vector and the Fiedler sign are fixed)
    a, b = ChebPatchPlan(Lt, 3, 24), ChebPatchPlan(Lt, 3, 24)
    assert np.array_equal(a.vid, b.vid) and np.array_equal(a.pinfo, b.pinfo)
