"""Host-side operator algebra for the fused gather-GEMM kernels.

The reference applies, per layer, a chain of fixed sparse matrices around a dense weight
contraction: unpool ``U`` (lib/models.py:782), the Chebyshev recurrence in the rescaled
Laplacian ``L~`` (:74-96) and pool ``D`` (:168).  Because those matrices are constants of the
mesh hierarchy, this module precomposes them ONCE on the host (float64, rounded once to
fp32) into per-order operators

    S_k = Dsel . T_k(L~) . U          k = 0..K-1

stored as column-sorted int32/fp32 CSR -- the kernel then needs no halo exchange between
SpMV and GEMM stages -- together with their transposes ``S_k^T`` for the data gradient.
``D`` is folded in only when it is a 0/1 row selection (what QSlim decimation produces and
what the reference ships: SURVEY appendix D); anything else is applied by the standalone
``cape_spmm`` operator.  Orders above ``FUSE_MAX_K`` use the explicit recurrence instead
(nnz of T_k grows with the k-ring).
"""
import numpy as np
import scipy.sparse as sp

from .mesh_sampling import rescale_L

FUSE_MAX_K = 3
# Entries below PRUNE_REL * max|S| are dropped after composition.  The shipped identity-level
# up-sampling matrices carry two ~1e-11 barycentric weights per row next to a 1.0 (SURVEY
# appendix D); their contribution (<= 2e-9 * max|x| per output) is far below fp32 resolution
# of the result, and pruning them turns those levels into exact identities (no gather).
PRUNE_REL = 1e-9


def as_csr64(mat):
    m = sp.csr_matrix(mat, dtype=np.float64)
    m.sum_duplicates()
    m.sort_indices()
    return m


def prune(mat, rel=PRUNE_REL):
    m = as_csr64(mat)
    if m.nnz == 0 or rel <= 0:
        return m
    thr = rel * np.abs(m.data).max()
    m.data[np.abs(m.data) < thr] = 0.0
    m.eliminate_zeros()
    return m


def is_identity(mat):
    m = as_csr64(mat)
    if m.shape[0] != m.shape[1] or m.nnz != m.shape[0]:
        return False
    return bool(np.array_equal(m.indices, np.arange(m.shape[0])) and np.all(m.data == 1.0)
                and np.array_equal(m.indptr, np.arange(m.shape[0] + 1)))


def is_row_selection(mat):
    """True if every row has exactly one entry equal to 1.0 (a 0/1 row-selection matrix)."""
    m = as_csr64(mat)
    return bool(m.nnz == m.shape[0] and np.all(np.diff(m.indptr) == 1) and np.all(m.data == 1.0))


def cheb_polys(L, K):
    """[T_0(L~), ..., T_{K-1}(L~)] as float64 CSR, with L~ = rescale_L(L, 2)
    (reference lib/models.py:74-75, 90-96)."""
    Lt = as_csr64(rescale_L(sp.csr_matrix(L), lmax=2))
    n = Lt.shape[0]
    terms = [sp.identity(n, dtype=np.float64, format="csr")]
    if K > 1:
        terms.append(Lt)
    for _ in range(2, K):
        terms.append(as_csr64(2.0 * (Lt @ terms[-1]) - terms[-2]))
    return terms[:K]


class HostCSR(object):
    """Column-sorted int32/fp32 CSR on the host (None arrays = identity)."""

    def __init__(self, mat):
        m = as_csr64(mat)
        self.shape = m.shape
        self.identity = is_identity(m)
        self.rowptr = m.indptr.astype(np.int32)
        self.colidx = m.indices.astype(np.int32)
        self.vals = m.data.astype(np.float32)
        self.nnz = int(m.nnz)
        self.max_row = int(np.diff(m.indptr).max()) if m.shape[0] else 0
        self.min_row = int(np.diff(m.indptr).min()) if m.shape[0] else 0

    def to_scipy(self):
        return sp.csr_matrix((self.vals, self.colidx, self.rowptr), shape=self.shape)


class ConvOperators(object):
    """Precomposed operators of one graph-conv layer: ``fwd[k] = S_k`` (Mo x Mi) and
    ``bwd[k] = S_k^T`` (Mi x Mo), or ``recurrence`` data for K > FUSE_MAX_K."""

    def __init__(self, L, K, unpool=None, pool=None, prune_rel=PRUNE_REL):
        self.K = int(K)
        n = L.shape[0] if L is not None else None
        self.pool_fused = pool is None or is_row_selection(pool) or is_identity(pool)
        self.fused = self.K <= FUSE_MAX_K
        U = None if unpool is None else prune(unpool, prune_rel)
        Dm = None if (pool is None or not self.pool_fused) else as_csr64(pool)
        self.unfused_pool = None if (pool is None or self.pool_fused) else as_csr64(pool)
        if self.K == 1:
            terms = [sp.identity(n if n is not None else U.shape[0], dtype=np.float64, format="csr")]
        elif self.fused:
            terms = cheb_polys(L, self.K)
        else:
            terms = None
        self.Mi = (U.shape[1] if U is not None else n)
        if self.fused:
            self.fwd, self.bwd = [], []
            for T in terms:
                S = T
                if U is not None:
                    S = S @ U
                if Dm is not None:
                    S = Dm @ S
                S = prune(S, prune_rel)
                self.fwd.append(HostCSR(S))
                self.bwd.append(HostCSR(S.T))
            self.Mo = self.fwd[0].shape[0]
        else:
            # explicit recurrence on L~ (and its transpose for the adjoint); unpool / pool are
            # applied as separate sparse operators by the caller.
            Lt = as_csr64(rescale_L(sp.csr_matrix(L), lmax=2))
            self.Lt = HostCSR(Lt)
            self.LtT = HostCSR(Lt.T)
            self.unfused_unpool = U
            if Dm is not None:
                self.unfused_pool = Dm
            self.Mo = (Dm.shape[0] if Dm is not None else n)
        self.row_scale_terms = None

    def cond_row_terms(self):
        """For vertex-constant input channels (the tiled condition vector, lib/models.py:813-832):
        S_k (1 y^T) = (S_k 1) y^T, so those channels contribute  s_k[r] * (y W_k)  -- returns the
        K row-sum vectors s_k = S_k 1 (fp32)."""
        assert self.fused
        return [np.asarray(h.to_scipy().astype(np.float64).sum(axis=1)).ravel().astype(np.float32)
                for h in self.fwd]


def vertex_edge_table(edges, num_verts):
    """CSR over vertices of incident edges for the edge-loss gradient: entry code = 2*e + s,
    s = 0 if the vertex is the first endpoint of edge e (sign +), 1 if the second (sign -)."""
    edges = np.asarray(edges, dtype=np.int64)
    E = edges.shape[0]
    verts = np.concatenate([edges[:, 0], edges[:, 1]])
    codes = np.concatenate([2 * np.arange(E), 2 * np.arange(E) + 1])
    order = np.argsort(verts, kind="stable")
    verts, codes = verts[order], codes[order]
    ptr = np.zeros(num_verts + 1, dtype=np.int64)
    np.add.at(ptr, verts + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr.astype(np.int32), codes.astype(np.int32)
