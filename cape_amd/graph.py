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


# ------------------------------------------------------------------------------------------------------------------
# Vertex patches for the on-chip Chebyshev recurrence (csrc/cheb_fused.hip; polynomial orders above FUSE_MAX_K, e.g.
# BASELINE configs[1]: K = 6).  The recurrence T_k = 2 L~ T_{k-1} - T_{k-2} (reference lib/models.py:88-96) couples a
# vertex with its (K-1)-ring, so a workgroup that keeps T_{k-1} / T_{k-2} of a vertex patch in LDS needs the patch plus a
# (K-1)-ring halo and recomputes the halo redundantly; step k is valid on the (K-1-k)-ring, the trailing contraction only
# runs on the patch itself.  Compact, equal-sized patches keep that halo small: recursive spectral bisection (Fiedler
# vector of the induced sub-graph, split at the median) -- balanced to +-1 vertex and, on the SMPL mesh, a 5-ring halo of
# 1.5x the patch (BFS-level cuts: 3x).
# ------------------------------------------------------------------------------------------------------------------
def _adjacency(Lt):
    A = sp.csr_matrix(Lt).copy()
    A.setdiag(0)
    A.eliminate_zeros()
    A = (A + A.T).tocsr()
    A.data[:] = 1.0
    return A


def spectral_patches(A, nparts):
    """owner[v] in [0, nparts): recursive spectral bisection of the graph with (symmetric 0/1) adjacency ``A``."""
    from scipy.sparse.csgraph import breadth_first_order, connected_components
    from scipy.sparse.linalg import eigsh
    M = A.shape[0]
    owner = np.zeros(M, dtype=np.int32)

    def ordering(sub):
        n = sub.shape[0]
        nc, lab = connected_components(sub, directed=False)
        if nc > 1:                                       # components one after another
            return np.argsort(lab, kind="stable")
        try:
            deg = np.asarray(sub.sum(1)).ravel()
            vals, vecs = eigsh((sp.diags(deg) - sub).tocsc(), k=2, sigma=-1e-3, which="LM", tol=1e-6,
                               v0=np.linspace(-1.0, 1.0, n))
            f = vecs[:, np.argsort(vals)[1]]
            if f[np.argmax(np.abs(f))] < 0:              # fixed sign: the plan must not depend on the solver's mood
                f = -f
            return np.argsort(f, kind="stable")
        except Exception:                                # noqa: BLE001 -- any ordering is correct, only the halo grows
            o = breadth_first_order(sub, 0, directed=False, return_predecessors=False)
            return np.concatenate([o, np.setdiff1d(np.arange(n), o)])

    def rec(idx, k, base):
        if k == 1:
            owner[idx] = base
            return
        o = ordering(A[idx][:, idx].astype(np.float64))
        k1 = k // 2
        cut = int(round(len(idx) * k1 / k))
        rec(idx[o[:cut]], k1, base)
        rec(idx[o[cut:]], k - k1, base + k1)

    rec(np.arange(M), int(nparts), 0)
    return owner


class ChebPatchPlan(object):
    """Patches + halos + local CSR of L~ for one (Laplacian, K, Cin) layer, flattened for csrc/cheb_fused.hip.

    Per patch p (``pinfo[p]``, 16 int32): [0] offset into ``vid``; [1] first row of the patch in ``ell_col`` / ``ell_val``
    (the kernel's form; ``csr_rowptr_off[p]`` is the offset into ``rowptr`` of the CSR form kept for checks); [2] offset into
    ``lcol`` / ``val``; [3 + j] R_j = number of local vertices within ring <= j (j = 0 .. K-1; R_0 = the patch itself,
    local indices are sorted by ring); rows i < R_{K-2} have CSR rows (their neighbours all lie within ring K-1).
    """
    LDS_BUDGET = 158 * 1024            # of the CU's 160 KB
    MAX_OWN = 256                      # 8 waves x one 32-row MFMA tile
    MAX_ROWS = 1024                    # one thread per local row
    MAX_ROW_NNZ = 12                   # entries of a row the kernel keeps in registers (CF_W)

    def __init__(self, Lt, K, Cin, reserve_bytes=0):
        """``reserve_bytes``: LDS the kernel needs besides the two recurrence buffers (the backward kernel's cross-wave
        reduction area, 8 * Cin * Fout floats)."""
        Lt = as_csr64(Lt)
        M, K = Lt.shape[0], int(K)
        assert 2 <= K <= 8
        A = _adjacency(Lt)
        pitch = int(Cin) + 4
        rmax_allowed = min((self.LDS_BUDGET - int(reserve_bytes)) // (2 * pitch * 4), self.MAX_ROWS)
        if int(np.diff(Lt.indptr).max()) > self.MAX_ROW_NNZ:
            raise ValueError("rows of the operator are longer than the %d entries the kernel holds in registers" % self.MAX_ROW_NNZ)
        plan = None
        for nparts in (4, 6, 8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128):
            if -(-M // nparts) > self.MAX_OWN:
                continue
            cand = self._build(Lt, A, K, nparts)
            if cand["rmax"] <= rmax_allowed:
                plan = cand
                break
        if plan is None:
            raise ValueError("no patch plan fits the LDS for K = %d, Cin = %d on a %d-vertex graph" % (K, Cin, M))
        self.__dict__.update(plan)
        self.K, self.M, self.Cin, self.pitch = K, M, int(Cin), pitch

    @staticmethod
    def _build(Lt, A, K, nparts):
        M = Lt.shape[0]
        owner = spectral_patches(A, nparts)
        indptr, indices, data = Lt.indptr, Lt.indices, Lt.data
        pinfo = np.zeros((nparts, 16), dtype=np.int32)
        vids, rowptrs, lcols, vals = [], [], [], []
        voff = roff = eoff = 0
        rmax = 0
        for p in range(nparts):
            ring = np.full(M, -1, dtype=np.int32)
            cur = owner == p
            ring[cur] = 0
            order = [np.flatnonzero(cur)]
            R = [int(cur.sum())]
            for j in range(1, K):
                nxt = np.asarray((A @ cur.astype(np.float64)) > 0).ravel() & (ring < 0)
                ring[nxt] = j
                order.append(np.flatnonzero(nxt))
                R.append(R[-1] + int(nxt.sum()))
                cur = nxt
            loc = np.concatenate(order).astype(np.int32)          # local index -> global vertex, sorted by ring
            g2l = np.full(M, -1, dtype=np.int64)
            g2l[loc] = np.arange(len(loc))
            nrows = R[K - 2]
            rp = [0]
            for i in range(nrows):
                v = loc[i]
                cols = g2l[indices[indptr[v]:indptr[v + 1]]]
                assert (cols >= 0).all()
                o = np.argsort(cols, kind="stable")
                lcols.append(cols[o].astype(np.int32))
                vals.append(data[indptr[v]:indptr[v + 1]][o].astype(np.float32))
                rp.append(rp[-1] + len(cols))
            pinfo[p, 0], pinfo[p, 1], pinfo[p, 2] = voff, roff, eoff
            pinfo[p, 3:3 + K] = R
            vids.append(loc)
            rowptrs.append(np.asarray(rp, dtype=np.int32))
            voff += len(loc)
            roff += nrows + 1
            eoff += rp[-1]
            rmax = max(rmax, len(loc))
        # ELL form of the same rows (what the kernel reads): MAX_ROW_NNZ entries per row, real entries first, padding =
        # (own local index, 0); pinfo[p][1] is re-pointed at the patch's first ELL row
        W_ = ChebPatchPlan.MAX_ROW_NNZ
        nrow_total = sum(len(r) - 1 for r in rowptrs)
        ell_col = np.zeros((nrow_total, W_), dtype=np.int32)
        ell_val = np.zeros((nrow_total, W_), dtype=np.float32)
        csr_rowptr_off = pinfo[:, 1].copy()
        row0 = e = 0
        for p in range(nparts):
            rp = rowptrs[p]
            for i in range(len(rp) - 1):
                d = int(rp[i + 1] - rp[i])
                ell_col[row0 + i, :] = i
                ell_col[row0 + i, :d] = lcols[e]
                ell_val[row0 + i, :d] = vals[e]
                e += 1
            pinfo[p, 1] = row0
            row0 += len(rp) - 1
        return dict(P=nparts, pinfo=pinfo, vid=np.concatenate(vids), rowptr=np.concatenate(rowptrs), csr_rowptr_off=csr_rowptr_off,
                    ell_col=ell_col, ell_val=ell_val,
                    lcol=np.concatenate(lcols) if lcols else np.zeros(0, np.int32),
                    val=np.concatenate(vals) if vals else np.zeros(0, np.float32), rmax=int((rmax + 3) // 4 * 4),
                    own_max=int(pinfo[:, 3].max()), halo_factor=float(pinfo[:, 3 + K - 1].sum()) / M)
