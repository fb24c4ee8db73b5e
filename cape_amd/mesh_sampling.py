"""Operator precompute for the Chebyshev hot path (host side, numpy/scipy): same names, argument meaning and
return values as the reference's lib/mesh_sampling.py, without its psbody dependency, so ``main.py:39-44`` /
``lib/load_data.py:17,31`` style callers keep working:

  * ``laplacian`` (reference :10-29) and ``rescale_L`` (:31-38) -- defined here;
  * ``generate_transform_matrices``, ``qslim_decimator_transformer``, ``setup_deformation_transfer``,
    ``vertex_quadrics``, ``_get_sparse_transform`` (:40-263) -- implemented in ``cape_amd.mesh_operators`` and
    re-exported; they regenerate the operators the reference ships for the SMPL template exactly
    (tests/test_mesh_operators.py).
"""
import numpy as np
import scipy.sparse as sp

from .mesh_operators import (Mesh, _get_sparse_transform, generate_transform_matrices, get_vert_connectivity,  # noqa: F401
                             get_vertices_per_edge, qslim_decimator_transformer, setup_deformation_transfer,
                             vertex_quadrics)


def laplacian(W, normalized=True):
    """Graph Laplacian of adjacency ``W`` (any scipy sparse format, square).

    normalized: ``I - D^-1/2 W D^-1/2`` with ``d = colsum(W) + spacing(0)``
    (reference lib/mesh_sampling.py:21-25); otherwise ``D - W`` (:17-19).
    Returns CSR in ``W.dtype``.
    """
    n = W.shape[0]
    deg = W.sum(axis=0)                      # 1 x n numpy matrix, W.dtype
    if not normalized:
        return (sp.diags(np.asarray(deg).ravel(), 0) - W).tocsr()
    deg = deg + np.spacing(np.array(0, W.dtype))
    inv_sqrt = np.asarray(1 / np.sqrt(deg)).ravel()
    Dm = sp.diags(inv_sqrt, 0)
    eye = sp.identity(n, dtype=W.dtype)
    L = eye - Dm * W * Dm
    L = L.tocsr()
    assert sp.isspmatrix_csr(L)
    return L


def rescale_L(L, lmax=2):
    """Map the spectrum of ``L`` from [0, lmax] to [-1, 1]: ``L / (lmax/2) - I``
    (reference lib/mesh_sampling.py:31-38).  Never mutates the caller's matrix
    (the reference only rebinds for lmax=2)."""
    n = L.shape[0]
    eye = sp.identity(n, format="csr", dtype=L.dtype)
    scaled = L / (lmax / 2)
    return (scaled - eye).tocsr()
