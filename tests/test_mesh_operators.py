"""Operator generation without psbody (cape_amd/mesh_operators.py, reference lib/mesh_sampling.py:40-263): the
hierarchy regenerated from the SMPL template must reproduce the operators the reference SHIPS for it
(data/transform_matrices/for_demo/{A,D,U}.npy, re-encoded in cape_amd/data/smpl_mesh_pack.npz) -- identical
down-sampling selections, identical adjacencies, up-sampling weights to the float32 precision they are stored in.
CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from cape_amd import mesh_operators as mo
from cape_amd import mesh_sampling
from cape_amd.load_data import load_pack

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pack():
    return load_pack()


@pytest.fixture(scope="module")
def template(pack):
    faces = np.load(os.path.join(GOLDEN, "template_faces.npy"))
    return mo.Mesh(pack["template_verts"], faces.astype(np.int64))


def shipped(pack, hier, name, i):
    k = "%s_%s_%d" % (hier, name, i)
    return sp.csc_matrix((pack[k + "_data"], pack[k + "_indices"], pack[k + "_indptr"]), shape=tuple(pack[k + "_shape"]))


@pytest.fixture(scope="module")
def hierarchy(template):
    # main.py:31-39 for num_conv_layers = 8, ds_factor = 2
    return mesh_sampling.generate_transform_matrices(template, [1, 2, 1, 2, 1, 2, 1, 1])


def test_connectivity_of_the_template(template, pack):
    A = mo.get_vert_connectivity(template)
    assert abs(A - shipped(pack, "for_demo", "A", 0)).max() == 0 and set(np.unique(A.data)) == {2.0}
    assert abs(A - shipped(pack, "ds2", "A", 0)).max() == 0
    E = mo.get_vertices_per_edge(template)
    ref = np.sort(pack["edges_smpl"].astype(np.int64), axis=1)              # data/edges_smpl.npy (lib/models.py:45)
    assert E.shape == ref.shape == (20664, 2)
    assert np.array_equal(E, ref[np.lexsort((ref[:, 1], ref[:, 0]))])


def test_regenerated_hierarchy_equals_the_shipped_one(hierarchy, pack):
    M, A, D, U, E = hierarchy
    assert [m.v.shape[0] for m in M] == [6890, 6890, 3445, 3445, 1723, 1723, 862, 862, 862]
    assert len(A) == 9 and len(D) == len(U) == 8 and len(E) == 9
    for i in range(8):
        Dr, Ur = shipped(pack, "for_demo", "D", i), shipped(pack, "for_demo", "U", i)
        assert D[i].shape == Dr.shape and U[i].shape == Ur.shape
        assert np.array_equal(D[i].tocsr().indices, Dr.tocsr().indices), "level %d keeps other vertices" % i
        assert np.all(D[i].data == 1.0) and D[i].nnz == D[i].shape[0]
        assert abs(U[i] - Ur).max() < 1e-7                                  # shipped as float32
        assert U[i].nnz == Ur.nnz == 3 * U[i].shape[0]                      # three stored weights per fine vertex
        assert abs(A[i + 1] - shipped(pack, "for_demo", "A", i + 1)).max() == 0
    # coarse meshes stay closed 2-manifolds of the template's genus: V - E + F = 2
    for m, e in zip(M, E):
        assert m.v.shape[0] - len(e) + len(m.f) == 2


def test_operators_drive_the_laplacian_path(hierarchy):
    """What main.py:40-44 does with the result: float32 casts and normalised Laplacians, one per level."""
    M, A, D, U, E = hierarchy
    p = [a.shape[0] for a in A]
    L = [mesh_sampling.laplacian(a.astype('float32'), normalized=True) for a in A]
    assert p == [6890, 6890, 3445, 3445, 1723, 1723, 862, 862, 862]
    for l in L:
        assert l.dtype == np.float32 and abs(l - l.T).max() < 1e-6 and abs(l.diagonal() - 1).max() < 1e-6
    # up-sampling after down-sampling returns the kept vertices to where they were
    x = M[1].v
    back = U[1].dot(D[1].dot(x))
    kept = D[1].tocsr().indices
    assert np.abs(back[kept] - x[kept]).max() < 1e-9
    assert np.abs(back - x).max() < 0.01                                    # metres, on the T-pose template


def test_closest_point_parts_and_ties():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5]], dtype=np.float64)
    f = np.array([[0, 1, 2], [1, 3, 2]])
    pts = np.array([[0.25, 0.25, 1.0],      # above the interior of face 0
                    [0.5, -1.0, 0.0],       # beyond edge v0v1
                    [-1.0, -1.0, 0.0],      # beyond vertex 0
                    [0.5, 0.5, 0.0],        # on the shared edge v1v2: tie, lowest face index wins
                    [2.0, 2.0, 0.5]])       # beyond vertex 3 (local vertex 1 of face 1)
    face, part, foot = mo.closest_points_on_mesh(v, f, pts)
    assert face.tolist() == [0, 0, 0, 0, 1]
    assert part.tolist() == [0, 1, 4, 2, 5]
    assert np.allclose(foot, [[0.25, 0.25, 0], [0.5, 0, 0], [0, 0, 0], [0.5, 0.5, 0], [1, 1, 0.5]])
    # the weights reproduce the foot point (interior) or the fitted point (edge / vertex)
    U = mo.setup_deformation_transfer(mo.Mesh(v + [0.3, 0.2, 0.1], f), mo.Mesh(pts + [0.3, 0.2, 0.1], None))
    assert U.shape == (5, 4) and U.nnz == 15


def test_decimation_argument_errors(template):
    with pytest.raises(Exception):
        mo.qslim_decimator_transformer(template)
    tet = mo.Mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]), np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]]))
    f, D = mo.qslim_decimator_transformer(tet, n_verts_desired=4)           # nothing to do
    assert D.shape == (4, 4) and len(f) == 4


def test_saved_operator_files_load_like_the_shipped_ones(tmp_path):
    """A.npy / D.npy / U.npy written for a small mesh are read back the way lib/load_data.py:9-15 reads the shipped
    files (np.load of a pickled list, astype float32)."""
    # an octahedron subdivided once: 18 vertices, 32 faces
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    mid, verts, faces = {}, [list(p) for p in v], []

    def m(a, b):
        k = (min(a, b), max(a, b))
        if k not in mid:
            p = (v[a] + v[b]) / 2
            verts.append(list(p / np.linalg.norm(p)))
            mid[k] = len(verts) - 1
        return mid[k]

    for a, b, c in f:
        ab, bc, ca = m(a, b), m(b, c), m(c, a)
        faces += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
    mesh = mo.Mesh(np.array(verts), np.array(faces))
    M, A, D, U, E = mo.generate_transform_matrices(mesh, [1, 2])
    assert [x.v.shape[0] for x in M] == [18, 18, 9]
    mo.save_transform_matrices(str(tmp_path / "ops"), A, D, U)
    for name, want in (("A", A), ("D", D), ("U", U)):
        got = list(np.load(str(tmp_path / "ops" / (name + ".npy")), encoding='latin1', allow_pickle=True))
        got = [g.astype('float32') for g in got]
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert sp.issparse(g) and g.shape == w.shape and abs(g - w.astype('float32')).max() < 1e-6
    # the coarse level is a closed manifold again and up-sampling reproduces the kept vertices
    assert M[2].v.shape[0] - len(E[2]) + len(M[2].f) == 2
    kept = D[1].tocsr().indices
    assert np.abs(U[1].dot(M[2].v)[kept] - M[1].v[kept]).max() < 1e-9
