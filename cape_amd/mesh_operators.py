"""Mesh hierarchy generation without psbody: the down-/up-sampling operators D, U and the adjacencies A that the
Chebyshev path consumes (host side, numpy/scipy; offline -- runs once per mesh topology).

Mirrors the operator precompute of the reference (lib/mesh_sampling.py:40-263, called from main.py:39):
``generate_transform_matrices(mesh, factors) -> M, A, D, U, E`` with
  * QSlim-style decimation restricted to subset placement (``qslim_decimator_transformer`` :111-225): vertex
    quadrics from the face planes (:40-65), a priority queue of edges keyed by the cheaper of the two
    "move one endpoint onto the other" costs, stale keys re-queued, the surviving vertices keep their positions,
    so D is a row selection (``_get_sparse_transform`` :228-242);
  * up-sampling by projecting every fine vertex onto the coarse surface and expressing the foot point in the
    vertices of the closest triangle (``setup_deformation_transfer`` :67-108);
  * vertex adjacency with one count per incident face (psbody's ``get_vert_connectivity``: interior edges 2.0,
    SURVEY appendix C11) and the unique edge list (``get_vertices_per_edge``).
The reference needs psbody.mesh (Mesh container, AABB tree, connectivity helpers); here a mesh is any object with
``.v`` [V,3] and ``.f`` [F,3] (``Mesh`` below), the closest-point search is exact over KD-tree candidates, and the queue
keeps per-vertex entry lists instead of scanning the whole queue at every collapse (same queue contents and
therefore the same collapse sequence, O(degree) instead of O(edges) per collapse).

Checked against the operators the reference ships for the SMPL template (tests/test_mesh_operators.py): the eight
``for_demo`` levels (factors 1,2,1,2,1,2,1,1: 6890 -> 3445 -> 1723 -> 862) are reproduced exactly -- identical
selections D, identical adjacencies A, up-sampling weights U to the float32 precision they are stored in.  (The
shipped ``ds2`` set was generated from another mesh: its U does not reconstruct the template.)
"""
import collections
import heapq
import math

import numpy as np
import scipy.sparse as sp

Mesh = collections.namedtuple("Mesh", ["v", "f"])


# ---------------------------------------------------------------------------------------------
# connectivity
# ---------------------------------------------------------------------------------------------
def get_vert_connectivity(mesh_v, mesh_f=None):
    """Sparse V x V matrix counting, for every vertex pair, the faces in which the two are joined by an edge
    (2.0 on interior edges of a closed manifold).  Accepts ``(mesh)`` or ``(verts, faces)`` like psbody."""
    if mesh_f is None:
        mesh_v, mesh_f = mesh_v.v, mesh_v.f
    n = len(mesh_v)
    f = np.asarray(mesh_f, dtype=np.int64)
    vc = sp.csc_matrix((n, n))
    for i in range(3):
        a, b = f[:, i], f[:, (i + 1) % 3]
        m = sp.csc_matrix((np.ones(len(a)), (a, b)), shape=(n, n))
        vc = vc + m + m.T
    return vc


def get_vertices_per_edge(mesh_v, mesh_f=None):
    """E x 2 array of the unique edges, smaller index first."""
    vc = sp.coo_matrix(get_vert_connectivity(mesh_v, mesh_f))
    keep = vc.row < vc.col
    e = np.column_stack((vc.row[keep], vc.col[keep]))
    return e[np.lexsort((e[:, 1], e[:, 0]))]


# ---------------------------------------------------------------------------------------------
# decimation
# ---------------------------------------------------------------------------------------------
def vertex_quadrics(mesh):
    """[V, 4, 4] sum over the incident faces of the outer product of the face's unit plane equation
    (reference :40-65; the plane is the null vector of [verts | 1], as there)."""
    v, f = np.asarray(mesh.v, dtype=np.float64), np.asarray(mesh.f, dtype=np.int64)
    Q = np.zeros((len(v), 4, 4))
    tri = np.concatenate((v[f], np.ones((len(f), 3, 1))), axis=2)          # [F, 3, 4]
    eq = np.linalg.svd(tri)[2][:, -1, :]                                  # last right-singular vector per face
    eq = eq / np.linalg.norm(eq[:, :3], axis=1, keepdims=True)
    outer = eq[:, :, None] * eq[:, None, :]
    for k in range(3):
        np.add.at(Q, f[:, k], outer)                                       # face order, like the reference's loop
    return Q


def _collapse_cost(Qv, r, c, v):
    Qsum = Qv[r] + Qv[c]
    p1 = np.append(v[r], 1.0).reshape(-1, 1)
    p2 = np.append(v[c], 1.0).reshape(-1, 1)
    destroy_c = float(p1.T.dot(Qsum).dot(p1)[0, 0])
    destroy_r = float(p2.T.dot(Qsum).dot(p2)[0, 0])
    return destroy_c, destroy_r, Qsum


def _get_sparse_transform(faces, num_original_verts):
    """Compact the surviving vertices of a decimated face list: returns the re-indexed faces and the selection matrix
    D [kept, original] with one unit entry per row at the kept vertex (reference :228-241).  ``np.unique`` delivers both the
    sorted survivors and, via ``return_inverse``, every face corner's rank among them -- the new index."""
    kept, corner_rank = np.unique(np.asarray(faces).reshape(-1), return_inverse=True)
    rows = np.arange(kept.size)
    select = sp.csc_matrix((np.ones(kept.size), (rows, kept)), shape=(kept.size, int(num_original_verts)))
    return corner_rank.reshape(-1, 3), select


def qslim_decimator_transformer(mesh, factor=None, n_verts_desired=None):
    """Simplify ``mesh`` to ``ceil(V * factor)`` (or ``n_verts_desired``) vertices.
    Returns ``(new_faces, D)`` with D the [V_kept, V] selection matrix (reference :111-225)."""
    if factor is None and n_verts_desired is None:
        raise Exception('Need either factor or n_verts_desired.')
    v = np.asarray(mesh.v, dtype=np.float64)
    if n_verts_desired is None:
        n_verts_desired = math.ceil(len(v) * factor)
    Qv = vertex_quadrics(mesh)

    adj = get_vertices_per_edge(mesh)
    adj = sp.csc_matrix((np.ones(len(adj)), (adj[:, 0], adj[:, 1])), shape=(len(v), len(v)))
    adj = (adj + adj.T).tocoo()

    # queue entries are mutable [cost, (r, c)] records (they order like the reference's tuples); first[x] / second[x]
    # hold the live entries whose first / second endpoint is x, so renaming a destroyed vertex touches only those
    queue = []
    first = collections.defaultdict(dict)
    second = collections.defaultdict(dict)

    def push(cost, edge):
        e = [cost, edge]
        heapq.heappush(queue, e)
        first[edge[0]][id(e)] = e
        second[edge[1]][id(e)] = e

    def forget(e):
        first[e[1][0]].pop(id(e), None)
        second[e[1][1]].pop(id(e), None)

    for r, c in zip(adj.row.tolist(), adj.col.tolist()):
        if r > c:
            continue
        dc, dr, _ = _collapse_cost(Qv, r, c, v)
        push(min(dc, dr), (r, c))

    faces = np.asarray(mesh.f).copy()
    nverts_total = len(v)
    while nverts_total > n_verts_desired:
        e = heapq.heappop(queue)
        forget(e)
        r, c = e[1]
        if r == c:
            continue
        dc, dr, Qsum = _collapse_cost(Qv, r, c, v)
        cost = min(dc, dr)
        if cost > e[0]:
            push(cost, e[1])                              # stale key: re-queue with the current cost
            continue
        to_keep, to_destroy = (r, c) if dc < dr else (c, r)
        faces[faces == to_destroy] = to_keep
        renamed_first = list(first[to_destroy].values())
        renamed_second = list(second[to_destroy].values())
        for q in renamed_first:
            first[to_destroy].pop(id(q), None)
            q[1] = (to_keep, q[1][1])
            first[to_keep][id(q)] = q
        for q in renamed_second:
            second[to_destroy].pop(id(q), None)
            q[1] = (q[1][0], to_keep)
            second[to_keep][id(q)] = q
        Qv[r] = Qsum
        Qv[c] = Qsum
        degenerate = (faces[:, 0] == faces[:, 1]) | (faces[:, 1] == faces[:, 2]) | (faces[:, 2] == faces[:, 0])
        faces = faces[~degenerate].copy()
        nverts_total = len(np.unique(faces))
    return _get_sparse_transform(faces, len(v))


# ---------------------------------------------------------------------------------------------
# up-sampling
# ---------------------------------------------------------------------------------------------
def _closest_on_triangles(a, b, c, p):
    """Closest point of triangle (a, b, c) to p, row by row ([n, 3] each): (part, foot point, squared distance).
    part: 0 interior, 1-3 on edge (ab, bc, ca), 4-6 at vertex a, b, c."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    va, vb, vc = d3 * d6 - d5 * d4, d5 * d2 - d1 * d6, d1 * d4 - d3 * d2
    with np.errstate(divide='ignore', invalid='ignore'):
        denom = va + vb + vc
        v_in, w_in = vb / denom, vc / denom
        t_ab = d1 / (d1 - d3)
        t_ac = d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    zeros, ones = np.zeros_like(d1), np.ones_like(d1)
    # region tests in the order of the classic closest-point-on-triangle routine; the first that holds wins
    regions = [((d1 <= 0) & (d2 <= 0), 4, (ones, zeros, zeros)),                        # vertex a
               ((d3 >= 0) & (d4 <= d3), 5, (zeros, ones, zeros)),                       # vertex b
               ((vc <= 0) & (d1 >= 0) & (d3 <= 0), 1, (1 - t_ab, t_ab, zeros)),         # edge ab
               ((d6 >= 0) & (d5 <= d6), 6, (zeros, zeros, ones)),                       # vertex c
               ((vb <= 0) & (d2 >= 0) & (d6 <= 0), 3, (1 - t_ac, zeros, t_ac)),         # edge ca
               ((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), 2, (zeros, 1 - t_bc, t_bc))]   # edge bc
    part = np.zeros(d1.shape, dtype=np.int64)
    bw = np.stack((1 - v_in - w_in, v_in, w_in), -1)                                    # interior by default
    done = np.zeros(d1.shape, dtype=bool)
    for cond, code, bary in regions:
        take = cond & ~done
        part[take] = code
        bw[take] = np.stack(bary, -1)[take]
        done |= take
    foot = bw[:, 0:1] * a + bw[:, 1:2] * b + bw[:, 2:3] * c
    dist = ((foot - p) ** 2).sum(-1)
    dist[~np.isfinite(dist)] = np.inf                                                   # degenerate faces never win
    return part, foot, dist


def closest_points_on_mesh(verts, faces, points):
    """For every point: (face index, part, foot point) of the closest point of the triangle mesh -- what the
    reference reads from psbody's ``AabbTree.nearest(..., nearest_part=True)`` (:73-75, :93-105).  Exact: the
    candidate faces of a point are ALL faces whose bounding sphere reaches into the ball around the point that
    touches the nearest mesh vertex; ties go to the lowest face index."""
    from scipy.spatial import cKDTree
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    points = np.asarray(points, dtype=np.float64)
    tri = verts[faces]                                                                  # [F, 3, 3]
    centre = tri.mean(1)
    radius = np.sqrt(((tri - centre[:, None, :]) ** 2).sum(-1)).max()                   # largest centroid-to-corner distance
    upper = cKDTree(verts).query(points)[0]                                             # the closest point is no farther
    cand = cKDTree(centre).query_ball_point(points, upper * (1 + 1e-9) + radius * (1 + 1e-9) + 1e-12)
    counts = np.array([len(x) for x in cand])
    pi = np.repeat(np.arange(len(points)), counts)
    fi = np.concatenate([np.sort(np.asarray(x, dtype=np.int64)) for x in cand])
    part, foot, dist = _closest_on_triangles(tri[fi, 0], tri[fi, 1], tri[fi, 2], points[pi])
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    out_f = np.zeros(len(points), dtype=np.int64)
    out_part = np.zeros(len(points), dtype=np.int64)
    out_pt = np.zeros((len(points), 3))
    for i, (s, n) in enumerate(zip(starts, counts)):
        k = s + int(np.argmin(dist[s:s + n]))
        out_f[i], out_part[i], out_pt[i] = fi[k], part[k], foot[k]
    return out_f, out_part, out_pt


def setup_deformation_transfer(source, target, use_normals=False):
    """[V_target, V_source] sparse matrix (3 stored entries per row) expressing every target vertex in the vertices
    of the closest source triangle (reference :67-108)."""
    sv, sf = np.asarray(source.v, dtype=np.float64), np.asarray(source.f, dtype=np.int64)
    tv = np.asarray(target.v, dtype=np.float64)
    n = tv.shape[0]
    rows = np.repeat(np.arange(n), 3)
    cols = np.zeros(3 * n, dtype=np.int64)
    coeffs = np.zeros(3 * n)
    nearest_faces, nearest_parts, nearest_pts = closest_points_on_mesh(sv, sf, tv)
    for i in range(n):
        nearest_f = sf[nearest_faces[i]]
        cols[3 * i:3 * i + 3] = nearest_f
        n_id = int(nearest_parts[i])
        if n_id == 0:                                                       # foot point inside the triangle
            A = sv[nearest_f].T
            coeffs[3 * i:3 * i + 3] = np.linalg.lstsq(A, nearest_pts[i], rcond=None)[0]
        elif n_id <= 3:                                                     # on an edge: fit the target vertex itself
            A = np.vstack((sv[nearest_f[n_id - 1]], sv[nearest_f[n_id % 3]])).T
            t = np.linalg.lstsq(A, tv[i], rcond=None)[0]
            coeffs[3 * i + n_id - 1] = t[0]
            coeffs[3 * i + n_id % 3] = t[1]
        else:                                                               # at a vertex
            coeffs[3 * i + n_id - 4] = 1.0
    return sp.csc_matrix((coeffs, (rows, cols)), shape=(n, sv.shape[0]))


# ---------------------------------------------------------------------------------------------
# the hierarchy
# ---------------------------------------------------------------------------------------------
def generate_transform_matrices(mesh, factors):
    """``M, A, D, U, E`` for a list of down-sampling factors (reference :244-263, main.py:31-39): meshes, adjacency
    matrices (len+1), down-sampling and up-sampling matrices (len) and edge lists (len+1)."""
    level = Mesh(np.asarray(mesh.v, dtype=np.float64), np.asarray(mesh.f, dtype=np.int64))
    levels, downs, ups = [level], [], []
    for f in factors:
        faces_next, select = qslim_decimator_transformer(level, factor=1.0 / f)
        coarse = Mesh(select.dot(level.v), faces_next)               # kept vertices keep their positions
        downs.append(select)
        ups.append(setup_deformation_transfer(coarse, level))        # fine vertices from the coarse surface
        levels.append(coarse)
        level = coarse
    adjacency = [get_vert_connectivity(m) for m in levels]
    edges = [get_vertices_per_edge(m) for m in levels]
    return levels, adjacency, downs, ups, edges


def load_obj(path):
    """Vertices and triangle faces of a Wavefront .obj (``v`` / ``f`` lines only; 1-based ``f a/b/c`` accepted)."""
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith('v '):
                v.append([float(t) for t in line.split()[1:4]])
            elif line.startswith('f '):
                f.append([int(t.split('/')[0]) - 1 for t in line.split()[1:4]])
    return Mesh(np.asarray(v, dtype=np.float64), np.asarray(f, dtype=np.int64))


def save_transform_matrices(out_dir, A, D, U):
    """Write ``A.npy``, ``D.npy``, ``U.npy`` the way the reference ships them (``data/transform_matrices/<set>/``:
    pickled object arrays of scipy matrices, read back by ``lib/load_data.py:9-11,22-24`` with ``np.load``)."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    for name, mats in (("A", A), ("D", D), ("U", U)):
        arr = np.empty(len(mats), dtype=object)
        for i, m in enumerate(mats):
            arr[i] = sp.csc_matrix(m)
        np.save(os.path.join(out_dir, name + ".npy"), arr, allow_pickle=True)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="generate the mesh down-/up-sampling operators of a template mesh "
                                             "(the precompute of the reference's main.py:31-44, without psbody)")
    ap.add_argument("obj", help="template mesh (.obj, triangles)")
    ap.add_argument("--factors", type=int, nargs="+", default=[1, 2, 1, 2, 1, 2, 1, 1],
                    help="down-sampling factor per conv layer (main.py:31-36)")
    ap.add_argument("--out", required=True, help="directory for A.npy / D.npy / U.npy")
    a = ap.parse_args(argv)
    M, A, D, U, E = generate_transform_matrices(load_obj(a.obj), a.factors)
    save_transform_matrices(a.out, A, D, U)
    print("levels:", " -> ".join(str(m.v.shape[0]) for m in M), " written to", a.out)


if __name__ == "__main__":
    main()
