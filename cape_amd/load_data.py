"""Loader for the precomputed mesh operators (adjacency A, down-sampling D, up-sampling U).

Host-side mirror of reference lib/load_data.py:7-32 ``load_graph_mtx`` (same name,
same return tuples).  Two sources:

* a CAPE checkout (``project_dir/data/transform_matrices/{ds2,for_demo}/*.npy``), read
  exactly the way the reference reads them (pickled scipy ``csc_matrix`` lists), or
* the package's plain-array pack ``cape_amd/data/smpl_mesh_pack.npz`` (made by
  tools/make_operator_pack.py from those same files) -- what the GPU box uses, since
  /root/reference does not exist there.

``BodyData`` (reference lib/load_data.py:35-150) is out of scope: licensed dataset.
"""
import os
import numpy as np
import scipy.sparse as sp

from .mesh_sampling import laplacian

_PACK_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                             "smpl_mesh_pack.npz")


def _from_pack(pack, hier, name):
    count = int(pack["%s_%s_count" % (hier, name)])
    out = []
    for i in range(count):
        key = "%s_%s_%d" % (hier, name, i)
        shape = tuple(int(v) for v in pack[key + "_shape"])
        out.append(sp.csc_matrix((pack[key + "_data"], pack[key + "_indices"], pack[key + "_indptr"]),
                                 shape=shape))
    return out


def _from_checkout(project_dir, hier, name):
    path = os.path.join(project_dir, "data", "transform_matrices", hier, name + ".npy")
    return list(np.load(path, encoding="latin1", allow_pickle=True))


def load_pack(pack_path=None):
    """Open the plain-array operator/fixture pack."""
    return np.load(pack_path or _PACK_DEFAULT)


def load_graph_mtx(project_dir=None, load_for_demo=False, pack_path=None):
    """Return ``(L_ds2, D_ds2, U_ds2)`` or, with ``load_for_demo``,
    ``(L, D, U, p, L_ds2, D_ds2, U_ds2)`` -- lists of fp32 scipy matrices
    (reference lib/load_data.py:7-32).  ``project_dir=None`` (or a directory without the
    shipped .npy files) falls back to the pack."""
    use_checkout = project_dir is not None and os.path.exists(
        os.path.join(project_dir, "data", "transform_matrices", "ds2", "A.npy"))
    if use_checkout:
        get = lambda hier, name: _from_checkout(project_dir, hier, name)
    else:
        pack = load_pack(pack_path)
        get = lambda hier, name: _from_pack(pack, hier, name)

    f32 = lambda mats: [m.astype("float32") for m in mats]
    A_ds2, D_ds2, U_ds2 = f32(get("ds2", "A")), f32(get("ds2", "D")), f32(get("ds2", "U"))
    L_ds2 = [laplacian(a, normalized=True) for a in A_ds2]
    if not load_for_demo:
        return L_ds2, D_ds2, U_ds2
    A = get("for_demo", "A")
    p = [a.shape[0] for a in A]
    A, D, U = f32(A), f32(get("for_demo", "D")), f32(get("for_demo", "U"))
    L = [laplacian(a, normalized=True) for a in A]
    return L, D, U, p, L_ds2, D_ds2, U_ds2


# SMPL joints that influence clothing (reference lib/utils.py:36); used to turn the
# 24x9 rotation-matrix pose into the 126-d condition (lib/utils.py:38-62).
USEFUL_JOINTS = (1, 2, 3, 4, 5, 6, 9, 12, 13, 14, 16, 17, 18, 19)


def filter_cloth_pose(pose_vec):
    """Keep the 14 clothing-related joints of a [n,72] pose / [n,216] rot-matrix array
    (reference lib/utils.py:38-62)."""
    n, dim = pose_vec.shape[0], pose_vec.shape[-1]
    if dim not in (72, 216):
        print('please provide either 72-dim pose vector or 216-dim rot matrix')
        return None
    per_joint = pose_vec.reshape(n, 24, dim // 24)
    return per_joint[:, list(USEFUL_JOINTS), :].reshape(n, -1)
