"""Seeded inputs shared by oracle/make_golden.py (which runs the reference code) and the tests that
replay the same inputs through the oracle / the HIP path.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import scipy.sparse as sp


CLOTH_JOINTS = (1, 2, 3, 4, 5, 6, 9, 12, 13, 14, 16, 17, 18, 19)     # SMPL joints that move the clothing (lib/utils.py:36)


def filter_cloth_pose(pose):
    """Restatement of the reference's lib/utils.py:38-62 for rotation-matrix input ([n, 216] -> [n, 126]: the 14 clothing
    joints' 3x3 blocks); oracle/make_golden.py cross-checks it against the reference's own function."""
    pose = np.asarray(pose)
    return pose.reshape(pose.shape[0], 24, -1)[:, list(CLOTH_JOINTS), :].reshape(pose.shape[0], -1)


def range_fields(template_verts, D):
    """Per-vertex scale fields of the "range" profile (oracle/weights.py): smooth over the mesh (a function of the template's
    height), so that neighbouring rows -- which every graph convolution mixes -- stay within a binade or two of each other
    and the spread survives the Laplacian.  Returns (input field [6890]: 2^-16 .. 2^6 with an exactly-zero region at the
    feet, decoder field [coarsest level]: 2^-18 .. 2^2)."""
    v = np.asarray(template_verts, np.float64)
    h = (v[:, 1] - v[:, 1].min()) / (v[:, 1].max() - v[:, 1].min())
    f_in = np.exp2(-16.0 + 22.0 * h)
    f_in[h < 0.06] = 0.0
    hc = h
    for d in D:                                   # row selections down to the coarsest level (lib/mesh_sampling.py:111-160)
        hc = sp.csr_matrix(d).astype(np.float64) @ hc
    return f_in, np.exp2(-18.0 + 20.0 * hc)


def golden_inputs(N, nz, seed, demo_rot, in_field=None):
    """demo_rot: the 'rot' array of the reference's data/demo_data/demo_pose_params.npz ([6,216]); in_field: per-vertex scale
    of the displacements (the "range" cases)."""
    rng = np.random.default_rng(seed)
    r32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    x = r32(N, 6890, 3)
    d = dict(x=x, gt=(x + 0.1 * r32(N, 6890, 3)).astype(np.float32), xd=r32(N, 6890, 3),
             clo=np.eye(4, dtype=np.float32)[np.arange(N) % 4], clo_d=np.eye(4, dtype=np.float32)[(np.arange(N) + 1) % 4],
             eps=r32(N, nz), cond_d=(0.5 * r32(N, 126)).astype(np.float32))
    d["cond"] = np.tile(filter_cloth_pose(np.asarray(demo_rot)), (N // 6 + 1, 1))[:N].astype(np.float32)  # demos.py:367-376
    if in_field is not None:
        f = np.asarray(in_field, np.float32)[None, :, None]
        for k in ("x", "gt", "xd"):
            d[k] = (d[k] * f).astype(np.float32)
    return d
