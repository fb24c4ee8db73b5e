"""Seeded inputs shared by oracle/make_golden.py (which runs the reference code) and the tests that
replay the same inputs through the oracle / the HIP path.  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from cape_amd.load_data import filter_cloth_pose


def golden_inputs(N, nz, seed, demo_rot):
    """demo_rot: the 'rot' array of the reference's data/demo_data/demo_pose_params.npz ([6,216])."""
    rng = np.random.default_rng(seed)
    r32 = lambda *s: rng.standard_normal(s).astype(np.float32)
    x = r32(N, 6890, 3)
    d = dict(x=x, gt=(x + 0.1 * r32(N, 6890, 3)).astype(np.float32), xd=r32(N, 6890, 3),
             clo=np.eye(4, dtype=np.float32)[np.arange(N) % 4], clo_d=np.eye(4, dtype=np.float32)[(np.arange(N) + 1) % 4],
             eps=r32(N, nz), cond_d=(0.5 * r32(N, 126)).astype(np.float32))
    d["cond"] = np.tile(filter_cloth_pose(np.asarray(demo_rot)), (N // 6 + 1, 1))[:N].astype(np.float32)  # demos.py:367-376
    return d
