"""Stand-in for the one cv2 call the reference makes (lib/utils.py:104, rot2pose: cv2.Rodrigues on 3 x 3 rotation matrices,
used by demos.py:102 when the pose condition is stored as rotation matrices).  TEST INFRASTRUCTURE ONLY."""
import numpy as np


def Rodrigues(R, *a, **k):
    """Rotation matrix -> axis-angle vector (3 x 1), Jacobian None: the part of cv2.Rodrigues the reference uses (``[0]``)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(w)
    if s < 1e-12:
        if c > 0:
            return np.zeros((3, 1)), None
        ax = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))       # theta = pi: axis from the diagonal
        return (np.pi * ax).reshape(3, 1), None
    return (w / s * theta).reshape(3, 1), None
