"""Synthetic stand-in for the CAPE dataset files that main.py's BodyData loads (reference lib/load_data.py:35-103; the real data
is licensed and not shipped): seeded displacement fields on the SMPL topology, rotation-matrix pose conditions (24 joints x 9,
valid rotations) and one-hot clothing types.  Used by BOTH halves of the train-mode entry-script test -- the CPU run of the
unmodified main.py writes these arrays as the .npy files it expects, the device replay rebuilds the same arrays -- so the two
halves see identical data without a large fixture.  TEST INFRASTRUCTURE ONLY."""
import numpy as np

N_TRAIN, N_VAL, N_TEST = 32, 100, 4           # BodyData(nVal=100) splits the last 100 training examples off (main.py:21)


def _rotations(rng, n):
    ax = rng.standard_normal((n, 24, 3))
    ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
    th = rng.uniform(-0.6, 0.6, size=(n, 24, 1, 1))
    K = np.zeros((n, 24, 3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -ax[..., 2], ax[..., 1], ax[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -ax[..., 0], -ax[..., 1], ax[..., 0]
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    return R.reshape(n, 216)


def arrays(seed=20):
    """dict of the six arrays main.py reads: train_disp [132, 6890, 3], train_rot [132, 216], train_clo_label [132, 4], test_*."""
    rng = np.random.default_rng(seed)
    out = {}
    for split, n in (("train", N_TRAIN + N_VAL), ("test", N_TEST)):
        smooth = np.cumsum(rng.standard_normal((n, 6890, 3)) * 1e-3, axis=1)          # spatially correlated, metres
        out[split + "_disp"] = smooth - smooth.mean(axis=1, keepdims=True)
        out[split + "_rot"] = _rotations(rng, n)
        out[split + "_clo_label"] = np.eye(4)[rng.integers(0, 4, size=n)]
    return out


CLOTH_JOINTS = [1, 2, 3, 4, 5, 6, 9, 12, 13, 14, 16, 17, 18, 19]       # the 14 clothing-related SMPL joints (lib/utils.py:38)


class Wrapper(object):
    """The fields BodyData holds after load / normalize / change_dtype (lib/load_data.py:35-133), rebuilt from ``arrays()``
    without the files.  The CPU half of the test asserts these equal what the reference's own BodyData produced."""

    def __init__(self, seed=20):
        a = arrays(seed)
        cut = lambda v: v.reshape(len(v), -1, 9)[:, CLOTH_JOINTS, :].reshape(len(v), -1)
        vt, ct, lt = a["train_disp"], a["train_rot"], a["train_clo_label"]
        self.vertices_train, self.vertices_val, self.vertices_test = vt[:-N_VAL].copy(), vt[-N_VAL:].copy(), a["test_disp"].copy()
        self.cond1_train_full, self.cond1_val_full, self.cond1_test_full = ct[:-N_VAL], ct[-N_VAL:], a["test_rot"]
        self.cond1_train, self.cond1_val, self.cond1_test = cut(ct[:-N_VAL]), cut(ct[-N_VAL:]), cut(a["test_rot"])
        self.cond2_train, self.cond2_val, self.cond2_test = lt[:-N_VAL], lt[-N_VAL:], a["test_clo_label"]
        self.n_vertex = self.vertices_train.shape[1]
        self.mean, self.std = np.mean(self.vertices_train, axis=0), np.std(self.vertices_train, axis=0)
        for k in ("vertices_train", "vertices_val", "vertices_test"):
            v = getattr(self, k)
            v -= self.mean
            v /= self.std
        for k in ("vertices", "cond1", "cond2"):
            for part in ("train", "val", "test"):
                setattr(self, "%s_%s" % (k, part), getattr(self, "%s_%s" % (k, part)).astype("float32"))

    FIELDS = tuple("%s_%s" % (k, part) for k in ("vertices", "cond1", "cond2") for part in ("train", "val", "test")) + \
        ("cond1_test_full", "mean", "std")


def summary(a):
    """Machine-independent fingerprint of an array (the two halves of the test run on different hosts, and float reductions
    may differ in the last bit between CPU generations): shape, dtype, float64 sum / sum of squares, 64 strided samples."""
    a = np.asarray(a)
    flat = a.reshape(-1).astype(np.float64)
    return dict(shape=list(a.shape), dtype=str(a.dtype), sum=float(flat.sum()), sumsq=float((flat * flat).sum()),
                sample=[float(x) for x in flat[::max(1, flat.size // 64)][:64]])


def summaries_match(got, want, rtol=1e-5):
    if got["shape"] != want["shape"] or got["dtype"] != want["dtype"]:
        return False
    scale = max(abs(want["sumsq"]), 1.0) ** 0.5
    return (abs(got["sum"] - want["sum"]) <= rtol * scale * max(1.0, np.prod(want["shape"]) ** 0.5)
            and abs(got["sumsq"] - want["sumsq"]) <= rtol * max(abs(want["sumsq"]), 1.0)
            and np.allclose(got["sample"], want["sample"], rtol=rtol, atol=1e-6))
