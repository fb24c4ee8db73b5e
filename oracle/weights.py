"""Deterministic, name-keyed variable initialisers shared by the oracle, the TF1 numpy
shim (oracle/tf1_numpy_shim) and the tests.  TEST INFRASTRUCTURE ONLY.

The reference initialises with ``tf.truncated_normal_initializer(0, 0.1)`` for graph-conv
weights (lib/models.py:217-221), ``tf.constant_initializer(0.1)`` for conv biases
(:223-227), and ``tf.layers.dense`` defaults (glorot-uniform kernel, zero bias; :496-510,
:557-560, :582), gamma=1 / beta=0 for group-norm (:702-703).  TF's RNG streams cannot be
reproduced without TF, so every variable is drawn from a numpy Generator seeded with
``(seed, crc32(variable_name))``: any two implementations that agree on the *names and
shapes*  get bit-identical weights, independent of creation order.
Values are produced in float64 and rounded once to float32.
"""
import contextlib
import zlib
import numpy as np

# Optional weight PROFILE (test cases only; None = the reference initialisers as they are).  "range": the variables are
# arranged so that activation rows span more than twenty binades inside the network, with exactly-zero rows among them --
# the operand-range cases of the fp16 two-piece contractions (cape_amd/csrc/gemm_h2.h) at model level:
#   * conv biases (reference: constant 0.1, lib/models.py:223-227) are zero, so a row's magnitude follows its inputs all the
#     way down instead of being floored at 0.1, and a zero input neighbourhood gives an exactly-zero row;
#   * the columns of the decoder's dense kernel (generator/decoder/fc1, [nz_total, 862 * C], vertex-major) are scaled by a
#     per-vertex field (``field``: one factor per coarsest-level vertex), so the decoder's rows span the same range.
# The profile is part of a golden case's recorded configuration: oracle/make_golden.py, the oracle, the twin and the HIP model
# (which loads the oracle's variables) all see the same arrays.
PROFILE = None


@contextlib.contextmanager
def profile(kind, field=None):
    global PROFILE
    old, PROFILE = PROFILE, (None if kind is None else dict(kind=kind, field=None if field is None else np.asarray(field, np.float64)))
    try:
        yield
    finally:
        PROFILE = old


def _apply_profile(arr, kind, name, kw):
    if PROFILE is None or PROFILE["kind"] != "range":
        return arr
    if kind == "const" and name.endswith("/bias") and kw.get("value") == 0.1:
        return np.zeros_like(arr)
    if name.endswith("decoder/fc1/dense/kernel") and PROFILE["field"] is not None:
        f = PROFILE["field"]
        assert arr.shape[1] % f.size == 0, (arr.shape, f.size)
        col = np.repeat(f, arr.shape[1] // f.size)               # flatten order of [M, C] is vertex-major (lib/models.py:554, :584)
        return (arr.astype(np.float64) * col[None, :]).astype(np.float32)
    return arr


def _rng(seed, name):
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode("utf-8"))])


def truncated_normal(shape, stddev, seed, name, mean=0.0):
    """Normal(mean, stddev) re-drawn until within 2 stddev (TF truncated_normal semantics)."""
    rng = _rng(seed, name)
    n = int(np.prod(shape))
    out = rng.standard_normal(n)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (mean + stddev * out).reshape(shape).astype(np.float32)


def glorot_uniform(shape, seed, name):
    fan_in, fan_out = int(shape[0]), int(shape[1])
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return _rng(seed, name).uniform(-limit, limit, size=shape).astype(np.float32)


def constant(shape, value):
    return np.full(shape, value, dtype=np.float32)


def init_variable(kind, shape, seed, name, **kw):
    return _apply_profile(_init_variable(kind, shape, seed, name, **kw), kind, name, kw)


def _init_variable(kind, shape, seed, name, **kw):
    shape = tuple(int(s) for s in shape)
    if kind == "trunc_normal":
        return truncated_normal(shape, kw.get("stddev", 0.1), seed, name, kw.get("mean", 0.0))
    if kind == "glorot_uniform":
        return glorot_uniform(shape, seed, name)
    if kind == "const":
        return constant(shape, kw["value"])
    if kind == "zeros":
        return constant(shape, 0.0)
    if kind == "ones":
        return constant(shape, 1.0)
    raise ValueError("unknown initialiser kind %r" % (kind,))
