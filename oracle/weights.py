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
import zlib
import numpy as np


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
