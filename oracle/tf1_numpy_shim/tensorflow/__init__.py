"""A numpy implementation of the small TF1 API surface that the reference's lib/models.py,
lib/losses.py and lib/utils.py touch when building and evaluating the CAPE graph in the 'demo'
phase.  TEST INFRASTRUCTURE ONLY -- it exists so that oracle/make_golden.py can execute the
reference's OWN graph-assembly code (layer order, indices, weight layouts, variable names) and
record golden vectors, since TensorFlow 1.13 cannot be installed here.

Semantics follow the TF1 documentation of each op; graphs are lazy (a node stores a numpy
function of its inputs) and every node is also evaluated once at construction time on zero
placeholders, which provides the static shapes the reference reads with get_shape().
Variables are initialised by oracle/weights.py keyed by their full scoped name.
Not implemented: gradients / optimizers (tf.train.*Optimizer) -- build the graph with
phase='demo'.
"""
import contextlib
import os
import sys

import numpy as np
import scipy.sparse as _sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import weights as _winit  # noqa: E402

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64
bool = np.bool_  # noqa: A001

_STATE = {"seed": 123, "dtype": np.float64, "eps": None, "var_scope": [], "reuse": [False], "vars": {},
          "kinds": {}, "reg_losses": []}


def shim_configure(seed=123, compute_dtype=np.float64, eps=None):
    _STATE.update(seed=seed, dtype=np.dtype(compute_dtype), eps=eps)


def shim_reset():
    _STATE.update(var_scope=[], reuse=[False], vars={}, kinds={}, reg_losses=[])


def shim_variables():
    return {k: v._value for k, v in _STATE["vars"].items()}


# ------------------------------------------------------------------------------------------------
class _Dim(int):
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    def as_list(self):
        return [int(d) for d in self]


class _Op(object):
    def __init__(self, name):
        self.name = name


class Tensor(object):
    __array_priority__ = 100

    def __init__(self, fn, inputs, name="t"):
        self._fn, self._inputs, self.name = fn, list(inputs), name
        self.op = _Op(name)
        self._value = None if fn is None else fn(*[_val(i) for i in self._inputs])

    def get_shape(self):
        return TensorShape(_Dim(d) for d in np.shape(self._value))

    @property
    def shape(self):
        return self.get_shape()

    @property
    def dtype(self):
        return np.asarray(self._value).dtype

    def __add__(self, o): return _binary(np.add, self, o)
    def __radd__(self, o): return _binary(np.add, o, self)
    def __sub__(self, o): return _binary(np.subtract, self, o)
    def __rsub__(self, o): return _binary(np.subtract, o, self)
    def __mul__(self, o): return _binary(np.multiply, self, o)
    def __rmul__(self, o): return _binary(np.multiply, o, self)
    def __truediv__(self, o): return _binary(np.divide, self, o)
    def __rtruediv__(self, o): return _binary(np.divide, o, self)
    def __neg__(self): return Tensor(np.negative, [self])
    def __pos__(self): return self
    def __lt__(self, o): return _binary(np.less, self, o)


class Variable(Tensor):
    def __init__(self, initial_value=0, name="Variable", trainable=True, dtype=None):
        Tensor.__init__(self, None, [], name)
        self._value = np.asarray(initial_value)
        self.trainable = trainable


def _val(x):
    return x._value if isinstance(x, Tensor) else x


def _wrap(x):
    return x if isinstance(x, Tensor) else Tensor(lambda: np.asarray(x), [])


def _binary(fn, a, b):
    return Tensor(lambda x, y: fn(x, y), [a, b])


def _evaluate(t, memo):
    if not isinstance(t, Tensor):
        return t
    key = id(t)
    if key in memo:
        return memo[key]
    if t._fn is None:
        out = t._value
    else:
        out = t._fn(*[_evaluate(i, memo) for i in t._inputs])
    memo[key] = out
    return out


# ------------------------------------------------------------------------------------------------
class Graph(object):
    @contextlib.contextmanager
    def as_default(self):
        yield self

    def get_tensor_by_name(self, name):
        return _STATE["vars"][name.split(":")[0]]


class Session(object):
    def __init__(self, graph=None):
        self.graph = graph

    def run(self, fetches, feed_dict=None):
        memo = {}
        for k, v in (feed_dict or {}).items():
            dt = _STATE["dtype"] if np.asarray(k._value).dtype.kind == "f" else np.asarray(k._value).dtype
            memo[id(k)] = np.asarray(v).astype(dt)
        if isinstance(fetches, (list, tuple)):
            return [_evaluate(f, memo) for f in fetches]
        return _evaluate(fetches, memo)

    def close(self):
        pass


def set_random_seed(seed):
    _STATE["seed"] = seed


class _Random(object):
    set_random_seed = staticmethod(set_random_seed)


random = _Random()


@contextlib.contextmanager
def name_scope(name):
    yield


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _STATE["var_scope"].append(name)
    _STATE["reuse"].append(_STATE["reuse"][-1] or builtins_bool(reuse))
    try:
        yield
    finally:
        _STATE["var_scope"].pop()
        _STATE["reuse"].pop()


def builtins_bool(x):
    return x is True or x == 1


@contextlib.contextmanager
def control_dependencies(deps):
    yield


def placeholder(dtype, shape=None, name=None):
    dt = _STATE["dtype"] if np.dtype(dtype).kind == "f" else np.dtype(dtype)
    t = Tensor(None, [], name or "placeholder")
    t._value = np.zeros(tuple(shape) if shape is not None else (), dtype=dt)
    return t


# ---- initialisers / variables -------------------------------------------------------------------
class truncated_normal_initializer(object):
    def __init__(self, mean=0.0, stddev=1.0):
        self.kind, self.kw, self.tag = "trunc_normal", dict(mean=mean, stddev=stddev), "conv"


class constant_initializer(object):
    def __init__(self, value=0.0):
        self.kind, self.kw = "const", dict(value=value)
        self.tag = "bias" if value == 0.1 else "gn"


class _Glorot(object):
    kind, kw, tag = "glorot_uniform", {}, "fc_kernel"


class _Zeros(object):
    kind, kw, tag = "zeros", {}, "fc_bias"


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    full = "/".join(_STATE["var_scope"] + [name])
    if full in _STATE["vars"]:
        v = _STATE["vars"][full]
        assert tuple(np.shape(v._value)) == tuple(int(s) for s in shape), (full, np.shape(v._value), shape)
        return v
    if _STATE["reuse"][-1]:
        raise ValueError("Variable %s does not exist (reuse=True)" % full)
    kind = initializer.kind
    if kind == "const" and initializer.kw["value"] == 1.0:
        kind = "ones"
    elif kind == "const" and initializer.kw["value"] == 0.0:
        kind = "zeros"
    arr = _winit.init_variable(kind, shape, _STATE["seed"], full, **initializer.kw)
    v = Variable(arr.astype(_STATE["dtype"]), name=full, trainable=trainable)
    v._store32 = arr
    _STATE["vars"][full] = v
    _STATE["kinds"][full] = initializer.tag
    return v


def trainable_variables():
    return [v for v in _STATE["vars"].values() if v.trainable]


def global_variables_initializer():
    return _wrap(0)


# ---- array ops ----------------------------------------------------------------------------------
def transpose(x, perm=None):
    return Tensor(lambda a: np.transpose(a, perm), [x])


def reshape(x, shape):
    shape = [int(s) for s in shape]
    return Tensor(lambda a: np.reshape(a, shape), [x])


def expand_dims(x, axis):
    return Tensor(lambda a: np.expand_dims(a, axis), [x])


def concat(values, axis):
    return Tensor(lambda *a: np.concatenate(a, axis=axis), list(values))


def identity(x, name=None):
    return Tensor(lambda a: a, [x], name or "identity")


def cast(x, dtype):
    return Tensor(lambda a: np.asarray(a).astype(_STATE["dtype"] if np.dtype(dtype).kind == "f" else dtype), [x])


def ones(shape, dtype=None):
    return Tensor(lambda: np.ones([int(s) for s in shape], dtype=_STATE["dtype"]), [])


def ones_like(x):
    return Tensor(lambda a: np.ones_like(a), [x])


def zeros_like(x):
    return Tensor(lambda a: np.zeros_like(a), [x])


def gather(x, indices, axis=0):
    idx = np.asarray(indices)
    return Tensor(lambda a: np.take(a, idx, axis=axis), [x])


def add(a, b): return _binary(np.add, a, b)
def multiply(a, b): return _binary(np.multiply, a, b)
def divide(a, b): return _binary(np.divide, a, b)
def sqrt(x): return Tensor(np.sqrt, [x])
def exp(x): return Tensor(np.exp, [x])
def square(x): return Tensor(np.square, [x])
def equal(a, b): return _binary(np.equal, a, b)


def reduce_sum(x, axis=None, keepdims=False):
    return Tensor(lambda a: np.sum(a, axis=axis, keepdims=keepdims), [x])


def reduce_mean(x, axis=None, keepdims=False):
    return Tensor(lambda a: np.mean(a, axis=axis, keepdims=keepdims), [x])


def norm(x, ord="euclidean", axis=None):
    assert ord in ("euclidean", 2)
    return Tensor(lambda a: np.sqrt(np.sum(a * a, axis=axis)), [x])


def matmul(a, b):
    return Tensor(lambda x, y: x @ y, [a, b])


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None):
    shape = [int(s) for s in shape]

    def draw():
        if _STATE["eps"] is not None:
            e = np.asarray(_STATE["eps"], dtype=_STATE["dtype"])
            assert list(e.shape) == shape, (e.shape, shape)
            return mean + stddev * e
        return (mean + stddev * np.random.standard_normal(shape)).astype(_STATE["dtype"])
    return Tensor(draw, [])


# ---- sparse -------------------------------------------------------------------------------------
class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = np.asarray(indices), np.asarray(values), tuple(dense_shape)


def sparse_reorder(sp_t):
    order = np.lexsort((sp_t.indices[:, 1], sp_t.indices[:, 0]))
    return SparseTensor(sp_t.indices[order], sp_t.values[order], sp_t.dense_shape)


def sparse_tensor_dense_matmul(sp_t, dense):
    m = _sp.csr_matrix((sp_t.values.astype(_STATE["dtype"]), (sp_t.indices[:, 0], sp_t.indices[:, 1])),
                       shape=sp_t.dense_shape)
    m.sort_indices()
    return Tensor(lambda a: m @ a, [dense])


# ---- nn / layers / losses -----------------------------------------------------------------------
class _NN(object):
    @staticmethod
    def leaky_relu(x, alpha=0.2):
        return Tensor(lambda a: np.maximum(a, alpha * a), [x])

    @staticmethod
    def relu(x):
        return Tensor(lambda a: np.maximum(a, 0), [x])

    @staticmethod
    def tanh(x):
        return Tensor(np.tanh, [x])

    @staticmethod
    def moments(x, axes, keep_dims=False):
        axes = tuple(axes)
        mean = Tensor(lambda a: np.mean(a, axis=axes, keepdims=keep_dims), [x])
        var = Tensor(lambda a: np.mean((a - np.mean(a, axis=axes, keepdims=True)) ** 2, axis=axes, keepdims=keep_dims), [x])
        return mean, var

    @staticmethod
    def sigmoid_cross_entropy_with_logits(logits=None, labels=None):
        return Tensor(lambda x, z: np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x))), [logits, labels])


nn = _NN()


class _Layers(object):
    @staticmethod
    def dense(x, units, activation=None, kernel_regularizer=None, trainable=True):
        with variable_scope("dense"):
            k = get_variable("kernel", [int(x.get_shape()[-1]), units], initializer=_Glorot(), trainable=trainable)
            b = get_variable("bias", [units], initializer=_Zeros(), trainable=trainable)
        if kernel_regularizer is not None:
            r = kernel_regularizer(k)
            if r is not None and not any(n == k.name for n, _ in _STATE["reg_losses"]):
                _STATE["reg_losses"].append((k.name, r))
        y = Tensor(lambda a, w, c: a @ w + c, [x, k, b])
        return activation(y) if activation is not None else y


layers = _Layers()


class _ContribLayers(object):
    @staticmethod
    def l2_regularizer(scale):
        if scale == 0.0:
            return lambda w: None
        return lambda w: Tensor(lambda a: scale * 0.5 * np.sum(a * a), [w])


class _Contrib(object):
    layers = _ContribLayers()


contrib = _Contrib()


class _Losses(object):
    class Reduction(object):
        MEAN = "mean"

    @staticmethod
    def absolute_difference(labels=None, predictions=None, weights=1.0, reduction=None):
        return Tensor(lambda p, l: np.mean(np.abs(p - l) * weights), [predictions, labels])

    @staticmethod
    def mean_squared_error(labels=None, predictions=None, weights=1.0, reduction=None):
        return Tensor(lambda p, l: np.mean((p - l) ** 2 * weights), [predictions, labels])

    @staticmethod
    def huber_loss(labels=None, predictions=None, weights=1.0, delta=1.0, reduction=None):
        def f(p, l):
            a = np.abs(p - l)
            return np.mean(np.where(a <= delta, 0.5 * a * a, delta * a - 0.5 * delta * delta) * weights)
        return Tensor(f, [predictions, labels])

    @staticmethod
    def get_regularization_loss(scope=None):
        terms = [t for n, t in _STATE["reg_losses"] if scope is None or n.startswith(scope)]
        if not terms:
            return _wrap(np.zeros((), dtype=_STATE["dtype"]))
        return Tensor(lambda *a: np.sum(a), terms)


losses = _Losses()


# ---- summaries / train stubs --------------------------------------------------------------------
class _Summary(object):
    @staticmethod
    def histogram(*a, **k): return None
    @staticmethod
    def scalar(*a, **k): return None
    @staticmethod
    def merge_all(): return _wrap(0)

    class FileWriter(object):
        def __init__(self, *a, **k): pass
        def add_summary(self, *a, **k): pass
        def close(self): pass


summary = _Summary()


class _EMA(object):
    def __init__(self, decay):
        self.decay, self.shadow = decay, {}

    def apply(self, tensors):
        for t in tensors:
            self.shadow[id(t)] = t
        return _wrap(0)

    def average(self, t):
        # zero-initialised shadow, one update: (1 - decay) * value
        return Tensor(lambda a: (1 - self.decay) * a, [t])


class _Saver(object):
    def __init__(self, *a, **k): pass
    def save(self, *a, **k): return None
    def restore(self, *a, **k): return None


class _Train(object):
    ExponentialMovingAverage = _EMA
    Saver = _Saver

    @staticmethod
    def latest_checkpoint(path):
        return None


train = _Train()
