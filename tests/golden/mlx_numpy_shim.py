"""numpy-backed stand-in for the ~45 `mlx.core` / `mlx.nn` / `mlx.optimizers` names the reference's
hot path uses (SURVEY.md 8(c) "optional stronger oracle").  BUILD CONTAINER ONLY, test infrastructure.

Purpose: let the reference's OWN source files (rendering/render.py, models/embedding.py,
models/NeRF.py, encoding/sinusoidal.py, encoding/spherical_harmonics.py, ops/pose.py, ops/metric.py,
sampling/uniform.py, sampling/linear_disparity.py, rendering/ray.py:ndc_rays) execute unchanged in a
container where `mlx==0.7.0` (Apple-only wheel) cannot be installed, so that
tests/golden/make_golden_mx.py can record their inputs and outputs as fixtures.

What this pins: the reference's op ORDER, tensor SHAPES, broadcasting, concatenation orders and
quirks (k^2 frequency bands, un-ReLU'd exclusive cumsum, no sigmoid, [B,n,1] weights, ...), i.e. our
READING of the reference.  What it does NOT pin: MLX's own float behaviour (its GEMM summation order,
its sin/cos/exp implementations, its RNG) -- every function below is float32 numpy.

Nothing here is reference code: each function is a one-line numpy equivalent of the documented
MLX API of the same name.  The shim and the generator are listed in .gpurunignore; only the .npz/.json
outputs are used by the tests.
"""
import sys
import types

import numpy as np

float32 = np.float32
int32 = np.int32
pi = float(np.pi)


def _dt(obj, dtype):
    if dtype is not None:
        return dtype
    a = np.asarray(obj)
    if a.dtype.kind == "f":
        return np.float32           # MLX has no float64 on device: doubles become float32
    if a.dtype.kind in "iu":
        return np.int32 if a.dtype.itemsize > 4 else a.dtype
    return a.dtype


class array(np.ndarray):
    """mx.array(obj, dtype=None): python floats / float64 -> float32, python ints -> int32."""

    def __new__(cls, obj, dtype=None):
        return np.array(obj, dtype=_dt(obj, dtype)).view(cls)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kw):
        """MLX type promotion has no float64: int32 (op) float32 -> float32, where numpy would give float64."""
        ins = [np.asarray(i) if isinstance(i, np.ndarray) else i for i in inputs]
        if out is not None:
            kw["out"] = tuple(np.asarray(o) for o in out)
        r = getattr(ufunc, method)(*ins, **kw)
        if out is not None:
            return out[0] if len(out) == 1 else out
        fix = lambda t: _w(t) if isinstance(t, np.ndarray) else (np.float32(t) if isinstance(t, np.float64) else t)
        return tuple(fix(t) for t in r) if isinstance(r, tuple) else fix(r)


def _w(x):
    x = np.asarray(x)
    if x.dtype == np.float64:
        x = x.astype(np.float32)
    return x.view(array)


def eval(*a, **k):                                   # noqa: A001  (lazy evaluation barrier: nothing to do)
    return None


def linspace(start, stop, num=50, dtype=np.float32):
    # mlx 0.7.0 ops.cpp: arange(0, num) * ((stop - start) / (num - 1)) + start, in float32
    step = np.float32((stop - start) / (num - 1))
    return _w(np.arange(num, dtype=np.float32) * step + np.float32(start))


def arange(*a, dtype=None):
    r = np.arange(*a)
    return _w(r.astype(dtype if dtype is not None else (np.int32 if r.dtype.kind in "iu" else np.float32)))


def zeros(shape, dtype=np.float32):
    return _w(np.zeros(shape, dtype=dtype))


def ones(shape, dtype=np.float32):
    return _w(np.ones(shape, dtype=dtype))


def zeros_like(x):
    return _w(np.zeros_like(np.asarray(x)))


def ones_like(x):
    return _w(np.ones_like(np.asarray(x)))


def concatenate(arrays, axis=0):
    return _w(np.concatenate([np.asarray(a) for a in arrays], axis=axis))


def stack(arrays, axis=0):
    return _w(np.stack([np.asarray(a) for a in arrays], axis=axis))


def split(x, indices_or_sections, axis=0):
    return [_w(p) for p in np.split(np.asarray(x), indices_or_sections, axis=axis)]


def reshape(x, shape):
    return _w(np.reshape(np.asarray(x), tuple(shape)))


def flatten(x, start_axis=0, end_axis=-1):
    x = np.asarray(x)
    nd = x.ndim
    s, e = start_axis % nd, end_axis % nd
    return _w(x.reshape(x.shape[:s] + (-1,) + x.shape[e + 1:]))


def expand_dims(x, axis):
    return _w(np.expand_dims(np.asarray(x), axis))


def repeat(x, repeats, axis=None):
    return _w(np.repeat(np.asarray(x), repeats, axis=axis))


def take(x, indices, axis=None):
    return _w(np.take(np.asarray(x), np.asarray(indices), axis=axis))


def where(c, a, b):
    return _w(np.where(np.asarray(c), a, b))


def _un(f):
    def g(x, *a, **k):
        x = np.asarray(x)
        if x.dtype.kind != "f":
            x = x.astype(np.float32)
        return _w(f(x, *a, **k))
    return g


sin, cos, exp, log, log10, floor, ceil, sqrt, abs = (_un(f) for f in (np.sin, np.cos, np.exp, np.log, np.log10,
                                                                       np.floor, np.ceil, np.sqrt, np.abs))


def maximum(a, b):
    return _w(np.maximum(a, b))


def minimum(a, b):
    return _w(np.minimum(a, b))


def clip(x, lo, hi):
    return _w(np.clip(x, lo, hi))


def sum(x, axis=None, keepdims=False):               # noqa: A001
    return _w(np.sum(np.asarray(x), axis=axis, keepdims=keepdims, dtype=np.asarray(x).dtype))


def mean(x, axis=None, keepdims=False):
    return _w(np.mean(np.asarray(x), axis=axis, keepdims=keepdims, dtype=np.asarray(x).dtype))


def max(x, axis=None, keepdims=False):               # noqa: A001
    return _w(np.max(np.asarray(x), axis=axis, keepdims=keepdims))


def min(x, axis=None, keepdims=False):               # noqa: A001
    return _w(np.min(np.asarray(x), axis=axis, keepdims=keepdims))


def cumsum(x, axis=None):
    x = np.asarray(x)
    return _w(np.cumsum(x, axis=axis, dtype=x.dtype))     # sequential float32 running sum


def sort(x, axis=-1):
    return _w(np.sort(np.asarray(x), axis=axis))


class _Linalg:
    @staticmethod
    def norm(x, ord=None, axis=None, keepdims=False):     # noqa: A002
        return _w(np.linalg.norm(np.asarray(x), ord=ord, axis=axis, keepdims=keepdims))


linalg = _Linalg()


class _Random:
    """Seedable stand-in for mx.random: the VALUES are numpy's, not MLX's (the reference leaves every stream
    unseeded, so no fixture depends on MLX's generator)."""

    def __init__(self):
        self.rng = np.random.default_rng(0)

    def seed(self, s):
        self.rng = np.random.default_rng(s)

    def uniform(self, low=0.0, high=1.0, shape=()):
        return _w((low + (high - low) * self.rng.random(size=tuple(shape))).astype(np.float32))

    def normal(self, shape=(), loc=0.0, scale=1.0):
        return _w((loc + scale * self.rng.standard_normal(size=tuple(shape))).astype(np.float32))


random = _Random()


class no_grad:
    def __call__(self, f):
        return f


# ------------------------------------------------------------------------------------------- mlx.nn

class Module:
    """mlx.nn.Module as far as the hot path needs it: attribute container with .parameters()."""

    def parameters(self):
        out = {}
        for k, v in vars(self).items():
            if isinstance(v, Module):
                out[k] = v.parameters()
            elif isinstance(v, list) and v and all(isinstance(e, Module) for e in v):
                out[k] = [e.parameters() for e in v]
            elif isinstance(v, np.ndarray):
                out[k] = v
        return out

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


class Linear(Module):
    """mlx.nn.Linear(input_dims, output_dims): weight [out, in], bias [out], both U(-1/sqrt(in), 1/sqrt(in));
    __call__ = x @ weight.T + bias."""
    _init_rng = np.random.default_rng(0)

    def __init__(self, input_dims, output_dims, bias=True):
        k = 1.0 / np.sqrt(input_dims)
        self.weight = _w(Linear._init_rng.uniform(-k, k, size=(output_dims, input_dims)).astype(np.float32))
        if bias:
            self.bias = _w(Linear._init_rng.uniform(-k, k, size=(output_dims,)).astype(np.float32))

    def __call__(self, x):
        y = np.asarray(x) @ np.asarray(self.weight).T
        if hasattr(self, "bias"):
            y = y + np.asarray(self.bias)
        return _w(y)


class Identity(Module):
    def __call__(self, x):
        return x


def relu(x):
    return _w(np.maximum(np.asarray(x), 0))


class _AdamPlaceholder:
    """create_NeRF constructs an optimiser (models/NeRF.py:120); its arithmetic lives in MLX and is NOT emulated."""

    def __init__(self, learning_rate, betas=(0.9, 0.999), eps=1e-8):
        self.learning_rate, self.betas, self.eps = learning_rate, betas, eps


def install():
    """Register the shim as `mlx`, `mlx.core`, `mlx.nn`, `mlx.optimizers` in sys.modules."""
    me = sys.modules[__name__]
    mlx = types.ModuleType("mlx")
    core = types.ModuleType("mlx.core")
    for k in dir(me):
        if not k.startswith("_") and k not in ("Module", "Linear", "Identity", "relu", "install", "np", "sys", "types"):
            setattr(core, k, getattr(me, k))
    core.optimizers = types.ModuleType("mlx.core.optimizers")
    nn = types.ModuleType("mlx.nn")
    nn.Module, nn.Linear, nn.Identity, nn.relu = Module, Linear, Identity, relu
    optim = types.ModuleType("mlx.optimizers")
    optim.Adam = _AdamPlaceholder
    mlx.core, mlx.nn, mlx.optimizers = core, nn, optim
    sys.modules.update({"mlx": mlx, "mlx.core": core, "mlx.nn": nn, "mlx.optimizers": optim})
    return core, nn
