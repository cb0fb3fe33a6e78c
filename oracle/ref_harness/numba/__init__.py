"""Decorator-only stand-in for ``numba`` -- TEST INFRASTRUCTURE, authoring
container only (never shipped, never imported by the product).

The reference imports ``numba`` for its @vectorize/@guvectorize kernels
(reference pyslam/losses.py:2, sensors/stereo_camera.py:2, utils.py:3) and
numba is not installed in this image.  This module supplies *only the
decorators*: the kernel bodies that run are the reference's own Python
functions, looped over by numpy.  It lets oracle/gen_golden.py import the
verbatim reference from /root/reference to produce tests/golden/*.npz.
"""
import re

import numpy as np


class _Type:
    """float32 / float64 / boolean: subscriptable (``float64[:, :]``) and
    callable (``float64(float64, float64)``), carrying only a numpy dtype."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __getitem__(self, item):
        return self

    def __call__(self, *args):
        return (self,) + tuple(args)


float32 = _Type(np.float64)   # the oracle runs everything in fp64
float64 = _Type(np.float64)
boolean = _Type(np.bool_)


def vectorize(signatures=None, **_kw):
    def wrap(fn):
        return np.vectorize(fn, otypes=[float])
    return wrap


def _parse_layout(layout):
    lhs, rhs = layout.replace(' ', '').split('->')
    dims = lambda s: [tuple(d for d in grp.split(',') if d)
                      for grp in re.findall(r'\(([^)]*)\)', s)]
    return dims(lhs), dims(rhs)[0]


def guvectorize(signatures, layout, **_kw):
    in_dims, out_dims = _parse_layout(layout)
    out_dtype = signatures[0][-1].dtype

    def wrap(fn):
        def call(*args):
            arrs = [np.asarray(a, dtype=float) for a in args]
            sizes = {}
            loop_shapes = []
            for a, dims in zip(arrs, in_dims):
                nd = len(dims)
                core = a.shape[a.ndim - nd:] if nd else ()
                for name, n in zip(dims, core):
                    sizes.setdefault(name, n)
                loop_shapes.append(a.shape[:a.ndim - nd])
            loop = np.broadcast_shapes(*loop_shapes)
            core_out = tuple(sizes[d] for d in out_dims) or (1,)
            out = np.zeros(loop + core_out, dtype=out_dtype)
            views = []
            for a, dims in zip(arrs, in_dims):
                nd = len(dims)
                core = a.shape[a.ndim - nd:] if nd else ()
                b = np.broadcast_to(a, loop + core)
                views.append(b if nd else b[..., None])
            for idx in np.ndindex(*loop):
                fn(*[v[idx] for v in views], out[idx])
            return out if out_dims else out[..., 0]
        call.__name__ = fn.__name__
        return call
    return wrap
