"""Small numeric helpers (reference pyslam/utils.py:8-13, 80-89).

``bilinear_interpolate`` belongs to the photometric (dense image alignment)
path, which is out of scope for this build.
"""
import numpy as np
import scipy.linalg


def invsqrt(x):
    """Inverse square root of a scalar, or inv(sqrtm(M)) of a square matrix."""
    if hasattr(x, 'shape'):
        return np.linalg.inv(scipy.linalg.sqrtm(x))
    return 1. / np.sqrt(x)


def stackmul(A, B):
    """Multiply stacks of small matrices: (...,n,m) x (...,m,p) -> (...,n,p)."""
    return np.matmul(A, B)
