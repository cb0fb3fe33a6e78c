"""Small numeric helpers (reference pyslam/utils.py:8-13, 16-75, 80-89)."""
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


def bilinear_interpolate(im, x, y):
    """Bilinear lookup of ``im`` (h, w[, channels]) at real-valued pixel coordinates x (columns), y (rows).

    What the body of the reference's ``_bilinear_interpolate`` computes (pyslam/utils.py:27-75): corner
    indices by truncation, the four weights from the UNCLIPPED corners, then the corners clamped to the
    image (equivalent to repeating the border rows / columns).  The committed reference reads ``x[1]``,
    ``y[1]`` and writes ``out[1]`` of one-element arrays (:36-37, :75) -- out of bounds, and its own test
    fails -- so this is the function as evidently intended, ``x[0]``, ``y[0]``, ``out[0]``;
    the device restatement is ``photo_bilinear`` in csrc/ps_photo.h.
    """
    im = np.atleast_3d(np.asarray(im, dtype=float))
    x = np.atleast_1d(np.asarray(x, dtype=float))
    y = np.atleast_1d(np.asarray(y, dtype=float))
    h, w = im.shape[0], im.shape[1]
    finite = np.isfinite(x) & np.isfinite(y)
    xs, ys = np.where(finite, x, 0.), np.where(finite, y, 0.)
    x0 = np.clip(np.trunc(xs), -2.**31, 2.**31 - 2).astype(np.int64)
    y0 = np.clip(np.trunc(ys), -2.**31, 2.**31 - 2).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    wa, wb = (x1 - xs) * (y1 - ys), (x1 - xs) * (ys - y0)
    wc, wd = (xs - x0) * (y1 - ys), (xs - x0) * (ys - y0)
    x0, x1 = np.clip(x0, 0, w - 1), np.clip(x1, 0, w - 1)
    y0, y1 = np.clip(y0, 0, h - 1), np.clip(y1, 0, h - 1)
    out = (wa[:, None] * im[y0, x0] + wb[:, None] * im[y1, x0] + wc[:, None] * im[y0, x1] + wd[:, None] * im[y1, x1])
    out[~finite] = np.nan
    return np.squeeze(out)
