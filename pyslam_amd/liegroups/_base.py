"""Shared machinery for the matrix Lie groups shipped with pyslam_amd.

The reference depends on the third-party ``liegroups`` package (reference
setup.py:12, README.md:8; un-vendored, version unpinned, absent from this
image).  This module restates the published numpy-backend conventions of that
package for the surface the hot path uses:

* twist ordering ``xi = [rho (translation); phi (rotation)]``;
* left perturbation ``T <- exp(xi) . T``;
* ``np.isclose(angle, 0.)`` small-angle switches (|angle| <= 1e-8).

Element-level parity of exp/log is pinned independently of the reference in
tests/test_liegroups.py (scipy.linalg.expm / logm known answers).
"""
import numpy as np

SMALL_ANGLE = 1e-8  # np.isclose(x, 0.) == (|x| <= 1e-8): rtol term vanishes at 0


def is_small(angle):
    return np.abs(angle) <= SMALL_ANGLE


class MatrixGroup:
    """Behaviour common to SO(n) and SE(n)."""

    dof = None
    dim = None

    # -- algebra helpers every concrete group provides -------------------
    @classmethod
    def identity(cls):
        raise NotImplementedError

    @classmethod
    def exp(cls, xi):
        raise NotImplementedError

    def log(self):
        raise NotImplementedError

    def inv(self):
        raise NotImplementedError

    def as_matrix(self):
        raise NotImplementedError

    def dot(self, other):
        raise NotImplementedError

    # reference example stereo_ba_frame_to_frame.py:40,70,84 writes ``T * p``
    def __mul__(self, other):
        return self.dot(other)

    def __repr__(self):
        return "<{}.{}>\n{}".format(
            self.__class__.__module__, self.__class__.__name__,
            self.as_matrix()).replace("\n", "\n| ")


def _project_to_so(mat):
    """Nearest proper rotation in the Frobenius sense (SVD projection)."""
    n = mat.shape[0]
    u, _, vt = np.linalg.svd(mat, full_matrices=False)
    fix = np.identity(n)
    fix[n - 1, n - 1] = np.linalg.det(u) * np.linalg.det(vt)
    return u.dot(fix).dot(vt)


def _looks_like_rotation(mat, n):
    return (mat.shape == (n, n)
            and np.isclose(np.linalg.det(mat), 1.)
            and np.allclose(mat.T.dot(mat), np.identity(n)))
