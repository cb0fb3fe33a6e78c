"""SO(2): planar rotations (dof 1, 2x2 matrices)."""
import numpy as np

from ._base import MatrixGroup, is_small, _project_to_so, _looks_like_rotation

_J = np.array([[0., -1.], [1., 0.]])  # generator: wedge(1)


class SO2(MatrixGroup):
    dof = 1
    dim = 2

    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=float)

    # ---- construction ---------------------------------------------------
    @classmethod
    def identity(cls):
        return cls(np.identity(2))

    @classmethod
    def from_angle(cls, theta):
        c, s = np.cos(theta), np.sin(theta)
        return cls(np.array([[c, -s], [s, c]]))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        mat = np.asarray(mat, dtype=float)
        if not _looks_like_rotation(mat, 2):
            if not normalize:
                raise ValueError("Invalid rotation matrix. Use normalize=True to handle rounding errors.")
            mat = _project_to_so(mat)
        return cls(mat)

    @classmethod
    def exp(cls, phi):
        return cls.from_angle(float(np.squeeze(phi)))

    # ---- algebra --------------------------------------------------------
    @staticmethod
    def wedge(phi):
        phi = np.atleast_1d(phi).astype(float)
        out = phi[:, None, None] * _J
        return np.squeeze(out)

    @staticmethod
    def vee(Phi):
        Phi = np.asarray(Phi, dtype=float)
        if Phi.ndim < 3:
            return Phi[1, 0]
        return Phi[:, 1, 0]

    @staticmethod
    def left_jacobian(phi):
        phi = float(np.squeeze(phi))
        if is_small(phi):
            return np.identity(2) + 0.5 * phi * _J
        return (np.sin(phi) / phi) * np.identity(2) + ((1. - np.cos(phi)) / phi) * _J

    @staticmethod
    def inv_left_jacobian(phi):
        phi = float(np.squeeze(phi))
        if is_small(phi):
            return np.identity(2) - 0.5 * phi * _J
        half = 0.5 * phi
        return (half / np.tan(half)) * np.identity(2) - half * _J

    # ---- group operations -----------------------------------------------
    def to_angle(self):
        return np.arctan2(self.mat[1, 0], self.mat[0, 0])

    def log(self):
        return self.to_angle()

    def inv(self):
        return self.__class__(self.mat.T.copy())

    def as_matrix(self):
        return self.mat

    def adjoint(self):
        return 1.

    def normalize(self):
        self.mat = _project_to_so(self.mat)

    def perturb(self, phi):
        self.mat = self.__class__.exp(phi).mat.dot(self.mat)

    def dot(self, other):
        if isinstance(other, self.__class__):
            return self.__class__(self.mat.dot(other.mat))
        other = np.atleast_2d(other)
        if other.shape[1] != self.dim:
            raise ValueError("Vector must have shape ({},) or (N,{})".format(self.dim, self.dim))
        return np.squeeze(other.dot(self.mat.T))
