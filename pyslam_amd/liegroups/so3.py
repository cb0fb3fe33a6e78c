"""SO(3): spatial rotations (dof 3, 3x3 matrices)."""
import numpy as np

from ._base import MatrixGroup, is_small, _project_to_so, _looks_like_rotation


class SO3(MatrixGroup):
    dof = 3
    dim = 3

    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=float)

    # ---- construction ---------------------------------------------------
    @classmethod
    def identity(cls):
        return cls(np.identity(3))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        mat = np.asarray(mat, dtype=float)
        if not _looks_like_rotation(mat, 3):
            if not normalize:
                raise ValueError("Invalid rotation matrix. Use normalize=True to handle rounding errors.")
            mat = _project_to_so(mat)
        return cls(mat)

    @classmethod
    def rotx(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[1., 0., 0.], [0., c, -s], [0., s, c]]))

    @classmethod
    def roty(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, 0., s], [0., 1., 0.], [-s, 0., c]]))

    @classmethod
    def rotz(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]]))

    @classmethod
    def from_rpy(cls, roll, pitch, yaw):
        return cls.rotz(yaw).dot(cls.roty(pitch).dot(cls.rotx(roll)))

    @classmethod
    def exp(cls, phi):
        phi = np.asarray(phi, dtype=float).reshape(3)
        angle = np.linalg.norm(phi)
        if is_small(angle):
            return cls(np.identity(3) + cls.wedge(phi))
        axis = phi / angle
        s, c = np.sin(angle), np.cos(angle)
        return cls(c * np.identity(3) + (1. - c) * np.outer(axis, axis) + s * cls.wedge(axis))

    # ---- algebra --------------------------------------------------------
    @staticmethod
    def wedge(phi):
        """(3,) -> 3x3 skew matrix; (N,3) -> (N,3,3)."""
        phi = np.atleast_2d(np.asarray(phi, dtype=float))
        if phi.shape[1] != 3:
            raise ValueError("phi must have shape (3,) or (N,3)")
        out = np.zeros((phi.shape[0], 3, 3))
        out[:, 0, 1] = -phi[:, 2]
        out[:, 1, 0] = phi[:, 2]
        out[:, 0, 2] = phi[:, 1]
        out[:, 2, 0] = -phi[:, 1]
        out[:, 1, 2] = -phi[:, 0]
        out[:, 2, 1] = phi[:, 0]
        return np.squeeze(out)

    @staticmethod
    def vee(Phi):
        Phi = np.asarray(Phi, dtype=float)
        if Phi.ndim < 3:
            return np.array([Phi[2, 1], Phi[0, 2], Phi[1, 0]])
        return np.stack([Phi[:, 2, 1], Phi[:, 0, 2], Phi[:, 1, 0]], axis=1)

    @classmethod
    def left_jacobian(cls, phi):
        phi = np.asarray(phi, dtype=float).reshape(3)
        angle = np.linalg.norm(phi)
        if is_small(angle):
            return np.identity(3) + 0.5 * cls.wedge(phi)
        axis = phi / angle
        s, c = np.sin(angle), np.cos(angle)
        return ((s / angle) * np.identity(3)
                + (1. - s / angle) * np.outer(axis, axis)
                + ((1. - c) / angle) * cls.wedge(axis))

    @classmethod
    def inv_left_jacobian(cls, phi):
        phi = np.asarray(phi, dtype=float).reshape(3)
        angle = np.linalg.norm(phi)
        if is_small(angle):
            return np.identity(3) - 0.5 * cls.wedge(phi)
        axis = phi / angle
        half = 0.5 * angle
        hc = half / np.tan(half)
        return (hc * np.identity(3)
                + (1. - hc) * np.outer(axis, axis)
                - half * cls.wedge(axis))

    # ---- group operations -----------------------------------------------
    def log(self):
        cos_angle = np.clip(0.5 * np.trace(self.mat) - 0.5, -1., 1.)
        angle = np.arccos(cos_angle)
        if is_small(angle):
            return self.vee(self.mat - np.identity(3))
        return self.vee((0.5 * angle / np.sin(angle)) * (self.mat - self.mat.T))

    def inv(self):
        return self.__class__(self.mat.T.copy())

    def as_matrix(self):
        return self.mat

    def adjoint(self):
        return self.mat

    def normalize(self):
        self.mat = _project_to_so(self.mat)

    def perturb(self, phi):
        self.mat = self.__class__.exp(phi).mat.dot(self.mat)

    def to_rpy(self):
        pitch = np.arctan2(-self.mat[2, 0], np.sqrt(self.mat[0, 0] ** 2 + self.mat[1, 0] ** 2))
        if np.isclose(pitch, np.pi / 2.):
            yaw = 0.
            roll = np.arctan2(self.mat[0, 1], self.mat[1, 1])
        elif np.isclose(pitch, -np.pi / 2.):
            yaw = 0.
            roll = -np.arctan2(self.mat[0, 1], self.mat[1, 1])
        else:
            sec = 1. / np.cos(pitch)
            yaw = np.arctan2(self.mat[1, 0] * sec, self.mat[0, 0] * sec)
            roll = np.arctan2(self.mat[2, 1] * sec, self.mat[2, 2] * sec)
        return roll, pitch, yaw

    def dot(self, other):
        if isinstance(other, self.__class__):
            return self.__class__(self.mat.dot(other.mat))
        other = np.atleast_2d(other)
        if other.shape[1] != self.dim:
            raise ValueError("Vector must have shape ({},) or (N,{})".format(self.dim, self.dim))
        return np.squeeze(other.dot(self.mat.T))
