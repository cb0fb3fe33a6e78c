"""SE(2) and SE(3): rigid motions stored as (rot, trans).

Twist ordering xi = [rho; phi]; exp(xi) = (SO.exp(phi), J_l(phi) rho);
log = [J_l^-1(phi) t; phi]; perturb is a LEFT perturbation (reference call
sites: problem.py:406, pose_to_pose_residual.py:24, reprojection_residual.py:27).
"""
import numpy as np

from ._base import MatrixGroup
from .so2 import SO2
from .so3 import SO3


class _SEBase(MatrixGroup):
    RotationType = None

    def __init__(self, rot, trans):
        self.rot = rot
        self.trans = np.asarray(trans, dtype=float).reshape(self.dim - 1)

    @classmethod
    def identity(cls):
        return cls(cls.RotationType.identity(), np.zeros(cls.dim - 1))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        mat = np.asarray(mat, dtype=float)
        n = cls.dim - 1
        bottom = np.append(np.zeros(n), 1.)
        if mat.shape != (cls.dim, cls.dim) or not np.allclose(mat[n, :], bottom):
            if not normalize:
                raise ValueError("Invalid transformation matrix. Use normalize=True to handle rounding errors.")
        rot = cls.RotationType.from_matrix(mat[0:n, 0:n], normalize=normalize)
        return cls(rot, mat[0:n, n].copy())

    @classmethod
    def exp(cls, xi):
        xi = np.asarray(xi, dtype=float).reshape(cls.dof)
        n = cls.dim - 1
        rho, phi = xi[0:n], xi[n:]
        return cls(cls.RotationType.exp(phi),
                   cls.RotationType.left_jacobian(phi).dot(rho))

    def log(self):
        phi = self.rot.log()
        rho = self.RotationType.inv_left_jacobian(phi).dot(self.trans)
        return np.hstack([rho, phi])

    def inv(self):
        inv_rot = self.rot.inv()
        return self.__class__(inv_rot, -(inv_rot.mat.dot(self.trans)))

    def as_matrix(self):
        n = self.dim - 1
        out = np.identity(self.dim)
        out[0:n, 0:n] = self.rot.mat
        out[0:n, n] = self.trans
        return out

    def normalize(self):
        self.rot.normalize()

    def perturb(self, xi):
        moved = self.__class__.exp(xi).dot(self)
        self.rot = moved.rot
        self.trans = moved.trans

    def dot(self, other):
        if isinstance(other, self.__class__):
            return self.__class__(self.rot.dot(other.rot),
                                  self.rot.mat.dot(other.trans) + self.trans)
        other = np.atleast_2d(other)
        n = self.dim - 1
        if other.shape[1] == n:
            return np.squeeze(other.dot(self.rot.mat.T) + self.trans)
        if other.shape[1] == self.dim:  # homogeneous coordinates
            return np.squeeze(other.dot(self.as_matrix().T))
        raise ValueError("Vector must have shape ({0},), ({1},), (N,{0}) or (N,{1})".format(n, self.dim))


class SE2(_SEBase):
    dof = 3
    dim = 3
    RotationType = SO2

    @staticmethod
    def wedge(xi):
        xi = np.atleast_2d(np.asarray(xi, dtype=float))
        out = np.zeros((xi.shape[0], 3, 3))
        out[:, 0:2, 0:2] = np.asarray(SO2.wedge(xi[:, 2])).reshape(-1, 2, 2)
        out[:, 0:2, 2] = xi[:, 0:2]
        return np.squeeze(out)

    @staticmethod
    def vee(Xi):
        Xi = np.asarray(Xi, dtype=float)
        if Xi.ndim < 3:
            return np.array([Xi[0, 2], Xi[1, 2], Xi[1, 0]])
        return np.stack([Xi[:, 0, 2], Xi[:, 1, 2], Xi[:, 1, 0]], axis=1)

    def adjoint(self):
        out = np.identity(3)
        out[0:2, 0:2] = self.rot.mat
        out[0, 2] = self.trans[1]
        out[1, 2] = -self.trans[0]
        return out

    @classmethod
    def odot(cls, p, directional=False):
        """(2,) -> 2x3 [I | J^T-rotated p]; (3,) homogeneous -> 3x3."""
        p = np.atleast_2d(np.asarray(p, dtype=float))
        out = np.zeros((p.shape[0], p.shape[1], 3))
        if p.shape[1] == 2:
            out[:, 0, 2] = -p[:, 1]
            out[:, 1, 2] = p[:, 0]
            if not directional:
                out[:, 0:2, 0:2] = np.identity(2)
        elif p.shape[1] == 3:
            out[:, 0, 2] = -p[:, 1]
            out[:, 1, 2] = p[:, 0]
            out[:, 0:2, 0:2] = p[:, 2][:, None, None] * np.identity(2)
        else:
            raise ValueError("p must have shape (2,), (3,), (N,2) or (N,3)")
        return np.squeeze(out)


class SE3(_SEBase):
    dof = 6
    dim = 4
    RotationType = SO3

    @staticmethod
    def wedge(xi):
        xi = np.atleast_2d(np.asarray(xi, dtype=float))
        out = np.zeros((xi.shape[0], 4, 4))
        out[:, 0:3, 0:3] = SO3.wedge(xi[:, 3:6]).reshape(-1, 3, 3)
        out[:, 0:3, 3] = xi[:, 0:3]
        return np.squeeze(out)

    @staticmethod
    def vee(Xi):
        Xi = np.asarray(Xi, dtype=float)
        if Xi.ndim < 3:
            return np.hstack([Xi[0:3, 3], SO3.vee(Xi[0:3, 0:3])])
        return np.hstack([Xi[:, 0:3, 3], SO3.vee(Xi[:, 0:3, 0:3])])

    def adjoint(self):
        C = self.rot.mat
        out = np.zeros((6, 6))
        out[0:3, 0:3] = C
        out[0:3, 3:6] = SO3.wedge(self.trans).dot(C)
        out[3:6, 3:6] = C
        return out

    @classmethod
    def odot(cls, p, directional=False):
        """(3,) -> 3x6 [I | -p^]; (N,3) -> (N,3,6); homogeneous (4,) -> 4x6."""
        p = np.atleast_2d(np.asarray(p, dtype=float))
        out = np.zeros((p.shape[0], p.shape[1], 6))
        if p.shape[1] == 3:
            out[:, 0:3, 3:6] = SO3.wedge(-p).reshape(-1, 3, 3)
            if not directional:
                out[:, 0:3, 0:3] = np.identity(3)
        elif p.shape[1] == 4:
            out[:, 0:3, 3:6] = SO3.wedge(-p[:, 0:3]).reshape(-1, 3, 3)
            out[:, 0:3, 0:3] = p[:, 3][:, None, None] * np.identity(3)
        else:
            raise ValueError("p must have shape (3,), (4,), (N,3) or (N,4)")
        return np.squeeze(out)
