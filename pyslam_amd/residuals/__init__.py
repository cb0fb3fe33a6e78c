"""Residual blocks: explicit re-exports of the classes, functions and constants the reference's pkgutil walk
surfaces from its residual modules (pyslam/residuals/__init__.py:1-14), ``fast_se3_odot`` / ``SE3_ODOT_SHAPE``
(reprojection_motion_only_residual.py:9-32) and the helpers those modules import (``stackmul``,
``bilinear_interpolate``, ``SE3``) included.  Not re-exported: the third-party names the walk drags along
(``np``, ``scipy``, ``time``, numba's ``guvectorize`` / ``float32`` / ``float64``)."""
from .pose import PoseResidual, PoseToPoseResidual, PoseToPoseOrientationResidual
from .reprojection import (ReprojectionResidual, ReprojectionMotionOnlyResidual,
                           ReprojectionMotionOnlyBatchResidual,
                           ReprojectionResidualFrameToFrame, fast_se3_odot, SE3_ODOT_SHAPE)
from pyslam_amd.utils import stackmul, bilinear_interpolate
from pyslam_amd.liegroups import SE3
from .quadratic import QuadraticResidual
from .photometric import PhotometricResidualSE3

__all__ = ["PoseResidual", "PoseToPoseResidual", "PoseToPoseOrientationResidual",
           "ReprojectionResidual", "ReprojectionMotionOnlyResidual",
           "ReprojectionMotionOnlyBatchResidual", "ReprojectionResidualFrameToFrame",
           "QuadraticResidual", "PhotometricResidualSE3", "fast_se3_odot", "SE3_ODOT_SHAPE", "stackmul",
           "bilinear_interpolate", "SE3"]
