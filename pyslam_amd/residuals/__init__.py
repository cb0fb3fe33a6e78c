"""Residual blocks (explicit re-exports of every name the reference's
pkgutil walk would surface, pyslam/residuals/__init__.py:1-14)."""
from .pose import PoseResidual, PoseToPoseResidual, PoseToPoseOrientationResidual
from .reprojection import (ReprojectionResidual, ReprojectionMotionOnlyResidual,
                           ReprojectionMotionOnlyBatchResidual,
                           ReprojectionResidualFrameToFrame)
from .quadratic import QuadraticResidual
from .photometric import PhotometricResidualSE3

__all__ = ["PoseResidual", "PoseToPoseResidual", "PoseToPoseOrientationResidual",
           "ReprojectionResidual", "ReprojectionMotionOnlyResidual",
           "ReprojectionMotionOnlyBatchResidual", "ReprojectionResidualFrameToFrame",
           "QuadraticResidual", "PhotometricResidualSE3"]
