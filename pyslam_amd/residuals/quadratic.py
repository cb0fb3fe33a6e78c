"""Scalar parabola-fit residual used by the reference's own unit tests
(reference pyslam/residuals/quadratic_residual.py:4-32).  Parameters are plain
floats, so this block is solved through the host-evaluated generic path."""
import numpy as np


class QuadraticResidual:
    KIND = "generic"

    def __init__(self, x, y, stiffness):
        self.x = np.array([x])
        self.y = np.array([y])
        self.stiffness = np.array([stiffness])

    def evaluate(self, params, compute_jacobians=None):
        a, b, c = params[0], params[1], params[2]
        x = self.x
        residual = self.stiffness * (a * x * x + b * x + c - self.y)
        if not compute_jacobians:
            return residual

        jacobians = [None, None, None]
        if compute_jacobians[0]:
            jacobians[0] = self.stiffness * x * x
        if compute_jacobians[1]:
            jacobians[1] = self.stiffness * x
        if compute_jacobians[2]:
            jacobians[2] = self.stiffness * 1.
        return residual, np.squeeze(jacobians)
