"""pyslam_amd: MI355X-native nonlinear least-squares (Gauss-Newton / LM) solver
behind the utiasSTARS/pyslam ``Problem`` / residual-block / loss API.

Import ``pyslam`` (the shim package at the repo root) for drop-in use of the
reference's module names, or ``pyslam_amd.*`` directly.
"""
from pyslam_amd.problem import Options, Problem, solve_tables  # noqa: F401

__version__ = "0.1.0"
